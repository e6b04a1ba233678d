"""Ablation timing of the 256x256 LDS-DMA GEMM (diagnostic kernel builds; their outputs are wrong by design).
cfg 4 full | 41 no epilogue | 42 no in-loop DMA | 43 neither | 45 no epilogue + fragments read once | 47 MFMA + barrier only"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit

dev = torch.device("cuda:0")
for (M, N, K) in [(32768, 3072, 768), (32768, 768, 3072), (32768, 768, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {}
    for rnd in range(3):
        for cfg in (4, 41, 43, 47, 49):
            t = timeit(lambda: ops.gemm_nt(a, w, None, out_bf16=out, tile_cfg=cfg), iters=10, warm=2)
            res.setdefault(cfg, []).append(t)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"cfg{c}={min(v)*1e6:6.1f}us({2*M*N*K/min(v)/1e12:5.0f}TF)" for c, v in res.items()), flush=True)
