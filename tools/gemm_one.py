#!/usr/bin/env python3
"""One NT GEMM shape, chosen tile configurations, a few launches each (for rocprofv3 counter passes: tools/pmc_gemm.sh).
usage: python tools/gemm_one.py M N K cfg[,cfg...] [iters]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
cfgs = [int(c) for c in sys.argv[4].split(",")]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
bias = torch.randn(N, device=dev, generator=g)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for c in cfgs:
    for _ in range(iters):
        ops.gemm_nt(a, w, bias, out_bf16=out, tile_cfg=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt(a, w, bias, out_bf16=out, tile_cfg=c)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print(f"M={M} N={N} K={K} cfg{c}: {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.0f} TF/s", flush=True)
