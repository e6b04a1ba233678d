"""Epilogue-only timing: K=64 GEMMs (one K tile) of the step's epilogue kinds vs a device copy of the same bytes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4]
dev = torch.device("cuda:0")
M, K = 32768, 64
for N in (768, 3072):
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); z = torch.full((M, N), 0.5, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(M, N, device=dev); r = torch.ones(M, N, device=dev)
    kinds = {"plain(bf16 out)": (lambda c: ops.gemm_nt(a, w, None, out_bf16=o, tile_cfg=c), M * N * 2),
             "resid(f32 in+out)": (lambda c: ops.gemm_nt(a, w, None, resid=r, out_f32=f, tile_cfg=c), M * N * 8),
             "f32 out": (lambda c: ops.gemm_nt(a, w, None, out_f32=f, tile_cfg=c), M * N * 4),
             "act(z+act out)": (lambda c: ops.gemm_nt(a, w, None, out_bf16=o, aux_out=z, act=ops.ACT_QUICK_GELU, tile_cfg=c), M * N * 4),
             "dact(z in, bf16 out)": (lambda c: ops.gemm_nt(a, w, None, aux_in=z, out_bf16=o, act=ops.ACT_DQUICK_GELU, tile_cfg=c), M * N * 4)}
    for name, (fn, nbytes) in kinds.items():
        line = []
        for c in cfgs:
            t = min(timeit(lambda: fn(c), iters=10, warm=2) for _ in range(3))
            line.append(f"cfg{c}={t*1e6:6.1f}us {nbytes/t/1e12:4.2f}TB/s")
        print(f"N={N:4d} {name:22s}: " + "  ".join(line), flush=True)
    t = min(timeit(lambda: o.copy_(z), iters=10, warm=2) for _ in range(3))
    print(f"N={N:4d} torch copy bf16 (r+w {M*N*4/1e6:.0f} MB): {t*1e6:6.1f}us {M*N*4/t/1e12:4.2f}TB/s", flush=True)
    t = min(timeit(lambda: o.fill_(1.0), iters=10, warm=2) for _ in range(3))
    print(f"N={N:4d} torch fill bf16 (w {M*N*2/1e6:.0f} MB): {t*1e6:6.1f}us {M*N*2/t/1e12:4.2f}TB/s", flush=True)
