"""Calibration only: the vendor library's transposed product dY^T.X (torch.matmul, bf16 in / bf16 out) on the tower's weight-gradient
shapes beside lc2is_amd's gemm_tn (bf16 in / fp32 out + fused bias gradient), one problem per launch.
  python tools/vendor_wgrad_ref.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit

dev = torch.device("cuda:0")
M = 32 * 1025
for name, N, K in (("qkv", 2304, 768), ("out_proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    dyt = dy.t()
    out = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    tv = min(timeit(lambda: torch.matmul(dyt, x, out=out), iters=10, warm=3) for _ in range(3))
    to = min(timeit(lambda: ops.gemm_tn(dy, x, dw, db=db), iters=10, warm=3) for _ in range(3))
    fl = 2.0 * M * N * K
    print(f"{name:9s} dW[{N:4d},{K:4d}]: vendor {tv * 1e6:7.1f} us {fl / tv / 1e12:6.0f} TF/s | lc2is_amd (single problem, split + reduce) {to * 1e6:7.1f} us {fl / to / 1e12:6.0f} TF/s", flush=True)
print("(in the step the 72 problems of the tower run as ONE grouped grid: 4.8 ms = 1.1 PF/s, tools/tn_tower.py)")
