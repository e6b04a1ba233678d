"""Calibration only (not a product path): the vendor library's time (torch.matmul -> hipBLASLt / rocBLAS assembly kernels) on the
vision tower's GEMM shapes, plain bf16 A.W^T without bias / activation, beside lc2is_amd's NT GEMM on the same operands.
  python tools/vendor_gemm_ref.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit

dev = torch.device("cuda:0")
M = 32 * 1025
for name, N, K in (("qkv", 2304, 768), ("out_proj / dout_proj", 768, 768), ("fc1 / dfc2", 3072, 768), ("fc2 / dfc1", 768, 3072), ("dqkv", 768, 2304)):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wt = w.t()
    tv = min(timeit(lambda: torch.matmul(a, wt, out=out), iters=20, warm=5) for _ in range(3))
    to = min(timeit(lambda: ops.gemm_nt(a, w, None, out_bf16=out), iters=20, warm=5) for _ in range(3))
    fl = 2.0 * M * N * K
    print(f"{name:22s} N={N:4d} K={K:4d}: vendor {tv * 1e6:7.1f} us {fl / tv / 1e12:6.0f} TF/s | lc2is_amd {to * 1e6:7.1f} us {fl / to / 1e12:6.0f} TF/s", flush=True)
