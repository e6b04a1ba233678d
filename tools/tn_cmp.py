"""Weight-gradient GEMM (TN) timing on the step's shapes; run with LC2IS_GEMM_TN_CFG=1 (128x128) / 2 (256x256 LDS-DMA)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit
dev = torch.device("cuda:0")
M = 32800
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ref = dy.float().T @ x.float()
    ops.gemm_tn(dy, x, dw, db=db)
    err = ((dw - ref).norm() / ref.norm()).item(); eb = ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item()
    t = min(timeit(lambda: ops.gemm_tn(dy, x, dw, db=db), iters=10, warm=2) for _ in range(3))
    print(f"cfg={os.environ.get('LC2IS_GEMM_TN_CFG','0')} N={N:4d} K={K:4d}: {t*1e6:7.1f}us {2*M*N*K/t/1e12:6.0f}TF  rel_err={err:.2e} db_err={eb:.2e}", flush=True)
