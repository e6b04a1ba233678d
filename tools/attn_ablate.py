"""Ablation timing of the attention forward (diagnostic builds, LC2IS_ATTN_DBG; outputs are wrong by design)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit
dev = torch.device("cuda:0")
B, H, S, D = 32, 12, 1025, 64
C = H * D
qkv = torch.randn(B * S, 3 * C, device=dev).bfloat16()
t = min(timeit(lambda: ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, S, S, D, D ** -0.5, save_lse=True), iters=20) for _ in range(3))
print(f"dbg={os.environ.get('LC2IS_ATTN_DBG', '0'):>2s}: {t * 1e6:7.1f} us  {4 * B * H * S * S * D / t / 1e12:6.0f} TF/s", flush=True)
