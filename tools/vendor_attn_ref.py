"""Calibration only (not a product path): torch's scaled_dot_product_attention (whatever fused kernel torch 2.10+rocm7.0 picks on
gfx950: AOTriton / CK flash attention, or the math fallback) at the vision tower's attention shape, beside lc2is_amd's kernels.
  python tools/vendor_attn_ref.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
import torch.nn.functional as F
from lc2is_amd import ops
from bench_kernels import timeit

dev = torch.device("cuda:0")
B, H, S, D = 32, 12, 1025, 64
C = H * D
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(B * S, 3 * C, device=dev, generator=g).bfloat16()
do = (torch.randn(B * S, C, device=dev, generator=g) * 0.5).bfloat16()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
sc = D ** -0.5
fl = 4.0 * B * H * S * S * D

# ours (token-major, packed qkv read in place)
o, lse = ops.attention_fwd(q, k, v, B, H, S, S, D, sc)
dqkv = torch.empty_like(qkv)
tf = min(timeit(lambda: ops.attention_fwd(q, k, v, B, H, S, S, D, sc, out=o), 20) for _ in range(3))
tb = min(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, D, sc, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:]), 20) for _ in range(3))
print(f"lc2is_amd   fwd {tf * 1e6:7.1f} us {fl / tf / 1e12:5.0f} TF/s | bwd {tb * 1e6:7.1f} us {2.5 * fl / tb / 1e12:5.0f} TF/s", flush=True)

# torch SDPA on [B,H,S,D] tensors (its preferred layout; the permutes are not timed)
qh, kh, vh = (t.reshape(B, S, H, D).transpose(1, 2).contiguous().requires_grad_(True) for t in (q, k, v))
doh = do.reshape(B, S, H, D).transpose(1, 2).contiguous()
for name, backends in (("flash", [torch.nn.attention.SDPBackend.FLASH_ATTENTION]), ("efficient", [torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION]),
                       ("default", None)):
    try:
        def fwd():
            return F.scaled_dot_product_attention(qh, kh, vh, scale=sc)
        if backends:
            with torch.nn.attention.sdpa_kernel(backends):
                out = fwd()
                t1 = min(timeit(fwd, 20) for _ in range(3))
                def fb():
                    oo = F.scaled_dot_product_attention(qh, kh, vh, scale=sc)
                    oo.backward(doh)
                t2 = min(timeit(fb, 10) for _ in range(3))
        else:
            out = fwd()
            t1 = min(timeit(fwd, 20) for _ in range(3))
            def fb():
                oo = F.scaled_dot_product_attention(qh, kh, vh, scale=sc)
                oo.backward(doh)
            t2 = min(timeit(fb, 10) for _ in range(3))
        print(f"torch SDPA [{name:9s}] fwd {t1 * 1e6:7.1f} us {fl / t1 / 1e12:5.0f} TF/s | fwd+bwd {t2 * 1e6:7.1f} us -> bwd ~{(t2 - t1) * 1e6:7.1f} us {2.5 * fl / max(t2 - t1, 1e-9) / 1e12:5.0f} TF/s", flush=True)
    except Exception as e:   # a backend that this build does not have on gfx950
        print(f"torch SDPA [{name}] unavailable: {type(e).__name__}: {str(e)[:120]}", flush=True)
