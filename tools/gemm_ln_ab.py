#!/usr/bin/env python3
"""out-proj / fc2 with the LayerNorm that follows: two launches (gemm_nt + layernorm_fwd) against the fused launch (gemm_nt_ln),
at the headline shape (M = 32 x 1025), 200 back-to-back repetitions behind 50 untimed ones.  usage: python tools/gemm_ln_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def timed(fn, n=200, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    M = 32 * 1025
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, K in (("out_proj K768", 768), ("fc2 K3072", 3072)):
        a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
        w = (torch.randn(768, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
        bias = torch.randn(768, generator=g).to(dev)
        resid = torch.randn(M, 768, generator=g).to(dev)
        gamma, beta = torch.ones(768, device=dev), torch.zeros(768, device=dev)
        xo = torch.empty(M, 768, device=dev)
        ho = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)

        def two():
            ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=xo)
            ops.layernorm_fwd(xo, gamma, beta, 1e-5, out_bf16=ho)

        def gemm_only():
            ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=xo)

        def fused():
            ops.gemm_nt_ln(a, w, bias, resid, gamma, beta, 1e-5, out_f32=xo, ln_out=ho)

        t2, tg, tf = timed(two), timed(gemm_only), timed(fused)
        print(f"{name:14s} gemm {tg:7.1f} us | gemm + layernorm {t2:7.1f} us | fused {tf:7.1f} us  ({t2 - tf:+.1f} us saved)")


if __name__ == "__main__":
    main()
