#!/usr/bin/env python3
"""LayerNorm backward at the towers' shape (M = 32 x 1025, C = 768; bf16 dy / dres / out): the lean kernel (next-row prefetch, 3 waves
per SIMD) against the generic one (LC2IS_LN_BWD_LEAN=0 is read once per process, so the generic kernel is reached through the fp32-dres
entry, which moves 2 more bytes per element: its time is an UPPER bound for the generic bf16 form), interleaved rounds.
usage: python tools/ln_bwd_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for M, C in ((32 * 1025, 768), (8 * 2026, 1024)):
        x = torch.randn(M, C, device=dev)
        gamma = torch.randn(C, device=dev)
        dy = torch.randn(M, C, device=dev).bfloat16()
        dres = torch.randn(M, C, device=dev).bfloat16()
        dres32 = dres.float()
        _, _, mean, rstd = ops.layernorm_fwd(x, gamma, None)
        variants = {
            "lean  (bf16 dres, bf16 out)": lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, want_f32=False),
            "generic (fp32 dres, bf16 out)": lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres32, want_f32=False),
            "generic (fp32 dres, fp32 + bf16 out: round 4's form)": lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres32),
        }
        ts = {k: [] for k in variants}
        for r in range(8):
            for k, fn in variants.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        for k, v in ts.items():
            v.sort()
            print(f"M={M} C={C}  {k:55s} median {v[len(v) // 2]:6.1f} us (min {v[0]:6.1f})", flush=True)


if __name__ == "__main__":
    main()
