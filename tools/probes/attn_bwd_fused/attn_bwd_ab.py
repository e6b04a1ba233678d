"""A/B of the attention backward forms in ONE process (interleaved rounds, random data): the fused five-product kernel
(+ its delta prologue and state memset) against the two-launch form, at the vision-tower shape of config 2
(B=32, H=12, S=1025, D=64), the decoder shape (H=8, S=1024, D=96) and ViT-L/14 @640 (H=16, S=2026, D=64).
usage: python tools/attn_bwd_ab.py [--shapes vit,dec,vitl] [--iters N] [--rounds R]"""
import argparse
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--shapes", default="vit,dec")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = dict(vit=(32, 12, 1025, 64), dec=(32, 8, 1024, 96), vitl=(8, 16, 2026, 64), sr=(16, 8, 4096, 64))
    for name in a.shapes.split(","):
        B, H, S, D = shapes[name]
        Sk = S // 4 if name == "sr" else S
        C = H * D
        g = torch.Generator(device=dev).manual_seed(1)
        q = torch.randn(B * S, C, device=dev, generator=g).bfloat16()
        k = torch.randn(B * Sk, C, device=dev, generator=g).bfloat16()
        v = torch.randn(B * Sk, C, device=dev, generator=g).bfloat16()
        do = (torch.randn(B * S, C, device=dev, generator=g) * 0.5).bfloat16()
        sc = D ** -0.5
        o, lse = ops.attention_fwd(q, k, v, B, H, S, Sk, D, sc)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        fl = 10.0 * B * H * S * Sk * D   # the algorithm's five products

        def run(fused):
            ops._ATTN_BWD_FUSED = fused
            ops.attention_bwd(q, k, v, o, do, lse, B, H, S, Sk, D, sc, dq=dq, dk=dk, dv=dv)

        res = {True: [], False: []}
        for f in (True, False):
            for _ in range(3):
                run(f)
        ops._ATTN_BWD_FUSED = True
        ops.attention_bwd_status()
        for _ in range(a.rounds):
            for f in (True, False):
                res[f].append(timeit(lambda: run(f), a.iters))
        ops._ATTN_BWD_FUSED = True
        for f, label in ((True, "fused (delta + memset + main)"), (False, "two launches (dq; dk/dv)")):
            med, mn = statistics.median(res[f]), min(res[f])
            print(f"{name} bwd {label:32s} median {med:8.1f} us  min {mn:8.1f} us  {fl / mn / 1e6:7.0f} TF/s "
                  f"({fl / mn / 1e6 / 2.5e3:.3f} of 2.5 PF, algorithmic 5 products)", flush=True)


if __name__ == "__main__":
    main()
