"""Attention micro-benchmark at the vision-tower shape of config 2 (B=32, H=12, S=1025, D=64) and the decoder shape
(H=8, S=1024, D=96): time of the forward / dQ / dK-dV launches (HIP events, median of rounds) and a numerics check against
fp32 torch on a few (batch, head) pairs.  usage: python tools/attn_bench.py [--iters N] [--check] [--only fwd|bwd]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--shapes", default="vit,dec")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = dict(vit=(32, 12, 1025, 64), dec=(32, 8, 1024, 96), vitl=(8, 16, 2026, 64))
    for name in a.shapes.split(","):
        B, H, S, D = shapes[name]
        C = H * D
        g = torch.Generator(device=dev).manual_seed(1)
        qkv = torch.randn(B * S, 3 * C, device=dev, generator=g).bfloat16()
        do = (torch.randn(B * S, C, device=dev, generator=g) * 0.5).bfloat16()
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        sc = D ** -0.5
        o, lse = ops.attention_fwd(q, k, v, B, H, S, S, D, sc)
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * B * H * S * S * D
        if a.only in ("", "fwd"):
            t = min(timeit(lambda: ops.attention_fwd(q, k, v, B, H, S, S, D, sc, out=o), a.iters) for _ in range(a.rounds))
            print(f"{name} fwd  {t * 1e6:8.1f} us  {fl / t / 1e12:7.0f} TF/s ({fl / t / 2.5e15:.3f} of 2.5 PF)", flush=True)
        if a.only in ("", "bwd"):
            t = min(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, D, sc, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C],
                                                     dv=dqkv[:, 2 * C:]), a.iters) for _ in range(a.rounds))
            print(f"{name} bwd  {t * 1e6:8.1f} us  {2.5 * fl / t / 1e12:7.0f} TF/s ({2.5 * fl / t / 2.5e15:.3f} of 2.5 PF)  (dQ + dK/dV launches)",
                  flush=True)
        if a.check:
            o, lse = ops.attention_fwd(q, k, v, B, H, S, S, D, sc)
            ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, D, sc, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
            worst = {}
            for b, h in ((0, 0), (B - 1, H - 1), (B // 2, 3)):
                sl = slice(h * D, (h + 1) * D)
                rows = slice(b * S, (b + 1) * S)
                qq, kk, vv = (t[rows, sl].float().requires_grad_(True) for t in (q, k, v))
                p = torch.softmax(qq @ kk.T * sc, -1)
                ro = p @ vv
                ro.backward(do[rows, sl].float())
                for nm, got, ref in (("o", o[rows, sl], ro), ("dq", dqkv[rows, sl], qq.grad),
                                     ("dk", dqkv[rows, C + h * D:C + (h + 1) * D], kk.grad),
                                     ("dv", dqkv[rows, 2 * C + h * D:2 * C + (h + 1) * D], vv.grad)):
                    r = ((got.float() - ref).norm() / ref.norm()).item()
                    worst[nm] = max(worst.get(nm, 0.0), r)
            print(f"{name} check rel-L2 vs fp32 torch: " + " ".join(f"{k}={v:.2e}" for k, v in worst.items()), flush=True)
            assert worst["o"] < 8e-3 and max(worst["dq"], worst["dk"], worst["dv"]) < 1.5e-2, worst


if __name__ == "__main__":
    main()
