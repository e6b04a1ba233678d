#!/usr/bin/env python3
"""The eight NT GEMMs of one ViT-B/16 layer at the headline shape (M = 32 x 1025 rows): each one PLAIN (bf16 out, no bias)
and with the epilogue the training step uses, timed two ways — BURST (3 launches, the way tools/gemm_shapes.py and round 3's
tables time them: the chip has not settled into its power state) and SUSTAINED (200 back-to-back launches behind 100 untimed
ones, ~20-60 ms of continuous matrix work: the clock the chip holds inside a training step).  The difference between the
columns splits the "isolated vs in-step" gap of the round-3 verdict (item 5) into epilogue bytes and clock.
usage: python tools/gemm_epilogue_cost.py [tile_cfg for the bf16-output shapes, default 0 = dispatch]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device("cuda:0")
    M = 32 * 1025
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s, scale=1.0, dt=torch.bfloat16):
        return (torch.randn(*s, generator=g) * scale).to(dt).to(dev)

    x768, x3072, x2304 = rnd(M, 768), rnd(M, 3072), rnd(M, 2304)
    res = rnd(M, 768, dt=torch.float32)
    z3072 = rnd(M, 3072)
    rows = [
        ("qkv       N2304 K768  bias, bf16", 2304, 768, x768, dict(out_bf16=True), True),
        ("out_proj  N768  K768  bias + fp32 residual -> fp32", 768, 768, x768, dict(resid=res, out_bf16=None, out_f32=True), False),
        ("fc1       N3072 K768  bias, quick_gelu, saves z", 3072, 768, x768, dict(act=ops.ACT_QUICK_GELU, out_bf16=True, aux_out=True), True),
        ("fc2       N768  K3072 bias + fp32 residual -> fp32", 768, 3072, x3072, dict(resid=res, out_bf16=None, out_f32=True), False),
        ("dfc2      N3072 K768  x quick_gelu'(z)", 3072, 768, x768, dict(act=ops.ACT_DQUICK_GELU, aux_in=z3072, out_bf16=True), True),
        ("dfc1      N768  K3072 -> fp32", 768, 3072, x3072, dict(out_bf16=None, out_f32=True), False),
        ("dout_proj N768  K768  bf16", 768, 768, x768, dict(out_bf16=True), True),
        ("dqkv      N768  K2304 -> fp32", 768, 2304, x2304, dict(out_bf16=None, out_f32=True), False),
    ]

    def timed(fn, n, warm):
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

    print(f"{'GEMM (M = 32800)':52s} | {'plain burst':>11s} {'plain sust.':>11s} | {'epilogue burst':>14s} {'epilogue sust.':>14s} | "
          f"{'epilogue cost':>13s} {'clock cost':>10s}   (us; cost columns on the sustained / epilogue forms)")
    tot = [0.0] * 4
    for name, N, K, a, kw, bf16_out in rows:
        w = rnd(N, K, scale=0.03)
        bias = rnd(N, dt=torch.float32)
        outs = {k: (torch.empty(M, N, dtype=torch.bfloat16 if k != "out_f32" else torch.float32, device=dev) if v is True else v)
                for k, v in kw.items()}
        plain_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        c_epi = cfg if bf16_out else 0
        plain = lambda: ops.gemm_nt(a, w, None, out_bf16=plain_out, tile_cfg=cfg)          # noqa: E731
        epi = lambda: ops.gemm_nt(a, w, bias, tile_cfg=c_epi, **outs)                       # noqa: E731
        pb = min(timed(plain, 3, 2) for _ in range(5))
        eb = min(timed(epi, 3, 2) for _ in range(5))
        ps = timed(plain, 200, 100)
        es = timed(epi, 200, 100)
        for i, v in enumerate((pb, ps, eb, es)):
            tot[i] += v
        print(f"{name:52s} | {pb:11.1f} {ps:11.1f} | {eb:14.1f} {es:14.1f} | {es - ps:13.1f} {es - eb:10.1f}", flush=True)
    print(f"{'layer total':52s} | {tot[0]:11.1f} {tot[1]:11.1f} | {tot[2]:14.1f} {tot[3]:14.1f} | {tot[3] - tot[1]:13.1f} {tot[3] - tot[2]:10.1f}")
    print(f"x 12 layers: plain burst {tot[0] * 12e-3:.2f} ms, plain sustained {tot[1] * 12e-3:.2f} ms, with epilogues burst {tot[2] * 12e-3:.2f} ms, "
          f"with epilogues sustained {tot[3] * 12e-3:.2f} ms")


if __name__ == "__main__":
    main()
