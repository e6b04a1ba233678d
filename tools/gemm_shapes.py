#!/usr/bin/env python3
"""The eight NT GEMMs of one ViT-B/16 layer (forward + data gradients) at the headline shape (M = 32 x 1025), each with the
epilogue the step uses, through the default dispatch (or tile_cfg=<n>[,<n>...] as argv[1]); interleaved rounds in one
process, median and min per shape.  Usage: python tools/gemm_shapes.py [tile_cfg] [rounds]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def main():
    cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev = torch.device("cuda:0")
    M = 32 * 1025
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s, scale=1.0, dt=torch.bfloat16):
        return (torch.randn(*s, generator=g) * scale).to(dt).to(dev)

    shapes = []   # name, N, K, kwargs builder
    x768, x3072, x2304 = rnd(M, 768), rnd(M, 3072), rnd(M, 2304)
    res = rnd(M, 768, dt=torch.float32)
    z3072 = rnd(M, 3072)
    for name, N, K, a, kw in [
        ("qkv        N2304 K768  bf16", 2304, 768, x768, dict(out_bf16=True)),
        ("out_proj   N768  K768  f32+res", 768, 768, x768, dict(resid=res, out_bf16=None, out_f32=True)),
        ("fc1        N3072 K768  qgelu+aux", 3072, 768, x768, dict(act=ops.ACT_QUICK_GELU, out_bf16=True, aux_out=True)),
        ("fc2        N768  K3072 f32+res", 768, 3072, x3072, dict(resid=res, out_bf16=None, out_f32=True)),
        ("dfc2       N3072 K768  dqgelu", 3072, 768, x768, dict(act=ops.ACT_DQUICK_GELU, aux_in=z3072, out_bf16=True)),
        ("dfc1       N768  K3072 f32", 768, 3072, x3072, dict(out_bf16=None, out_f32=True)),
        ("dout_proj  N768  K768  bf16", 768, 768, x768, dict(out_bf16=True)),
        ("dqkv       N768  K2304 f32", 768, 2304, x2304, dict(out_bf16=None, out_f32=True)),
    ]:
        w = rnd(N, K, scale=0.03)
        bias = rnd(N, dt=torch.float32)
        outs = {k: (torch.empty(M, N, dtype=torch.bfloat16 if k != "out_f32" else torch.float32, device=dev) if v is True else v)
                for k, v in kw.items()}
        shapes.append((name, N, K, a, w, bias, outs))
    times = {(s[0], c): [] for s in shapes for c in cfgs}
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    for r in range(rounds + 1):
        for (name, N, K, a, w, bias, outs) in shapes:
            for c in cfgs:
                e0, e1 = ev(), ev()
                try:
                    ops.gemm_nt(a, w, bias, tile_cfg=c, **outs)     # untimed first call of the pair warms the caches the same way
                    e0.record()
                    for _ in range(3):
                        ops.gemm_nt(a, w, bias, tile_cfg=c, **outs)
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[(name, c)].append(e0.elapsed_time(e1) / 3 * 1e3)
                except RuntimeError as e:
                    times[(name, c)] = None
                    if r == 0:
                        print(f"{name} cfg{c}: {e}")
    tot = {c: 0.0 for c in cfgs}
    fl_tot = 0.0
    for (name, N, K, *_), in [(s,) for s in shapes]:
        fl = 2.0 * M * N * K
        fl_tot += fl
        line = f"{name:34s}"
        for c in cfgs:
            t = times[(name, c)]
            if not t:
                line += f" | cfg{c:<2d}      n/a          "
                tot[c] = float("nan")
                continue
            t = sorted(t)
            med, mn = t[len(t) // 2], t[0]
            tot[c] += med
            line += f" | cfg{c:<2d} {med:7.1f} us (min {mn:7.1f}) {fl / med / 1e6:6.0f} TF/s"
        print(line, flush=True)
    for c in cfgs:
        print(f"layer total cfg{c}: {tot[c]:.1f} us  = {fl_tot / tot[c] / 1e6:.0f} TF/s over the 8 GEMMs")


if __name__ == "__main__":
    main()
