#!/usr/bin/env python3
"""A/B of the de-synchronised start (LC2IS_GEMM_STAGGER_NS) of the persistent NT GEMM on the three multi-round problems of a
ViT-B/16 layer at the headline shape (M = 32 x 1025): qkv, fc1 (quick_gelu + saved z), dfc2 (its derivative).  Interleaved
rounds in one process, median and min per setting.  Usage: python tools/gemm_stagger_ab.py [ns,ns,...] [rounds]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def main():
    settings = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 5000, 10000, 15000, 20000, 30000]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    dev = torch.device("cuda:0")
    M = 32 * 1025
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s, scale=1.0, dt=torch.bfloat16):
        return (torch.randn(*s, generator=g) * scale).to(dt).to(dev)

    x768 = rnd(M, 768)
    z3072 = rnd(M, 3072)
    shapes = []
    for name, N, K, a, kw in [
        ("qkv   N2304 K768  bf16", 2304, 768, x768, dict(out_bf16=True)),
        ("fc1   N3072 K768  qgelu+aux", 3072, 768, x768, dict(act=ops.ACT_QUICK_GELU, out_bf16=True, aux_out=True)),
        ("dfc2  N3072 K768  dqgelu", 3072, 768, x768, dict(act=ops.ACT_DQUICK_GELU, aux_in=z3072, out_bf16=True)),
    ]:
        w = rnd(N, K, scale=0.03)
        bias = rnd(N, dt=torch.float32)
        outs = {k: (torch.empty(M, N, dtype=torch.bfloat16, device=dev) if v is True else v) for k, v in kw.items()}
        shapes.append((name, N, K, a, w, bias, outs))
    # reference outputs without the stagger: the schedule must not change a bit
    os.environ["LC2IS_GEMM_STAGGER_NS"] = "0"
    refs = []
    for (name, N, K, a, w, bias, outs) in shapes:
        ops.gemm_nt(a, w, bias, **outs)
        torch.cuda.synchronize()
        refs.append({k: v.clone() for k, v in outs.items() if k in ("out_bf16", "aux_out") and torch.is_tensor(v)})
    times = {(s[0], c): [] for s in shapes for c in settings}
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    for r in range(rounds + 1):
        for si, (name, N, K, a, w, bias, outs) in enumerate(shapes):
            for c in settings:
                os.environ["LC2IS_GEMM_STAGGER_NS"] = str(c)
                e0, e1 = ev(), ev()
                ops.gemm_nt(a, w, bias, **outs)
                e0.record()
                for _ in range(3):
                    ops.gemm_nt(a, w, bias, **outs)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[(name, c)].append(e0.elapsed_time(e1) / 3 * 1e3)
                else:
                    for k, v in refs[si].items():
                        assert torch.equal(outs[k], v), f"{name}: stagger {c} changed {k}"
    for (name, N, K, *_rest) in shapes:
        fl = 2.0 * M * N * K
        line = f"{name:30s}"
        for c in settings:
            t = sorted(times[(name, c)])
            med, mn = t[len(t) // 2], t[0]
            line += f" | {c:6d} ns {med:6.1f} us (min {mn:6.1f}) {fl / med / 1e6:5.0f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
