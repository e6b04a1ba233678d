"""A/B of the 256x384-tile NT GEMM (tile_cfg 16) against the 256x256 plan on the tower's N = 768 fp32-residual problems
(out-proj, fc2, dfc1, dqkv at M = 32 800 / 32 768): bitwise comparison and HIP-event times, interleaved rounds.
usage: python tools/gemm_w384_ab.py [--iters N] [--rounds R]"""
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
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, K, use_res in (("out_proj", 32768, 768, True), ("fc2", 32768, 3072, True), ("dfc1", 32768, 3072, False),
                                ("dqkv", 32768, 2304, False), ("fc2 ragged (auto plans)", 32800, 3072, True)):
        N = 768
        x = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
        bias = torch.randn(N, device=dev, generator=g) if use_res else None
        resid = torch.randn(M, N, device=dev, generator=g) if use_res else None
        o4 = torch.empty(M, N, device=dev)
        o16 = torch.empty(M, N, device=dev)
        cfgs = (4, 16) if M % 256 == 0 else (4, 0)

        def run(c, out):
            ops.gemm_nt(x, w, bias, resid=resid, out_bf16=False, out_f32=out, tile_cfg=c)

        run(cfgs[0], o4)
        run(cfgs[1], o16)
        torch.cuda.synchronize()
        same = torch.equal(o4, o16)
        res = {c: [] for c in cfgs}
        for _ in range(a.rounds):
            for c, o in zip(cfgs, (o4, o16)):
                res[c].append(timeit(lambda: run(c, o), a.iters))
        fl = 2.0 * M * N * K
        line = f"{name:24s} M={M} K={K} bitwise equal: {same} |"
        for c in cfgs:
            mn = min(res[c])
            line += f" cfg{c}: median {statistics.median(res[c]):7.1f} min {mn:7.1f} us {fl / mn / 1e6:6.0f} TF/s |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
