#!/usr/bin/env python3
"""What the DATA-PARALLEL configuration of the step costs on ONE GPU, taken apart: a 1-rank `nccl` (= RCCL) process
group, the headline step (B = 32, 512x512), interleaved in one process:
  plain      TrainStep without a reducer (whole-tower weight-gradient grid)
  groups     GradReducer attached, every all_reduce replaced by a no-op: the DP wiring alone (3-layer weight-gradient groups,
             per-layer callbacks, bucket bookkeeping)
  rccl       the real thing: ~16 async all_reduce calls per step on RCCL's stream, event waits at finish_step
For each: GPU time per step, and the time after which the host has ENQUEUED a step (idle queue).
usage: python tools/dp_one_gpu_cost.py [--steps N]"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _Done:
    def wait(self):
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29300 + os.getpid() % 100))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True            # as bench.py creates it
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
    import bench
    import lc2is_amd.nn as N
    from lc2is_amd.dp import GradReducer
    from lc2is_amd.step import TrainStep
    inputs, labels = bench.synth_batch(32, 512, 128, 16, 2, dev)

    def make(reducer):
        torch.manual_seed(1024)
        m = N.BaseModelWithText(patch_size=16, in_size=512, out_size=128).to(dev).train()
        return TrainStep(m, optimizer="sgd", lr=1e-5, reducer=reducer)

    real_all_reduce = dist.all_reduce
    variants = {"plain": make(None), "groups": make(GradReducer()), "rccl": make(GradReducer())}

    def run(name, n):
        dist.all_reduce = (lambda t, *x, **k: _Done()) if name == "groups" else real_all_reduce
        ts = variants[name]
        for _ in range(n):
            ts.step(inputs, labels)

    for name in variants:
        run(name, 3)
    torch.cuda.synchronize()
    res = {k: [] for k in variants}
    enq = {k: [] for k in variants}
    for _ in range(a.rounds):
        for name in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(name, a.steps)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(name, 1)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            enq[name].append((t1 - t0) * 1e3)
    dist.all_reduce = real_all_reduce
    base = min(res["plain"])
    for name in variants:
        r = variants[name].reducer
        extra = f", {r.collectives_last_step} all_reduce calls per step" if r is not None else ""
        print(f"{name:7s} {min(res[name]):7.2f} ms/step ({32e3 / min(res[name]):7.1f} img/s, {100 * (min(res[name]) / base - 1):+5.1f} % vs plain); "
              f"one step enqueued after {min(enq[name]):6.2f} ms{extra}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
