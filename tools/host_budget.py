"""Host budget of the eager train step: how long the host needs to ENQUEUE one step (Python + ctypes + HIP launch
calls) against how long the GPU needs to run it.  The data-parallel step stays eager (TrainStep.capture refuses a reducer), so the
host has to stay ahead of the GPU by itself; this prints the margin on one GPU.  The number of launches per step comes from the
kernel trace (profiles/r04_summary.txt).
usage: python tools/host_budget.py [--steps N] [--batch B]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
import lc2is_amd.nn as N  # noqa: E402
from lc2is_amd.step import TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1024)
    model = N.BaseModelWithText(patch_size=16, in_size=512, out_size=128).to(dev).train()
    ts = TrainStep(model, optimizer="sgd", lr=1e-5)
    inputs, labels = bench.synth_batch(a.batch, 512, 128, 16, 2, dev)
    for _ in range(5):
        ts.step(inputs, labels)
    torch.cuda.synchronize()

    t_host, t_all = [], []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ts.step(inputs, labels)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_host.append((t1 - t0) / a.steps * 1e3)
        t_all.append((t2 - t0) / a.steps * 1e3)
    print(f"eager step, B={a.batch}, {a.steps} steps back to back: the host returns from step() after {min(t_host):.2f} ms per step "
          f"(enqueue: Python + ctypes + launch calls), the GPU finishes after {min(t_all):.2f} ms per step")
    # one step from an idle queue: the enqueue time without back-pressure from a full launch queue
    t_one = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts.step(inputs, labels)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_one.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    e, d = min(t_one)
    print(f"one step from an idle queue: enqueued after {e:.2f} ms, done after {d:.2f} ms -> the host needs {e / d:.2f} of the GPU's time; "
          f"a reducer adds ~18 all_reduce calls and the bucket bookkeeping per step on the host")


if __name__ == "__main__":
    main()
