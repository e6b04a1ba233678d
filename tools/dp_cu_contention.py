"""What a kernel of another queue that HOLDS compute units costs the step, and whether a CU budget for the tile planners pays (DESIGN §5,
decision 2b).  One-GPU stand-in for RCCL's channel kernels under data parallelism: tools/probes/cu_hog/cu_hog.hip — n workgroups that spin
for the duration of a step on a side stream (bounded: they exit after the requested time) — started before every step; the step is timed
by HIP events on the main stream with the planners counting on all 256 CUs (budget 0) or on 256 - n.
usage: python tools/dp_cu_contention.py [--steps N]"""
import argparse
import ctypes
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
import lc2is_amd.nn as N  # noqa: E402
from lc2is_amd import ops  # noqa: E402
from lc2is_amd.step import TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    src = ROOT / "tools" / "probes" / "cu_hog" / "cu_hog.hip"
    so = src.with_suffix(".so")
    if not so.exists():
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    hog = ctypes.CDLL(str(so))
    hog.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1024)
    model = N.BaseModelWithText(patch_size=16, in_size=512, out_size=128).to(dev).train()
    ts = TrainStep(model, optimizer="sgd", lr=1e-5)
    inputs, labels = bench.synth_batch(32, 512, 128, 16, 2, dev)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(5):
        ts.step(inputs, labels)
    torch.cuda.synchronize()

    def run(nhog, budget):
        ops.set_cu_budget(budget)
        times = []
        for i in range(a.steps + 2):
            torch.cuda.synchronize()
            if nhog:
                rc = hog.cu_hog_launch(nhog, 38000, sink.data_ptr(), side.cuda_stream)   # ~ one step long
                assert rc == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ts.step(inputs, labels)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                times.append(e0.elapsed_time(e1))
        ops.set_cu_budget(0)
        times.sort()
        return times[len(times) // 2]

    base = run(0, 0)
    print(f"no other kernel, budget 0 (all 256 CUs):              {base:6.2f} ms/step", flush=True)
    for nhog in (8, 16, 32):
        t0 = run(nhog, 0)
        t1 = run(nhog, 256 - nhog)
        print(f"{nhog:3d} CUs held by another queue: planners on 256 CUs {t0:6.2f} ms/step ({t0 / base - 1:+.1%}), "
              f"on {256 - nhog} CUs {t1:6.2f} ms/step ({t1 / base - 1:+.1%})", flush=True)


if __name__ == "__main__":
    main()
