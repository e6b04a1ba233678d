"""The vision tower's whole-tower weight-gradient launch (12 layers x {fc2, fc1, out-proj, q, k, v} = 72 problems, 1296 tiles
of 256x256 over 32 800 tokens) alone, on distinct operand buffers per layer as in the step.
  python tools/tn_tower.py [--iters 5] [--layers 12] [--check]
Run it under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` for the launch's L2-miss reads; LC2IS_TN_LOCKSTEP=0/1 is the A/B."""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lc2is_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, C, F = 32800, 768, 3072
    g = torch.Generator(device=dev).manual_seed(3)

    def rnd(rows, cols):
        return torch.randn(rows, cols, device=dev, generator=g, dtype=torch.float32).bfloat16()

    problems = []
    for _ in range(a.layers):
        g16, act, dz, h2, gm16, o, dqkv, h = rnd(M, C), rnd(M, F), rnd(M, F), rnd(M, C), rnd(M, C), rnd(M, C), rnd(M, 3 * C), rnd(M, C)
        problems.append((g16, act, torch.empty(C, F, device=dev), torch.empty(C, device=dev), False))
        problems.append((dz, h2, torch.empty(F, C, device=dev), torch.empty(F, device=dev), False))
        problems.append((gm16, o, torch.empty(C, C, device=dev), torch.empty(C, device=dev), False))
        for j in range(3):
            problems.append((dqkv[:, j * C:(j + 1) * C], h, torch.empty(C, C, device=dev), torch.empty(C, device=dev), False))
    torch.cuda.synchronize()
    ops.gemm_tn_grouped(problems)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_tn_grouped(problems)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    flops = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in problems)
    best = min(ts)
    print(f"lockstep={os.environ.get('LC2IS_TN_LOCKSTEP', '0')} layers={a.layers}: best {best * 1e3:.0f} us  median {sorted(ts)[len(ts) // 2] * 1e3:.0f} us"
          f"  {flops / best / 1e9:.0f} TF/s  (launch + reduce; {len(problems)} problems)", flush=True)
    if a.check:
        worst = 0.0
        for dy, x, dw, db, _ in problems[:6] + problems[-6:]:
            ref = dy.float().T @ x.float()
            worst = max(worst, ((dw - ref).norm() / ref.norm()).item(), ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item())
        print(f"max relative error vs fp32 matmul over 12 problems: {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
