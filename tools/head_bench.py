"""Fused upsample + cross-entropy head kernel alone (lc2is_head_upsample_ce) at the headline shape (32 x 32x32 -> 128x128,
151 classes, bicubic x4), at config 5's (8 x 128x128 -> 512x512, 150 classes, bilinear x4) and at AuxiliaryLoss shapes (x16, x8):
time per launch (HIP events).
  python tools/head_bench.py [--iters 20]     LC2IS_LIB=<other build> for an A/B in one call"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lc2is_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, B, h, C, mode, S in (("config 2: bicubic x4, 151 classes, 32 x 128x128", 32, 32, 151, ops.INTERP_BICUBIC, 4),
                                   ("config 5: bilinear x4, 150 classes, 8 x 512x512", 8, 128, 150, ops.INTERP_BILINEAR, 4),
                                   ("AuxiliaryLoss: bilinear x16, 150 classes, 8 x 32x32 -> 512x512", 8, 32, 150, ops.INTERP_BILINEAR, 16),
                                   ("AuxiliaryLoss: bilinear x8, 150 classes, 8 x 64x64 -> 512x512", 8, 64, 150, ops.INTERP_BILINEAR, 8)):
        g = torch.Generator(device="cpu").manual_seed(1)
        lo = torch.zeros(B * h * h, 192)
        lo[:, :C] = torch.randn(B * h * h, C, generator=g) * 3
        lo = lo.to(dev)
        labels = torch.randint(0, C, (B, S * h, S * h), generator=g).to(dev)
        for _ in range(3):
            ops.head_upsample_ce(lo, labels, B, h, h, C, S, mode, want_grad=True, grad_scale=1.0 / labels.numel())
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            loss = torch.zeros(2, device=dev); dlo = torch.zeros_like(lo)
            nbytes = ops._fn("lc2is_head_upsample_ce_workspace_bytes")(B, h, h, C, S, mode, 1)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()   # both launches: tile kernel + fixed-order finish
            rc = ops._fn("lc2is_head_upsample_ce")(ops._ptr(lo), 192, ops._ptr(labels), ops._ptr(dlo), None, ops._ptr(loss), B, h, h, C, S, mode,
                                                   -100, 1.0 / labels.numel(), ops._ptr(ws), nbytes, ops._stream())
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"{name}: best {ts[0]:.1f} us  median {ts[len(ts) // 2]:.1f} us  loss {loss[0].item() / max(loss[1].item(), 1.0):.6f}", flush=True)


if __name__ == "__main__":
    main()
