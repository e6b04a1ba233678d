#!/usr/bin/env python3
"""The ViT image encoder ALONE, forward + backward, on the headline shapes (B = 32, 512x512): run under
`rocprofv3 --kernel-trace` its dispatches give the set of (kernel, grid, block) signatures that belong to the encoder, which
tools/roofline_sum.py then looks up in the trace of the FULL bench.py step (the encoder-only MFMA fraction is
defined on rocprof kernel time of the encoder's kernels).  Usage: python3 tools/encoder_only.py [--steps 3 --warmup 2]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--in-size", type=int, default=512)
    args = ap.parse_args()
    import lc2is_amd.nn as N
    dev = torch.device("cuda:0")
    torch.manual_seed(1024)
    enc = N.ImageEncoderCLIP(args.in_size, 16).to(dev).train()
    pix = torch.randn(args.batch, 3, args.in_size, args.in_size, device=dev)
    dout = None
    for _ in range(args.warmup + args.steps):
        for p in enc.parameters():
            p.grad = None
        out = enc(pix)
        if dout is None:
            dout = torch.randn_like(out) * 0.01
        out.backward(dout)
    torch.cuda.synchronize()
    print("encoder_only: done", tuple(out.shape))


if __name__ == "__main__":
    main()
