"""Deterministic weights shared by tools/make_golden.py (which loads them into the REFERENCE modules) and the
tests (which load them into the oracle / the HIP modules): fixtures then only need inputs and outputs."""
import torch


def make_weights(shapes: dict, seed: int) -> dict:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if "norm" in name and name.endswith("weight"):
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=g) * (max(fan_in, 1) ** -0.5)
    return out


def ftn_inputs(seed: int, B: int = 1):
    """Stage tensors of ftn.Decoder's hard-coded geometry (model/ftn.py:70,104) and an output gradient."""
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, h * h, c, generator=g) for h, c in zip((128, 64, 32, 16), (128, 256, 512, 1024))]
    dout = torch.randn(B, 16384, 512, generator=g) * 0.1
    return xs, dout
