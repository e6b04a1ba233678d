"""GPU parity of the older FTN pyramid (ftn.Decoder / ftn.Transformer, model/ftn.py:67-157) against
the REFERENCE-produced vectors in tests/golden/ftn_decoder.pt, plus the standalone Transformer against the CPU oracle."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
G = HERE / "golden"


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_ftn_decoder_vs_reference(dev):
    from golden_util import ftn_inputs, make_weights
    from lc2is_amd.nn.ftn import Decoder
    fx = torch.load(G / "ftn_decoder.pt", weights_only=True)
    st = int(fx["stride"])
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    m = Decoder()
    named = dict(m.named_parameters())
    assert {k: list(v.shape) for k, v in named.items()} == shapes          # the reference's parameter names and shapes
    w = make_weights(shapes, int(fx["wseed"]))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    m = m.to(dev).eval()                                                    # reference vectors are eval-mode (dropout 0.1 hard-coded)
    xs, dout = ftn_inputs(int(fx["xseed"]))
    xs = [x.to(dev).requires_grad_(True) for x in xs]
    out = m(xs)
    assert out.shape == (1, 16384, 512)
    r = _rel(out[:, ::st], fx["out"])
    assert r < 1.5e-2, r
    out.backward(dout.to(dev))
    for i in range(4):
        r = _rel(xs[i].grad[:, ::st], fx["dx"][i])
        assert r < 5e-2, (i, r)
    names = list(shapes)
    unused = {names[int(i)] for i in fx["no_grad"]}
    named = dict(m.named_parameters())
    worst = 0.0
    for k, g in fx["grad_full"].items():
        worst = max(worst, _rel(named[k].grad, g))
    assert worst < 8e-2, worst
    for k, stat in fx["grad_stats"].items():
        g = named[k].grad
        if k in unused:
            assert g is None, k
            continue
        assert g is not None, k
        ref_abs = float(stat[1])
        if ref_abs < 1e-6 * g.numel():
            assert float(g.abs().mean()) < 1e-3, k
            continue
        assert abs(float(g.abs().sum()) - ref_abs) < 0.1 * ref_abs, (k, float(g.abs().sum()), ref_abs)
    # training mode with the reference's hard-coded dropout 0.1 (model/ftn.py:135) runs and IS stochastic
    m.train()
    with torch.no_grad():
        o1, o2 = m([x.detach() for x in xs]), m([x.detach() for x in xs])
    assert torch.isfinite(o1).all() and _rel(o1, o2) > 1e-3 and _rel(o1[:, ::st], fx["out"]) > 1e-2


@pytest.mark.parametrize("sr_ratio,repeat,upsample", [(2, 2, True), (1, 1, False)])
def test_ftn_transformer_vs_oracle(dev, sr_ratio, repeat, upsample):
    """Standalone ftn.Transformer incl. the non-square second upsample (grid height stays h) and the sr_ratio=1 form."""
    from golden_util import make_weights
    from lc2is_amd.nn.ftn import Transformer
    from oracle import ref_cpu as O
    m = Transformer(repeat=repeat, upsample=upsample, sr_ratio=sr_ratio, dim=512, nhead=8, dropout=0.0)
    shapes = {k: list(v.shape) for k, v in m.named_parameters()}
    w = make_weights(shapes, 77)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(w[k])
    g = torch.Generator().manual_seed(9)
    B, h = 2, 8
    x = torch.randn(B, h * h, 512, generator=g)
    sd = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    xr = x.clone().requires_grad_(True)
    ref = O.ftn_transformer(sd, "", xr, h, repeat, sr_ratio, upsample, 8)
    dout = torch.randn(ref.shape, generator=g) * 0.1
    ref.backward(dout)
    m = m.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    out = m(xd, h)
    assert out.shape == ref.shape
    assert _rel(out, ref.detach()) < 1.5e-2
    out.backward(dout.to(dev))
    assert _rel(xd.grad, xr.grad) < 5e-2
    for k, p in m.named_parameters():
        gr = sd[k].grad
        if sr_ratio == 1 and (k.startswith("sr.") or k.startswith("norm.")):
            assert p.grad is None and gr is None                            # conv/norm unused when sr_ratio == 1
            continue
        if float(gr.abs().sum()) < 1e-6 * gr.numel():
            assert float(p.grad.abs().mean()) < 1e-3, k
            continue
        assert _rel(p.grad, gr) < 8e-2, (k, _rel(p.grad, gr))
