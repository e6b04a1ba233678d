"""GPU parity of the multi-scale decoders (BASELINE config 5) against the REFERENCE-produced vectors
(tests/golden/hier_tiny.pt) and, for FTNDecoder (nhead fixed to 8 -> needs dim 512), against the CPU oracle."""
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


def _load_named(module, weights):
    named = dict(module.named_parameters())
    assert set(named) == set(weights), (set(named) ^ set(weights))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(weights[k])


@pytest.mark.parametrize("name", ["cross", "selfa"])
def test_hierarchical_vs_reference(dev, name):
    import lc2is_amd.nn as N
    from golden_util import make_weights
    fx = torch.load(G / "hier_tiny.pt", weights_only=True)
    c = fx[name]
    in_dims, dim = [64, 128, 192, 256], 128
    if name == "cross":
        m = N.HierarchicalCrossA(in_dims, [2, 1, 1], dim, nhead=2, dropout=0, batch_first=True)
    else:
        m = N.HierarchicalSelfA(in_dims, [1, 1, 2], dim, nhead=2, dropout=0, batch_first=True)
    _load_named(m, make_weights({k: v.tolist() for k, v in c["shapes"].items()}, int(c["seed"])))
    m = m.to(dev).train()
    vis = [fx[f"visual{i}"].to(dev).requires_grad_(True) for i in range(4)]
    txt = fx["textual"].to(dev).requires_grad_(True)
    out = m(vis, txt) if name == "cross" else m(vis)
    assert out.shape == c["out"].shape
    r = _rel(out, c["out"])
    assert r < 1.5e-2, r
    out.backward(c["dout"].to(dev))
    assert _rel(vis[0].grad, c["dvisual0"]) < 3e-2
    assert _rel(vis[3].grad, c["dvisual3"]) < 5e-2
    assert vis[1].grad is None and vis[2].grad is None
    if name == "cross":
        assert _rel(txt.grad, c["dtextual"]) < 5e-2
    named = dict(m.named_parameters())
    worst = 0.0
    for k, g in c["grad_full"].items():
        worst = max(worst, _rel(named[k].grad, g))
    assert worst < 8e-2, worst
    for k, st in c["grad_stats"].items():
        g = named[k].grad
        assert g is not None, k
        ref_abs = float(st[1])
        if ref_abs < 1e-6 * g.numel():
            assert float(g.abs().mean()) < 1e-3, k
            continue
        assert abs(float(g.abs().sum()) - ref_abs) < 0.1 * ref_abs, (k, float(g.abs().sum()), ref_abs)


def test_ftn_decoder_and_blocks_vs_oracle(dev):
    """FTNDecoder hard-codes 8 heads (model/decoder.py:49): dim 512 -> head_dim 64; small grids keep the CPU oracle fast."""
    import lc2is_amd.nn as N
    from golden_util import make_weights
    from oracle import ref_cpu as O
    in_dims, dim = [96, 192, 384, 768], 512       # Swin-small widths: stage 1 exercises the zero-padded K path
    m = N.FTNDecoder(in_dims, dim, dropout=0)
    shapes = {k: list(v.shape) for k, v in m.named_parameters()}
    w = make_weights(shapes, 5)
    _load_named(m, w)
    g = torch.Generator().manual_seed(3)
    B, K = 1, 12
    visual = [torch.randn(B, p, c, generator=g) for p, c in zip((256, 64, 16, 4), in_dims)]
    textual = torch.randn(B, K, dim, generator=g)
    vis_r = [v.clone().requires_grad_(True) for v in visual]
    txt_r = textual.clone().requires_grad_(True)
    ref = O.hierarchical(w, "", vis_r, txt_r, nhead=8, depth=(1, 1, 1), layer_key="attention_block.")
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    m = m.to(dev).train()
    vis = [v.to(dev).requires_grad_(True) for v in visual]
    txt = textual.to(dev).requires_grad_(True)
    out = m(vis, txt)
    assert _rel(out, ref) < 1.5e-2
    out.backward(dout.to(dev))
    assert _rel(vis[0].grad, vis_r[0].grad) < 3e-2 and _rel(vis[3].grad, vis_r[3].grad) < 6e-2
    assert _rel(txt.grad, txt_r.grad) < 6e-2
    # a block on its own (CrossABlock.forward), shared weights applied twice
    blk = N.CrossABlock(N.SRTransformerCrossA(d_model=128, nhead=2, sr_ratio=2, dropout=0, batch_first=True), depth=2)
    wb = make_weights({k: list(v.shape) for k, v in blk.named_parameters()}, 9)
    _load_named(blk, wb)
    x = torch.randn(2, 64, 128, generator=g)
    mem = torch.randn(2, 5, 128, generator=g)
    refb = O.attn_block(wb, "", x, mem, nhead=2, depth=2, layer_key="layers.0.")
    blk = blk.to(dev).eval()
    with torch.no_grad():
        outb = blk(x.to(dev), mem.to(dev))
    assert outb.shape == (2, 256, 128) and _rel(outb, refb) < 1.5e-2


def test_score_map_tail_vs_reference(dev):
    import lc2is_amd.nn as N
    t = torch.load(G / "hier_tiny.pt", weights_only=True)["tail"]
    tail = N.ScoreMapTail(4)
    ve = t["ve"].to(dev).requires_grad_(True)
    te = t["te"].to(dev).requires_grad_(True)
    sm = tail(ve, te)
    assert sm.shape == t["score"].shape and (sm.cpu() - t["score"]).abs().max().item() < 1e-2   # cosine scores in [-1,1]
    loss = tail.loss(ve, te, t["labels"].to(dev))
    assert abs(loss.item() - t["loss"].item()) < 2e-3
    loss.backward()
    assert _rel(ve.grad, t["dve"]) < 3e-2 and _rel(te.grad, t["dte"]) < 3e-2
    # unfused: materialised map -> CrossEntropyLoss -> backward gives the same gradients
    ve2 = t["ve"].to(dev).requires_grad_(True)
    te2 = t["te"].to(dev).requires_grad_(True)
    l2 = N.CrossEntropyLoss()(tail(ve2, te2), t["labels"].to(dev))
    l2.backward()
    assert abs(l2.item() - loss.item()) < 1e-4 and _rel(ve2.grad, ve.grad) < 1e-2 and _rel(te2.grad, te.grad) < 1e-2
