"""GPU parity of the Swin backbone against vectors produced by the REFERENCE's own
SwinTransformer.forward (tests/golden/swin_tiny.pt, tools/make_golden.py swin) and, per kernel, against plain torch."""
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


def test_rows_gather(dev):
    from lc2is_amd import ops
    g = torch.Generator().manual_seed(0)
    src = torch.randn(37, 24, generator=g).to(dev)
    idx = torch.randint(-1, 37, (53,), generator=g).to(torch.int32).to(dev)
    add = torch.randn(53, 24, generator=g).to(dev)
    ref = torch.where(idx[:, None] >= 0, src[idx.clamp_min(0).long()], torch.zeros(1, device=dev)) + add
    out = ops.rows_gather(src, idx, add=add)
    assert torch.equal(out, ref)
    out16 = ops.rows_gather(src.bfloat16(), idx, out_dtype=torch.bfloat16)
    assert torch.equal(out16, torch.where(idx[:, None] >= 0, src.bfloat16()[idx.clamp_min(0).long()], torch.zeros(1, device=dev, dtype=torch.bfloat16)))
    wide = torch.zeros(53, 32, dtype=torch.bfloat16, device=dev)                 # strided destination, fp32 -> bf16
    ops.rows_gather(src, idx, out=wide, cols=24)
    assert torch.equal(wide[:, :24], (ref - add).bfloat16()) and float(wide[:, 24:].abs().max()) == 0.0


@pytest.mark.parametrize("ws,shift,nH", [(7, 0, 3), (7, 3, 3), (5, 2, 2), (4, 2, 1)])
def test_swin_window_attention(dev, ws, shift, nH):
    """Window attention kernel vs torch: relative-position bias, cyclic-shift region mask, forward + backward."""
    from lc2is_amd import ops
    g = torch.Generator().manual_seed(ws * 10 + shift)
    B, nwy, nwx = 2, 3, 2
    S, C, D = ws * ws, 32 * nH, 32
    Hp, Wp = nwy * ws, nwx * ws
    nwin = B * nwy * nwx
    qkv = (torch.randn(nwin * S, 3 * C, generator=g) * 0.7).bfloat16()
    bias = torch.randn(nH, S, S, generator=g) * 0.5
    do = (torch.randn(nwin * S, C, generator=g) * 0.3).bfloat16()
    # torch reference
    q, k, v = [t.float().view(nwin, S, nH, D).transpose(1, 2).double().requires_grad_(True) for t in qkv.split(C, dim=1)]
    bref = bias.double().requires_grad_(True)
    logits = q @ k.transpose(-1, -2) * D ** -0.5 + bref[None]
    if shift > 0:
        hr = (torch.arange(Hp) >= Hp - ws).long() + (torch.arange(Hp) >= Hp - shift).long()
        wr = (torch.arange(Wp) >= Wp - ws).long() + (torch.arange(Wp) >= Wp - shift).long()
        img = (hr[:, None] * 3 + wr[None, :]).double()
        mw = img.view(nwy, ws, nwx, ws).transpose(1, 2).reshape(-1, S)
        am = mw[:, None, :] - mw[:, :, None]
        am = torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))
        logits = (logits.view(B, nwy * nwx, nH, S, S) + am[None, :, None]).view(nwin, nH, S, S)
    oref = (torch.softmax(logits, -1) @ v).transpose(1, 2).reshape(nwin * S, C)
    oref.backward(do.double())
    dqkv_ref = torch.cat([t.grad.transpose(1, 2).reshape(nwin * S, C) for t in (q, k, v)], dim=1)
    # HIP
    qd, bd, dod = qkv.to(dev), bias.to(dev), do.to(dev)
    o, lse = ops.swin_attn_fwd(qd, bd, nwin, nwy * nwx, nwx, Hp, Wp, ws, shift, nH, D ** -0.5)
    assert _rel(o.float(), oref.detach()) < 6e-3
    dbias = torch.full((nH, S, S), 3.0, device=dev)
    dqkv = ops.swin_attn_bwd(qd, o, dod, lse, bd, nwin, nwy * nwx, nwx, Hp, Wp, ws, shift, nH, D ** -0.5, dbias=dbias)
    assert _rel(dqkv.float(), dqkv_ref) < 1.5e-2
    assert _rel(dbias, bref.grad) < 1.5e-2
    dbias2 = dbias.clone()
    dqkv2 = ops.swin_attn_bwd(qd, o, dod, lse, bd, nwin, nwy * nwx, nwx, Hp, Wp, ws, shift, nH, D ** -0.5, dbias=dbias2,
                              accumulate_dbias=True)
    assert torch.equal(dqkv2, dqkv) and _rel(dbias2, 2 * bref.grad) < 1.5e-2      # reproducible; accumulate flag


def test_swin_vs_reference(dev):
    from golden_util import make_weights
    from lc2is_amd.nn.swin import SwinArch, SwinTransformer
    fx = torch.load(G / "swin_tiny.pt", weights_only=True)
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    m = SwinTransformer(SwinArch(32, (2, 2, 2, 2), (1, 2, 4, 8), 5), drop_path_rate=0.0)
    named = dict(m.named_parameters())
    assert {k: list(v.shape) for k, v in named.items()} == shapes          # the reference's (transformers') names and shapes
    w = make_weights(shapes, int(fx["wseed"]))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    m = m.to(dev).train()
    outs = m(fx["pixel_values"].to(dev))
    assert len(outs) == 4
    for i, (o, r) in enumerate(zip(outs, fx["outs"])):
        assert o.shape == r.shape
        assert _rel(o, r) < 1.5e-2, (i, _rel(o, r))
    sum((o * d.to(dev)).sum() for o, d in zip(outs, fx["douts"])).backward()
    names = list(shapes)
    unused = {names[int(i)] for i in fx["no_grad"]}
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for k, g in fx["grad_full"].items():
        if float(g.abs().max()) < 1e-5:                                    # key biases: mathematically zero gradient;
            qg = fx["grad_full"][k.replace("k_proj", "q_proj")]              # here the bf16 round-off of ~4k summed dK rows
            assert float(named[k].grad.abs().max()) < 0.05 * float(qg.abs().max()) + 1e-3, k
            continue
        r = _rel(named[k].grad, g)
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] < 8e-2, worst
    for k, st in fx["grad_stats"].items():
        g = named[k].grad
        if k in unused:
            assert g is None, k
            continue
        assert g is not None, k
        ref_abs = float(st[1])
        if ref_abs < 1e-6 * g.numel():                                         # zero by construction (key biases): round-off only,
            scale = float(fx["grad_stats"][k.replace("k_proj", "q_proj")][1]) / g.numel()   # judged against the query bias
            assert float(g.abs().mean()) < 0.05 * scale + 1e-3, (k, float(g.abs().mean()), scale)
            continue
        assert abs(float(g.abs().sum()) - ref_abs) < 0.1 * ref_abs, (k, float(g.abs().sum()), ref_abs)


def test_swin_legacy_keys_and_drop_path(dev):
    from lc2is_amd.nn.swin import SwinArch, SwinTransformer
    arch = SwinArch(32, (2, 2, 2, 2), (1, 2, 4, 8), 5)
    m = SwinTransformer(arch, drop_path_rate=0.0)
    sd = m.state_dict()
    legacy = {}
    for k, v in sd.items():                                                # transformers-4.x checkpoint names
        k = k.replace(".attention.q_proj.", ".attention.self.query.").replace(".attention.k_proj.", ".attention.self.key.")
        k = k.replace(".attention.v_proj.", ".attention.self.value.").replace(".attention.o_proj.", ".attention.output.dense.")
        k = k.replace(".attention.relative_position_bias.relative_position_bias_table", ".attention.self.relative_position_bias_table")
        k = k.replace(".mlp.fc1.", ".intermediate.dense.").replace(".mlp.fc2.", ".output.dense.")
        legacy[k] = v.clone() + 1.0
    legacy["encoder.encoder.layers.0.blocks.0.attention.self.relative_position_index"] = torch.zeros(25, 25, dtype=torch.long)
    m2 = SwinTransformer(arch, drop_path_rate=0.0)
    m2.load_state_dict(legacy)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k] + 1.0), k
    m3 = SwinTransformer(arch).to(dev).train()                              # reference default drop_path_rate 0.1: runs in
    assert len(m3(torch.randn(1, 3, 176, 176, device=dev))) == 4            # training mode (tests/test_gpu_dropout.py checks it)
    m3.eval()
    assert len(m3(torch.randn(1, 3, 176, 176, device=dev))) == 4
