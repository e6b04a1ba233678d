"""GPU parity of the composed drop-ins PromptFTN and DenseClip (lc2is_amd/nn/compose.py) END TO END against vectors produced by
the REFERENCE's own ``forward``s (model/model.py:106-214; tools/make_golden.py make_prompt_ftn / make_dense_clip — only the
hub-naming ``__init__``s were bypassed): config 5's model as the reference composes it, on a 512 x 512 image."""
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


def _load(m, fx):
    from golden_util import make_weights
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes      # the reference's parameter names and shapes
    res = m.load_state_dict(make_weights(shapes, int(fx["wseed"])), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m


def test_prompt_ftn_end_to_end_vs_reference(dev):
    import lc2is_amd.nn as N
    from golden_util import prompt_ftn_inputs
    fx = torch.load(G / "prompt_ftn.pt", weights_only=True)
    m = N.PromptFTN(swin_arch=N.SwinArch(32, (1, 1, 1, 1), (1, 2, 4, 8), 7, drop_path_rate=0.0),
                    text_arch=N.ClipArch(512, 8, 1, 128, vocab=512, max_pos=16, eos_token_id=511), prompt_layers=2, dropout=0.0)
    m = _load(m, fx).to(dev).train()
    assert all(not p.requires_grad for p in m.textual_encoder.parameters())     # frozen like model/model.py:178-180
    inputs, labels = prompt_ftn_inputs(int(fx["iseed"]))
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    with torch.no_grad():
        none, score_map = m(dinputs)
    assert none is None and score_map.shape == (1, 6, 512, 512)
    err = (score_map[:, :, ::8, ::8].cpu() - fx["score_s8"]).abs().max().item()
    assert err < 5e-3, err                                                       # cosine scores in [-1, 1]
    loss = m.forward_loss(dinputs, labels.to(dev))
    assert abs(loss.item() - float(fx["loss"])) < 2e-3
    loss.backward()
    named = dict(m.named_parameters())
    for k in fx["no_grad"]:
        assert named[k].grad is None, k
    worst = 0.0
    for k, g in fx["grad_full"].items():
        mine = named[k].grad
        mine = mine if mine.numel() == g.numel() else mine.flatten()[::37]       # big tensors: every 37th element was kept
        r = _rel(mine, g)
        worst = max(worst, r)
        assert r < 8e-2, (k, r)
    for k, st in fx["grad_stats"].items():
        g = named[k].grad
        assert g is not None, k
        ref_abs = float(st[1])
        if ref_abs > 1e-6 * g.numel():
            assert abs(float(g.abs().sum()) - ref_abs) < 0.12 * ref_abs, (k, float(g.abs().sum()), ref_abs)


def test_dense_clip_end_to_end_vs_reference(dev):
    import lc2is_amd.nn as N
    from golden_util import dense_clip_inputs
    fx = torch.load(G / "dense_clip_tiny.pt", weights_only=True)
    m = N.DenseClip(16, 64, 16, vision_arch=N.ClipArch(256, 4, 2, 256),
                    text_arch=N.ClipArch(64, 1, 1, 128, vocab=512, max_pos=16, eos_token_id=511), num_layers=2, dim=128, nhead=2,
                    prompt_dropout=0.0, dim_feedforward=128)
    m = _load(m, fx).to(dev).train()
    inputs, ds, do = dense_clip_inputs(int(fx["iseed"]))
    none, score_map, out = m({k: v.to(dev) for k, v in inputs.items()})
    assert none is None and score_map.shape == (2, 5, 4, 4) and out.shape == (2, 17, 256)
    assert (score_map.detach().cpu() - fx["score_map"]).abs().max().item() < 1e-2
    assert _rel(out.detach(), fx["out"]) < 1.5e-2
    ((score_map * ds.to(dev)).sum() + (out * do.to(dev)).sum()).backward()
    named = dict(m.named_parameters())
    for k in fx["no_grad"]:
        assert named[k].grad is None, k
    for k, g in fx["grad_full"].items():
        if k.endswith("multihead_attn.in_proj_weight") and k not in named:       # packed key of the d_kv == d_model layers
            pre = k[: -len("in_proj_weight")]
            mine = torch.cat([named[pre + n].grad for n in ("q_proj_weight", "k_proj_weight", "v_proj_weight")], 0)
        else:
            mine = named[k].grad
        r = _rel(mine, g)
        assert r < 8e-2, (k, r)
