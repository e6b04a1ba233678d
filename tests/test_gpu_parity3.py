"""Round-3 parity gates on the GPU (all through the C ABI):

* the relu-flip claim of round 2 turned into a measurement: the oracle is re-run with the 0/1 activation pattern the HIP
  path actually used (taken from its saved relu output) in place of its own sign test; the gradient error of
  PromptDecoder (a7) and of the config-2 decoder's linear1 must collapse to the class of the smooth layers;
* config 5 at its REAL shapes: HierarchicalCrossA([96,192,384,768], [1,1,1], 512, 8) on Swin-small stage tensors of a
  512x512 image (16384 / 4096 / 1024 / 256 tokens), K = 150 text queries, score-map tail (bilinear x4) + CE at 512x512:
  forward, loss and gradients against the CPU oracle;
* config 4 deeper than one layer: ViT-L/14 @640x640 (2026 tokens), depth 4, text width 768, decoder head_dim 128,
  with the gradient list of test_config2_full_depth_vs_oracle;
* the mIoU edge case of metrics.py:94-97 (an image whose label holds only ignore_index gives NaN, like the reference).

Measured values land in gpurun_out/parity_r04.json (copied to profiles/r04_parity.json; round 3's: profiles/r03_parity.json).
"""
import json
import math
import os
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
G = HERE / "golden"
REPORT = {}


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _note(k, v):
    REPORT[k] = v
    out = Path(os.environ.get("GRAFT_REPO_ROOT", ".")) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "parity_r04.json").write_text(json.dumps(REPORT, indent=1))
    except OSError:
        pass


class _CaptureRelu:
    """Records, per decoder layer call, the relu output the HIP path saved for its backward (`a` = bf16(relu(z)), whose sign
    IS the mask lc2is_gemm_nt's DRELU epilogue applies)."""

    def __init__(self):
        import lc2is_amd.nn.decoder as D
        self.D, self.orig, self.acts = D, D._layer_fwd, []

    def __enter__(self):
        def wrapped(*args, **kw):
            x3, sv = self.orig(*args, **kw)
            if sv is not None:
                self.acts.append(sv["a"])
            return x3, sv
        self.D._layer_fwd = wrapped
        return self

    def __exit__(self, *exc):
        self.D._layer_fwd = self.orig


# ---------------------------------------------------------------------------------------------------------------------
# a7: is the 3.4e-2 gradient error the relu mask?  
# ---------------------------------------------------------------------------------------------------------------------
def test_prompt_decoder_gradient_error_is_the_relu_mask(dev):
    import lc2is_amd.nn as N
    from golden_util import make_weights, prompt_inputs
    from oracle import ref_cpu as O
    fx = torch.load(G / "prompt_decoder.pt", weights_only=True)
    dec = N.PromptDecoder(N.PromptLayer(d_model=512, d_kv=1024, nhead=8, batch_first=True, dropout=0.0), num_layers=2)
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    w = make_weights(shapes, int(fx["wseed"]))
    named = dict(dec.named_parameters())
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    dec = dec.to(dev).train()
    tgt, mem, dout = prompt_inputs(int(fx["iseed"]))
    B, K, F = tgt.shape[0], tgt.shape[1], 2048
    t = tgt.to(dev).requires_grad_(True)
    m = mem.to(dev).requires_grad_(True)
    with _CaptureRelu() as cap:
        out = dec(tgt=t, memory=m)
    out.backward(dout.to(dev))
    assert len(cap.acts) == 2
    masks = [(a > 0).float().cpu().view(B, K, F) for a in cap.acts]

    def oracle(drops):
        params = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        tt, mm = tgt.clone().requires_grad_(True), mem.clone().requires_grad_(True)
        o = O.decoder_block(params, "", tt, mm, 8, 2, norm_first=False, drops=drops)
        o.backward(dout)
        return o.detach(), tt.grad, mm.grad, {k: p.grad for k, p in params.items()}

    o_own, dt_own, dm_own, g_own = oracle(None)
    o_fed, dt_fed, dm_fed, g_fed = oracle([{"relu_mask": mk} for mk in masks])
    own = dict(dtgt=_rel(t.grad, dt_own), dmem=_rel(m.grad, dm_own))
    fed = dict(dtgt=_rel(t.grad, dt_fed), dmem=_rel(m.grad, dm_fed))
    worst_own = worst_fed = 0.0
    for k in g_own:
        if named[k].grad is None:
            continue
        ro, rf = _rel(named[k].grad, g_own[k]), _rel(named[k].grad, g_fed[k])
        _note("a7_relu/own/" + k, ro); _note("a7_relu/fed/" + k, rf)
        worst_own, worst_fed = max(worst_own, ro), max(worst_fed, rf)
    _note("a7_relu/own/dtgt", own["dtgt"]); _note("a7_relu/fed/dtgt", fed["dtgt"])
    _note("a7_relu/own/dmem", own["dmem"]); _note("a7_relu/fed/dmem", fed["dmem"])
    _note("a7_relu/out_own", _rel(out, o_own)); _note("a7_relu/out_fed", _rel(out, o_fed))
    # with the activation pattern fixed, what is left is bf16 arithmetic: the class of the smooth layers (<= 1e-2)
    assert fed["dtgt"] < 1e-2 and fed["dmem"] < 1e-2, (own, fed)
    assert worst_fed < 1e-2, (worst_own, worst_fed)
    # and the pattern really was the cause: the own-mask error is several times larger
    assert own["dtgt"] > 2.5 * fed["dtgt"], (own, fed)


def test_config2_decoder_linear1_gradient_with_fed_relu_mask(dev):
    """The 3.9e-2 on vision_decoder.layers.0.linear1.weight (test_fused_head_gradients_with_ignored_labels) by the same
    measurement: oracle fed the HIP path's relu pattern."""
    from oracle import ref_cpu as O
    from test_gpu_edges import _tiny
    m, sd, cfg, _ = _tiny(dev)
    g = torch.Generator().manual_seed(77)
    B, L = 2, 8
    ids = torch.randint(1, 500, (B, L), generator=g)
    inputs = dict(pixel_values=torch.randn(B, 3, 64, 64, generator=g), input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long))
    labels = torch.randint(1, 151, (B, 16, 16), generator=g)
    labels[:, :6] = -100
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    with _CaptureRelu() as cap:
        loss = m.forward_loss(dinputs, labels.to(dev), -100)
    loss.backward()
    assert len(cap.acts) == 1
    mask = (cap.acts[0] > 0).float().cpu().view(B, 16, 128)
    named = dict(m.named_parameters())
    res = {}
    for tag, drops in (("own", None), ("fed", [{"relu_mask": mask}])):
        params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
        _, _, ref = O.base_model_with_text({**sd, **params}, inputs, cfg, dec_drops=drops)
        O.cross_entropy(ref, labels, -100).backward()
        for k in ("vision_decoder.layers.0.linear1.weight", "vision_decoder.layers.0.linear2.weight",
                  "vision_encoder.enc.encoder.layers.1.mlp.fc2.weight"):
            res[(tag, k)] = _rel(named[k].grad, params[k].grad)
            _note(f"cfg2tiny_relu/{tag}/{k}", res[(tag, k)])
    k1 = "vision_decoder.layers.0.linear1.weight"
    assert res[("fed", k1)] < 1.2e-2, res
    assert res[("fed", k1)] < res[("own", k1)], res


# ---------------------------------------------------------------------------------------------------------------------
# config 5 at its real shapes
# ---------------------------------------------------------------------------------------------------------------------
def test_config5_real_shapes_vs_oracle(dev):
    """BASELINE configs[4] / SURVEY §8d config 5, B = 1: Swin-small stage tensors of a 512x512 image, K = 150 text rows,
    HierarchicalCrossA(depth [1,1,1], dim 512, 8 heads, dropout 0) -> score-map tail -> CE at 512x512.  The spatial-reduction
    attention runs at Sq = 4096 / Sk = 1024 (stage 4's third block, stage 3's second, stage 2's one), the stage-1 path at
    16384 tokens, the fused bilinear-x4 + CE tail on K = 150 classes at 512x512."""
    import lc2is_amd.nn as N
    from golden_util import make_weights
    from oracle import ref_cpu as O
    in_dims, dim, K = [96, 192, 384, 768], 512, 150
    m = N.HierarchicalCrossA(in_dims, [1, 1, 1], dim, nhead=8, dropout=0, batch_first=True)
    tail = N.ScoreMapTail(4)
    shapes = {k: list(v.shape) for k, v in m.named_parameters()}
    w = make_weights(shapes, 55)
    named = dict(m.named_parameters())
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    g = torch.Generator().manual_seed(5)
    visual = [torch.randn(1, p, c, generator=g) for p, c in zip((16384, 4096, 1024, 256), in_dims)]
    textual = torch.randn(1, K, dim, generator=g)
    labels = torch.randint(0, K, (1, 512, 512), generator=g)
    labels[:, :40] = -100                                            # a band of ignored pixels (nn.CrossEntropyLoss default)
    def oracle(drops):   # fp32, the box's host cores
        params = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        vis_r = [v.clone().requires_grad_(True) for v in visual]
        txt_r = textual.clone().requires_grad_(True)
        emb_r = O.hierarchical(params, "", vis_r, txt_r, nhead=8, depth=(1, 1, 1), layer_key="layers.0.", drops=drops)
        sm_r = O.score_map_tail(emb_r, txt_r, 4)
        loss_r = O.cross_entropy(sm_r, labels)
        loss_r.backward()
        return params, vis_r, txt_r, emb_r, sm_r, loss_r

    params, vis_r, txt_r, emb_r, sm_r, loss_r = oracle(None)
    # HIP (recording the relu output every SR layer saved for its backward: its sign is the activation pattern the HIP path used)
    import lc2is_amd.nn.hier as Hm
    m = m.to(dev).train()
    vis = [v.to(dev).requires_grad_(True) for v in visual]
    txt = textual.to(dev).requires_grad_(True)
    acts, orig_fwd = [], Hm._sr_layer_fwd

    def recording(x32, x16, mem16, layer, s_, B_, P_, K_, save, ds=None):
        y32, y16, sv = orig_fwd(x32, x16, mem16, layer, s_, B_, P_, K_, save, ds)
        if sv is not None:
            acts.append((P_, sv["a"]))
        return y32, y16, sv

    Hm._sr_layer_fwd = recording
    try:
        emb = m(vis, txt)
    finally:
        Hm._sr_layer_fwd = orig_fwd
    assert emb.shape == (1, 16384, dim)
    # six SR layer applications, in the pyramid's order: stage 4 at 256 / 1024 / 4096 tokens, stage 3 at 1024 / 4096, stage 2 at 4096
    assert [p_ for p_, _ in acts] == [256, 1024, 4096, 1024, 4096, 4096]
    keys = ["attention_stage_4.0.", "attention_stage_4.1.", "attention_stage_4.2.", "attention_stage_3.0.", "attention_stage_3.1.",
            "attention_stage_2.0."]
    fed_drops = {k: [{"relu_mask": (a > 0).float().cpu().view(1, p_, -1)}] for k, (p_, a) in zip(keys, acts)}
    r_emb = _rel(emb, emb_r.detach())
    _note("config5_real/embeddings_rel", r_emb)
    with torch.no_grad():
        sm = tail(emb.detach(), txt.detach())
    assert sm.shape == (1, K, 512, 512)
    sm_err = (sm.cpu() - sm_r.detach()).abs().max().item()
    agree = (sm.argmax(1).cpu() == sm_r.argmax(1)).float().mean().item()
    _note("config5_real/score_map_maxabs", sm_err); _note("config5_real/argmax_agreement", agree)
    del sm
    loss = tail.loss(emb, txt, labels.to(dev))
    _note("config5_real/loss_hip", loss.item()); _note("config5_real/loss_oracle", loss_r.item())
    loss.backward()
    rv0, rv3, rt = _rel(vis[0].grad, vis_r[0].grad), _rel(vis[3].grad, vis_r[3].grad), _rel(txt.grad, txt_r.grad)
    _note("config5_real/dvisual0", rv0); _note("config5_real/dvisual3", rv3); _note("config5_real/dtextual", rt)
    assert vis[1].grad is None and vis[2].grad is None                # never read (model/hierarchical.py:102-112)
    # the same gradients against the oracle FED the HIP path's six relu patterns (the relu-flip
    # explanation of a7 / config 2 shown for the three-stage post-norm SR stack itself)
    params_f, vis_f, txt_f, emb_f, _, loss_f = oracle(fed_drops)
    fv0, fv3, ft = _rel(vis[0].grad, vis_f[0].grad), _rel(vis[3].grad, vis_f[3].grad), _rel(txt.grad, txt_f.grad)
    _note("config5_real/fed/dvisual0", fv0); _note("config5_real/fed/dvisual3", fv3); _note("config5_real/fed/dtextual", ft)
    _note("config5_real/fed/embeddings_rel", _rel(emb, emb_f.detach())); _note("config5_real/fed/loss_oracle", loss_f.item())
    worst, worst_k, worst_f, worst_fk = 0.0, None, 0.0, None
    for k in ("linear2_stage_1.weight", "linear2_stage_4.weight", "linear_stage_3.weight",
              "attention_stage_4.2.layers.0.self_attn.in_proj_weight", "attention_stage_4.2.layers.0.sr.weight",
              "attention_stage_3.1.layers.0.multihead_attn.out_proj.weight", "attention_stage_2.0.layers.0.linear1.weight",
              "attention_stage_4.0.layers.0.norm.weight"):
        if k not in named:
            continue
        rg, rf = _rel(named[k].grad, params[k].grad), _rel(named[k].grad, params_f[k].grad)
        _note("config5_real/grad/" + k, rg); _note("config5_real/fed/grad/" + k, rf)
        if rg > worst:
            worst, worst_k = rg, k
        if rf > worst_f:
            worst_f, worst_fk = rf, k
    _note("config5_real/params_checked", sum(1 for k in REPORT if k.startswith("config5_real/grad/")))
    # Thresholds = measured value x <= 2 (profiles/r04_parity.json; round 3 allowed 2.3-3x).  Own-mask oracle: embeddings 3.8e-3,
    # score map 9.6e-4, argmax agreement 0.9929 (cosine scores of near-tied classes under random weights: the 0.7 % that differ have
    # top-2 margins below the 1e-3 score error — BASELINE.md's 99.9 % gate is for trained weights and is covered by the mIoU gates
    # of test_gpu_parity2.py), dvisual0 3.3e-3, dvisual3 2.6e-2, dtextual 6.5e-3, worst parameter gradient 3.5e-2.
    assert r_emb < 7.5e-3, r_emb
    assert sm_err < 2e-3 and agree > 0.986, (sm_err, agree)           # cosine scores in [-1, 1]
    assert abs(loss.item() - loss_r.item()) < 3e-3
    assert rv0 < 7e-3 and rv3 < 5.2e-2 and rt < 1.3e-2, (rv0, rv3, rt)
    assert sum(1 for k in REPORT if k.startswith("config5_real/grad/")) >= 4
    assert worst < 7e-2, (worst_k, worst)
    # fed-mask oracle: what is left is bf16 arithmetic through three post-norm stages (measured values in profiles/r04_parity.json)
    assert fv3 < FED_DV3_MAX and worst_f < FED_GRAD_MAX, (fv3, worst_fk, worst_f)
    assert fv3 < rv3 and worst_f < worst, ("the activation pattern should explain part of the error", fv3, rv3, worst_f, worst)


# measured with the fed masks (round 4): dvisual3 2.6e-2 -> 5.1e-3, worst parameter gradient 3.5e-2 -> 7.5e-3 (attention_stage_4.0's
# shared norm), dtextual 6.5e-3 -> 4.8e-3: the relu pattern carried 4/5 of the config-5 gradient error.  Bounds = measured x 2.
FED_DV3_MAX, FED_GRAD_MAX = 1.0e-2, 1.5e-2


# ---------------------------------------------------------------------------------------------------------------------
# config 4 beyond one layer, with gradients
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_depth4_gradients_vs_oracle(dev):
    """ViT-L/14 @640x640 (grid 45 -> 2026 tokens, widths 1024 / 16 heads / 4096), 4 layers; text width 768 / 12 heads, 2 layers;
    decoder d_model 1024 with 8 heads (head_dim 128), d_kv 768; output 180x180.  No reference code path exists for it
    (model/encoder.py:18-21 maps patch 16 only): parity is against the CPU oracle."""
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    torch.manual_seed(14)
    m = N.BaseModelWithText(14, 640, 180, vision_arch=N.ClipArch(1024, 16, 4, 4096),
                            text_arch=N.ClipArch(768, 12, 2, 3072, vocab=1000, eos_token_id=999), nhead=8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(15)
    B, L = 1, 8
    inputs = dict(pixel_values=torch.randn(B, 3, 640, 640, generator=g), input_ids=torch.randint(1, 998, (B, L), generator=g),
                  attention_mask=torch.ones(B, L, dtype=torch.int64))
    inputs["input_ids"][:, -1] = 999
    labels = torch.randint(0, 151, (B, 180, 180), generator=g)
    cfg = O.BaseCfg(in_size=640, out_size=180, patch=14, vision=O.ClipCfg(1024, 16, 4, patch=14),
                    text=O.ClipCfg(768, 12, 2, eos_token_id=999), dec_heads=8, dec_layers=1)
    ref_loss, ref_logits, ref_grads, _ = O.train_step_sgd(sd, inputs, labels, cfg, 1e-5)
    m = m.to(dev).train()
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    with torch.no_grad():
        out = m(dinputs)["outputs"]
    r = _rel(out, ref_logits)
    _note("config4_d4/logits_rel", r)
    loss = m.forward_loss(dinputs, labels.to(dev))
    _note("config4_d4/loss_hip", loss.item()); _note("config4_d4/loss_oracle", float(ref_loss))
    loss.backward()
    named = dict(m.named_parameters())
    worst, worst_k = 0.0, None
    for k in ("vision_encoder.enc.embeddings.patch_embedding.weight",
              "vision_encoder.enc.encoder.layers.0.self_attn.q_proj.weight",
              "vision_encoder.enc.encoder.layers.0.mlp.fc1.weight",
              "vision_encoder.enc.encoder.layers.3.mlp.fc2.weight",
              "vision_encoder.enc.encoder.layers.3.self_attn.out_proj.weight",
              "text_encoder.enc.encoder.layers.1.mlp.fc1.weight",
              "vision_decoder.layers.0.self_attn.in_proj_weight",
              "vision_decoder.layers.0.multihead_attn.k_proj_weight",
              "pixel_patch.visual.weight", "class_prototypes"):
        rg = _rel(named[k].grad, ref_grads[k])
        _note("config4_d4/grad/" + k, rg)
        if rg > worst:
            worst, worst_k = rg, k
    assert out.shape == (B, 151, 180, 180)
    assert r < 1.5e-2, r
    assert abs(loss.item() - float(ref_loss)) < 1e-2
    assert worst < 4e-2, (worst_k, worst)


def test_config4_full_depth_forward_and_loss_vs_oracle(dev):
    """BASELINE configs[3] at its REAL depth: ViT-L/14 @640x640 (2026 tokens, width 1024, 16 heads, ff 4096), all 24 layers; CLIP
    text width 768 / 12 heads, 12 layers; decoder d_model 1024 / 8 heads (head_dim 128) / d_kv 768; output 180x180, B = 1.
    Forward logits, the CE loss AND (round 5) the ten-gradient list of the depth-4 test — first / middle / last
    ViT layer, patch embedding, text tower, decoder, head — against one train step of the fp32 CPU oracle (forward + backward of
    24 ViT-L layers at 2026 tokens: tens of seconds on the box's cores).  No reference code path exists for ViT-L/14
    (model/encoder.py:18-21): the oracle is the only pin.  Tolerances: measured x 2 (profiles/r05_parity.json)."""
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    torch.manual_seed(24)
    m = N.BaseModelWithText(14, 640, 180, vision_arch=N.ClipArch(1024, 16, 24, 4096),
                            text_arch=N.ClipArch(768, 12, 12, 3072, vocab=1000, eos_token_id=999), nhead=8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(25)
    B, L = 1, 8
    inputs = dict(pixel_values=torch.randn(B, 3, 640, 640, generator=g), input_ids=torch.randint(1, 998, (B, L), generator=g),
                  attention_mask=torch.ones(B, L, dtype=torch.int64))
    inputs["input_ids"][:, -1] = 999
    labels = torch.randint(0, 151, (B, 180, 180), generator=g)
    cfg = O.BaseCfg(in_size=640, out_size=180, patch=14, vision=O.ClipCfg(1024, 16, 24, patch=14),
                    text=O.ClipCfg(768, 12, 12, eos_token_id=999), dec_heads=8, dec_layers=1)
    ref_loss, ref_logits, ref_grads, _ = O.train_step_sgd(sd, inputs, labels, cfg, 1e-5)
    m = m.to(dev).eval()
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    with torch.no_grad():
        out = m(dinputs)["outputs"]
    r = _rel(out, ref_logits)
    agree = (out.argmax(1).cpu() == ref_logits.argmax(1)).float().mean().item()
    _note("config4_d24/logits_rel", r); _note("config4_d24/argmax_agreement", agree)
    m.train()
    loss = m.forward_loss(dinputs, labels.to(dev))
    _note("config4_d24/loss_hip", loss.item()); _note("config4_d24/loss_oracle", float(ref_loss))
    loss.backward()
    named = dict(m.named_parameters())
    worst, worst_k = 0.0, None
    for k in ("vision_encoder.enc.embeddings.patch_embedding.weight",
              "vision_encoder.enc.encoder.layers.0.self_attn.q_proj.weight",
              "vision_encoder.enc.encoder.layers.0.mlp.fc1.weight",
              "vision_encoder.enc.encoder.layers.12.mlp.fc2.weight",
              "vision_encoder.enc.encoder.layers.23.mlp.fc2.weight",
              "vision_encoder.enc.encoder.layers.23.self_attn.out_proj.weight",
              "text_encoder.enc.encoder.layers.11.mlp.fc1.weight",
              "vision_decoder.layers.0.self_attn.in_proj_weight",
              "vision_decoder.layers.0.multihead_attn.k_proj_weight",
              "pixel_patch.visual.weight", "class_prototypes"):
        rg = _rel(named[k].grad, ref_grads[k])
        _note("config4_d24/grad/" + k, rg)
        if rg > worst:
            worst, worst_k = rg, k
    assert out.shape == (B, 151, 180, 180)
    assert r < CFG4_D24_LOGITS_MAX, r
    assert abs(loss.item() - float(ref_loss)) < CFG4_D24_LOSS_MAX
    assert worst < CFG4_D24_GRAD_MAX, (worst_k, worst)


# measured (round 4): logits rel-L2 6.6e-3, argmax agreement 0.9915 (random weights), loss 17.1838 vs 17.1778.  Bounds = measured x 2.
CFG4_D24_LOGITS_MAX, CFG4_D24_LOSS_MAX = 1.3e-2, 1.2e-2
# round 5: the eleven gradients at depth 24 measure 0.60e-2 .. 1.45e-2 (profiles/r05_parity.json).  Bound = worst x 2.
CFG4_D24_GRAD_MAX = 3e-2


# ---------------------------------------------------------------------------------------------------------------------
# mIoU: an image with no class outside ignore_index
# ---------------------------------------------------------------------------------------------------------------------
def test_device_miou_all_ignored_image_is_nan_like_the_reference(dev):
    """metrics.py:94-97: `classes = label.unique(); classes = classes[classes != ignore_index]; iou[classes].mean()` — the
    mean of an EMPTY selection is NaN, and the dataset mean (:101) inherits it.  The device metric must not turn that image
    into a 0 (round 2 did: clamp_min(1))."""
    from lc2is_amd.metrics import compute_mIOU
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(9)
    labels = torch.randint(1, 151, (3, 8, 8), generator=g)
    labels[1] = 0                                                     # image 1: only the ignored class
    logits = torch.randn(3, 151, 8, 8, generator=g)
    ref = O.compute_miou(logits, labels)
    got = compute_mIOU(logits.to(dev), labels.to(dev))["mIOU_label"]
    assert math.isnan(ref) and math.isnan(got), (ref, got)
    # without that image both are finite and agree
    keep = [0, 2]
    ref2 = O.compute_miou(logits[keep], labels[keep])
    got2 = compute_mIOU(logits[keep].to(dev), labels[keep].to(dev))["mIOU_label"]
    assert abs(ref2 - got2) < 1e-3, (ref2, got2)
