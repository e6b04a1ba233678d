"""Round-2 parity gates on the GPU (all through the C ABI):

* a7  PromptDecoder / PromptLayer at its real shape (post-norm, d_kv 1024 != d_model 512, K = 150 text queries over 256
      visual tokens) against vectors from the REFERENCE's own classes (tests/golden/prompt_decoder.pt);
* a3  ImageEncoderCLIPFull (CLS row kept; its gradient path) against reference vectors (clip_full_tiny.pt);
* config 2 at FULL DEPTH (ViT-B/16 12 layers + CLIP text 12 layers + decoder + head, 512x512, K = 151, real prototypes):
      logits, cross-entropy and selected parameter gradients against the CPU oracle, plus the error per depth;
* g1  the end-to-end mIoU gate: evaluation driver (lc2is_amd.evalloop, engine.py:125-168) over the 16 synthetic
      config-1 images of SURVEY.md §8d — HIP logits -> device mIoU vs oracle logits -> oracle mIoU, |delta| <= 0.1 mIoU
      point — with random-init weights, with weights overfitted on the shard (the reference's own evaluate.sh scores an
      "overfit" checkpoint), and at 512x512 on 4 images.

Tolerances are <= 2x the values measured on MI355X (profiles/r02_parity.json = the run's gpurun_out/parity_r02.json;
DESIGN.md §2).
"""
import json
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
        (out / "parity_r02.json").write_text(json.dumps(REPORT, indent=1))
    except OSError:
        pass


# ---------------------------------------------------------------------------------------------------------------------
# a7
# ---------------------------------------------------------------------------------------------------------------------
def _prompt_decoder(dev, dropout):
    import lc2is_amd.nn as N
    from golden_util import make_weights, prompt_inputs
    fx = torch.load(G / "prompt_decoder.pt", weights_only=True)
    kw = {} if dropout is None else dict(dropout=dropout)
    dec = N.PromptDecoder(N.PromptLayer(d_model=512, d_kv=1024, nhead=8, batch_first=True, **kw), num_layers=2)
    named = dict(dec.named_parameters())
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    assert {k: list(v.shape) for k, v in named.items()} == shapes          # the reference's parameter set (torch-2.10 drift)
    w = make_weights(shapes, int(fx["wseed"]))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    return dec.to(dev), fx, prompt_inputs(int(fx["iseed"]))


def test_prompt_decoder_eval_default_dropout_vs_reference(dev):
    """The reference's own construction (dropout 0.1 default, model/decoder.py:26) in eval mode: forward only."""
    dec, fx, (tgt, mem, _) = _prompt_decoder(dev, None)
    assert dec.layers[0].dropout_p == 0.1 and dec.layers[0].norm_first is False
    dec.eval()
    with torch.no_grad():
        out = dec(tgt=tgt.to(dev), memory=mem.to(dev))
    r = _rel(out, fx["out"])
    _note("prompt_decoder_eval_out_rel", r)
    assert out.shape == (2, 150, 512) and r < 6e-3                       # measured 3.1e-3


def test_prompt_decoder_train_vs_reference(dev):
    """dropout = 0 training forward + backward (same numbers as the reference's eval-mode pass): output, input gradients
    (tgt = text embeddings, memory = visual tokens) and parameter gradients of the post-norm branch with d_kv != d_model."""
    dec, fx, (tgt, mem, dout) = _prompt_decoder(dev, 0.0)
    dec.train()
    t = tgt.to(dev).requires_grad_(True)
    m = mem.to(dev).requires_grad_(True)
    out = dec(tgt=t, memory=m)
    assert _rel(out, fx["out"]) < 6e-3
    out.backward(dout.to(dev))
    rt, rm = _rel(t.grad, fx["dtgt"]), _rel(m.grad[:, ::4], fx["dmem_rows"])
    _note("prompt_decoder_dtgt_rel", rt); _note("prompt_decoder_dmem_rel", rm)
    # measured 3.4e-2 / 3.1e-2, and 2.9-3.5e-2 on every parameter BEHIND the feed-forward block in backward order (0.3e-2 in
    # front of it).  Cause, MEASURED in round 3 (test_gpu_parity3.py::test_prompt_decoder_gradient_error_is_the_relu_mask): the
    # relu pattern of the HIP path (pre-activations from bf16 inputs) differs from the fp32 oracle's in the units next to zero;
    # with the oracle fed the HIP path's pattern the same gradients agree to 3.6e-3 / 7.2e-3 (all 26 parameters <= 8e-3).
    # The tolerance below is 2x the measured own-pattern error; the smooth quick_gelu towers of config 2 measure 0.7-1.9e-2.
    assert rt < 6.5e-2 and rm < 6e-2
    named = dict(dec.named_parameters())
    worst = 0.0
    for k, g in fx["grad_full"].items():
        r = _rel(named[k].grad, g)
        _note("prompt_decoder_grad/" + k, r)
        worst = max(worst, r)
    assert worst < 7e-2, worst                                            # measured 3.5e-2
    for k, st in fx["grad_stats"].items():
        g = named[k].grad
        assert g is not None, k
        ref_abs = float(st[1])
        if ref_abs < 1e-6 * g.numel():
            assert float(g.abs().mean()) < 1e-3, k
            continue
        assert abs(float(g.abs().sum()) - ref_abs) < 0.08 * ref_abs, (k, float(g.abs().sum()), ref_abs)


# ---------------------------------------------------------------------------------------------------------------------
# a3
# ---------------------------------------------------------------------------------------------------------------------
def test_image_encoder_clip_full_vs_reference(dev):
    import lc2is_amd.nn as N
    from golden_util import clip_full_inputs, make_weights
    fx = torch.load(G / "clip_full_tiny.pt", weights_only=True)
    enc = N.ImageEncoderCLIPFull(64, 16, arch=N.ClipArch(128, 2, 2, 256))
    named = dict(enc.named_parameters())
    shapes = {k: v.tolist() for k, v in fx["shapes"].items()}
    assert {k: list(v.shape) for k, v in named.items()} == shapes
    w = make_weights(shapes, int(fx["wseed"]))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    enc = enc.to(dev).train()
    pix, dout = clip_full_inputs(int(fx["iseed"]))
    out = enc(pix.to(dev))
    assert out.shape == (2, 17, 128)                                       # 16 patches + the CLS row
    r = _rel(out, fx["out"])
    rcls = _rel(out[:, 0], fx["out"][:, 0])
    _note("clip_full_out_rel", r); _note("clip_full_cls_row_rel", rcls)
    assert r < 8e-3 and rcls < 8e-3                                        # measured 4.2e-3
    out.backward(dout.to(dev))
    named = dict(enc.named_parameters())
    for k, g in fx["grad_full"].items():
        rg = _rel(named[k].grad, g)
        _note("clip_full_grad/" + k, rg)
        assert rg < 2e-2, (k, rg)                                          # measured 0.5-0.9e-2
    # the same upstream gradient with the CLS row zeroed must give a DIFFERENT class-embedding gradient (the CLS path is live)
    g_cls = named["enc.embeddings.class_embedding"].grad.clone()
    for p in enc.parameters():
        p.grad = None
    d2 = dout.clone(); d2[:, 0] = 0
    enc(pix.to(dev)).backward(d2.to(dev))
    assert _rel(named["enc.embeddings.class_embedding"].grad, g_cls) > 1e-2


# ---------------------------------------------------------------------------------------------------------------------
# config 2 at full depth
# ---------------------------------------------------------------------------------------------------------------------
def _cfg_full(O, in_size, out_size, vl=12, tl=12):
    return O.BaseCfg(in_size=in_size, out_size=out_size, patch=16, vision=O.ClipCfg(768, 12, vl, patch=16),
                     text=O.ClipCfg(512, 8, tl), dec_heads=8, dec_layers=1)


def test_vision_tower_error_per_depth(dev):
    """ViT-B/16 at 512x512 (1025 tokens), depth 1/2/4/8/12 with the same weights: rel-L2 of the HIP tokens against the
    fp32 oracle, recorded per depth (bf16 rounding compounds; the bound is 2x the measured 12-layer value)."""
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    torch.manual_seed(1024)
    full = N.ImageEncoderCLIP(512, 16)
    sd_full = {k: v.detach().clone() for k, v in full.state_dict().items()}
    pix = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(2))
    osd = {"v." + k: v for k, v in sd_full.items()}
    errs = {}
    for depth in (1, 2, 4, 8, 12):
        enc = N.ImageEncoderCLIP(512, 16, arch=N.ClipArch(768, 12, depth, 3072))
        enc.load_state_dict({k: v for k, v in sd_full.items() if not k.startswith("enc.encoder.layers.")
                             or int(k.split(".")[3]) < depth}, strict=True)
        enc = enc.to(dev).eval()
        with torch.no_grad():
            out = enc(pix.to(dev))
        ref = O.image_encoder_clip(osd, "v.", pix, O.ClipCfg(768, 12, depth, patch=16))
        errs[depth] = _rel(out, ref)
        _note(f"vit_b16_512_rel_depth{depth}", errs[depth])
        del enc
    # measured on MI355X: 2.42e-3, 2.49e-3, 2.62e-3, 2.86e-3, 3.06e-3 at depth 1, 2, 4, 8, 12
    assert all(errs[d] < 6e-3 for d in errs), errs
    assert errs[12] < 2.0 * errs[1] + 1e-3, errs                            # slow growth with depth, no blow-up


def _config2_vs_oracle(dev, batch, tag):
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    from bench import synth_batch
    torch.manual_seed(1024)
    m = N.BaseModelWithText(16, 512, 128)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    inputs, labels = synth_batch(batch, 512, 128, 16, 2, "cpu")
    cfg = _cfg_full(O, 512, 128)
    ref_loss, ref_logits, ref_grads, _ = O.train_step_sgd(sd, inputs, labels, cfg, 1e-5)
    m = m.to(dev).train()
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    with torch.no_grad():
        out = m(dinputs)["outputs"]
    r = _rel(out, ref_logits)
    mx = (out.cpu() - ref_logits).abs().max().item() / ref_logits.abs().max().item()
    agree = (out.argmax(1).cpu() == ref_logits.argmax(1)).float().mean().item()
    _note(tag + "_logits_rel_l2", r); _note(tag + "_logits_maxabs_over_maxlogit", mx)
    _note(tag + "_argmax_agreement", agree)
    loss = m.forward_loss(dinputs, labels.to(dev))
    _note(tag + "_loss_hip", loss.item()); _note(tag + "_loss_oracle", float(ref_loss))
    loss.backward()
    named = dict(m.named_parameters())
    worst = 0.0
    for k in ("vision_encoder.enc.encoder.layers.0.self_attn.q_proj.weight",
              "vision_encoder.enc.encoder.layers.11.mlp.fc2.weight",
              "vision_encoder.enc.embeddings.patch_embedding.weight",
              "text_encoder.enc.embeddings.token_embedding.weight",
              "text_encoder.enc.encoder.layers.11.mlp.fc1.weight",
              "vision_decoder.layers.0.multihead_attn.k_proj_weight",
              "pixel_patch.visual.weight", "class_prototypes"):
        rg = _rel(named[k].grad, ref_grads[k])
        _note(tag + "_grad/" + k, rg)
        worst = max(worst, rg)
    return r, mx, agree, loss.item(), float(ref_loss), worst


def test_config2_full_depth_vs_oracle(dev):
    """BASELINE configs[1] architecture at full depth, B = 2, text length 16: logits, CE and parameter gradients."""
    r, mx, agree, loss, ref_loss, worst = _config2_vs_oracle(dev, 2, "config2_full")
    # measured: logits rel-L2 6.4e-3, max-abs 6.0e-3 of the largest logit, argmax agreement 99.05 % (random-init logits are
    # nearly tied), CE 16.9624 vs 16.9638, gradients 0.7-1.9e-2
    assert r < 1.3e-2 and mx < 1.2e-2, (r, mx)
    # argmax agreement: measured 0.9905, bound = 1 - 2 x (1 - measured).  BASELINE.md §4's 99.9 % starting gate is for TRAINED
    # weights (separated classes); at random init the top-2 logit margin of ~1 % of the pixels is below the 6e-3 logit error, and
    # what the gate protects — identical mIoU — is asserted on a fitted model in test_miou_gate_* below.
    assert agree > 0.981, agree
    assert abs(loss - ref_loss) < 5e-3
    assert worst < 4e-2, worst   # measured 1.9e-2 (x 2)


def test_config2_bench_batch_vs_oracle(dev):
    """The SAME comparison at the batch bench.py times (B = 32, M = 32 800 token rows): the 256x384-tile, tail-fold and persistent
    GEMM plans, the whole-tower weight-gradient table and the 3456-block attention grids only exist at this size; until round 5
    they were pinned by bitwise-vs-single-kernel tests alone.  The oracle's fp32 train step of 32 images
    takes about a minute of the box's host cores."""
    r, mx, agree, loss, ref_loss, worst = _config2_vs_oracle(dev, 32, "config2_b32")
    # bounds = the B = 2 test's (the error is per token row, not a function of the batch)
    assert r < 1.3e-2 and mx < 1.2e-2, (r, mx)
    assert agree > 0.981, agree
    assert abs(loss - ref_loss) < 5e-3, (loss, ref_loss)
    assert worst < 4e-2, worst


# ---------------------------------------------------------------------------------------------------------------------
# g1: the mIoU gate
# ---------------------------------------------------------------------------------------------------------------------
def _miou_gate(dev, m, sd, batches, cfg, tag):
    import lc2is_amd.nn as N
    from lc2is_amd.evalloop import Evaluator, segmentation_metrics
    from oracle import ref_cpu as O
    ev = Evaluator(m, batches, N.CrossEntropyLoss(), compute_metrics=segmentation_metrics, device=dev)
    got = ev.evaluate()
    ref = O.evaluate(sd, batches, cfg)
    d_pts = 100.0 * abs(got["eval_mIOU_label"] - ref["eval_mIOU_label"])
    _note(tag + "_miou_hip", got["eval_mIOU_label"]); _note(tag + "_miou_oracle", ref["eval_mIOU_label"])
    _note(tag + "_miou_delta_points", d_pts)
    _note(tag + "_eval_loss_hip", got["eval_loss"]); _note(tag + "_eval_loss_oracle", ref["eval_loss"])
    assert set(got) == {"eval_loss", "eval_mIOU_label"}
    assert abs(got["eval_loss"] - ref["eval_loss"]) < 2e-2
    assert d_pts <= 0.1, (got, ref["eval_mIOU_label"])     # north_star: mIoU within +-0.1 of the CPU reference
    return got, ref


def test_miou_gate_config1_random_init(dev):
    """configs[0]: 16 images, 128x128, batch 1, real prototypes, weights seed 1024 (evaluate.py:24)."""
    import lc2is_amd.nn as N
    from golden_util import config1_batch
    from oracle import ref_cpu as O
    torch.manual_seed(1024)
    m = N.BaseModelWithText(16, 128, 32)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batches = [config1_batch(i) for i in range(16)]
    _miou_gate(dev, m, sd, batches, _cfg_full(O, 128, 32), "config1_random")


def test_miou_gate_config1_overfit(dev):
    """The same shard after overfitting it on the HIP path (AdamW, 150 steps of the 16-image batch): predictions now
    follow the labels, so the mIoU is far from the chance level and a logit error would move it."""
    import lc2is_amd.nn as N
    from golden_util import config1_batch
    from lc2is_amd.step import TrainStep
    from oracle import ref_cpu as O
    torch.manual_seed(1024)
    m = N.BaseModelWithText(16, 128, 32).to(dev).train()
    batches = [config1_batch(i) for i in range(16)]
    big = {k: torch.cat([b[0][k] for b in batches]).to(dev) for k in ("pixel_values", "input_ids", "attention_mask")}
    lab = torch.cat([b[0]["label"] for b in batches]).to(dev)
    ts = TrainStep(m, optimizer="adamw", lr=1e-4)
    first = last = None
    for it in range(150):
        last = ts.step(big, lab)
        if it == 0:
            first = float(last)
    last = float(last)
    _note("config1_overfit_loss_first", first); _note("config1_overfit_loss_last", last)
    assert last < 0.6 * first, (first, last)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    got, ref = _miou_gate(dev, m, sd, batches, _cfg_full(O, 128, 32), "config1_overfit")
    assert ref["eval_mIOU_label"] > 0.05                                    # well above the ~0.3 % of chance


def test_miou_gate_512_four_images(dev):
    """configs[1] geometry (512x512 -> 128x128 logits -> x4 for the metric), 4 images, batch 1."""
    import lc2is_amd.nn as N
    from golden_util import config1_batch
    from oracle import ref_cpu as O
    torch.manual_seed(1024)
    m = N.BaseModelWithText(16, 512, 128)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batches = [config1_batch(i, in_size=512, out_size=128) for i in range(4)]
    _miou_gate(dev, m, sd, batches, _cfg_full(O, 512, 128), "config2_4img")


# ---------------------------------------------------------------------------------------------------------------------
# ignored labels: gradient scale of the fused head
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ignore_index", [-100, 0])
def test_fused_head_gradients_with_ignored_labels(dev, ignore_index):
    """nn.CrossEntropyLoss divides by the number of NON-ignored pixels; the fused head must scale its gradient by the same
    count (default ignore_index = -100 included): compared with the oracle and with the unfused CrossEntropyLoss path."""
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    from test_gpu_edges import _tiny
    m, sd, cfg, _ = _tiny(dev)
    g = torch.Generator().manual_seed(77)
    B, L = 2, 8
    ids = torch.randint(1, 500, (B, L), generator=g)
    inputs = dict(pixel_values=torch.randn(B, 3, 64, 64, generator=g), input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long))
    labels = torch.randint(1, 151, (B, 16, 16), generator=g)
    labels[:, :6] = ignore_index                                             # 37.5 % of the pixels ignored
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    _, _, ref = O.base_model_with_text({**sd, **params}, inputs, cfg)
    ref_loss = O.cross_entropy(ref, labels, ignore_index)
    ref_loss.backward()
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    loss = m.forward_loss(dinputs, labels.to(dev), ignore_index)
    assert abs(float(loss) - float(ref_loss)) < 2e-2
    loss.backward()
    named = dict(m.named_parameters())
    keys = ("class_prototypes", "pixel_patch.visual.weight", "vision_decoder.layers.0.linear1.weight")
    fused = {k: named[k].grad.clone() for k in keys}
    for k in keys:
        r = _rel(fused[k], params[k].grad)
        _note(f"ignored{ignore_index}_fused_grad/{k}", r)
        assert r < (8e-2 if "linear1" in k else 1.2e-2), (k, r)               # measured 5.8e-3 / 6.0e-3 / 3.9e-2 (relu mask flips)
    for p in m.parameters():
        p.grad = None
    N.CrossEntropyLoss(ignore_index=ignore_index)(m(dinputs)["outputs"], labels.to(dev)).backward()
    for k in keys:
        assert _rel(named[k].grad, fused[k]) < 2e-2, k


# ---------------------------------------------------------------------------------------------------------------------
# optimizer semantics for parameters without a gradient: torch.optim skips them, weight decay included
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opt", ["sgd", "adamw"])
def test_frozen_parameters_are_untouched_under_weight_decay(dev, opt):
    from lc2is_amd.step import TrainStep
    from test_gpu_edges import _tiny
    m, sd, cfg, _ = _tiny(dev)
    for p in m.text_encoder.parameters():                                   # the reference freezes its text towers so
        p.requires_grad = False                                             # (model/model.py:115-117, ftn.py:32-34)
    g = torch.Generator().manual_seed(5)
    inputs = dict(pixel_values=torch.randn(2, 3, 64, 64, generator=g).to(dev),
                  input_ids=torch.randint(1, 500, (2, 6), generator=g).to(dev),
                  attention_mask=torch.ones(2, 6, dtype=torch.long, device=dev))
    labels = torch.randint(0, 151, (2, 16, 16), generator=g).to(dev)
    ts = TrainStep(m, optimizer=opt, lr=1e-2, weight_decay=0.1)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ts.step(inputs, labels)
    ts.step(inputs, labels)
    after = m.state_dict()
    untouched = [k for k in before if k.startswith("text_encoder.") or "post_layernorm" in k]
    assert len(untouched) > 10
    for k in untouched:                                                     # frozen tower + the unreached post_layernorm
        assert torch.equal(after[k], before[k]), k
    moved = [k for k in before if k.startswith("vision_decoder.") and not torch.equal(after[k], before[k])]
    assert len(moved) > 5                                                   # the trainable parameters did move (and decay)
    w0, w1 = before["pixel_patch.visual.weight"], after["pixel_patch.visual.weight"]
    assert float((w1 - w0).abs().max()) > 0
