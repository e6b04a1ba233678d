"""Generate tests/golden/*.pt by running the REFERENCE's own modules (imported from /root/reference) on
seeded inputs, in the build container (CPU).  Fixtures are tensors only (loadable with weights_only=True):
weights, inputs, and the reference's outputs / losses / gradients / post-SGD parameters.

Hub-named constructors cannot run offline, so the reference wrappers are created without
their ``__init__`` and given config-built, seeded transformers models; every ``forward`` executed below is
the reference's own, unchanged.

Usage:  python tools/make_golden.py            (needs /root/reference; never runs on the GPU box)
"""
from __future__ import annotations

import sys
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"

import torch  # noqa: E402
from torch import nn  # noqa: E402


def _ref_imports():
    sys.path.insert(0, str(REF))
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPVisionConfig, CLIPVisionModel  # noqa
    import model.decoder as rdec  # noqa
    import model.encoder as renc  # noqa
    import model.loss as rloss  # noqa
    import model.model as rmodel  # noqa
    import model.text_patch as rtp  # noqa
    return dict(CLIPTextConfig=CLIPTextConfig, CLIPTextModel=CLIPTextModel, CLIPVisionConfig=CLIPVisionConfig,
                CLIPVisionModel=CLIPVisionModel, rdec=rdec, renc=renc, rloss=rloss, rmodel=rmodel, rtp=rtp)


def _bare(cls):
    obj = object.__new__(cls)
    nn.Module.__init__(obj)
    return obj


def make_base_tiny(R):
    """BaseModelWithText (model/model.py:12-56) at HIP-compatible tiny dims + one SGD training step."""
    torch.manual_seed(1024)  # evaluate.py:24 default seed
    vcfg = R["CLIPVisionConfig"](hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                 image_size=64, patch_size=16)
    tcfg = R["CLIPTextConfig"](vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                               num_attention_heads=1, max_position_embeddings=77, eos_token_id=511, bos_token_id=510,
                               pad_token_id=511)
    m = _bare(R["rmodel"].BaseModelWithText)
    m.patch_size, m.in_size, m.out_size = 16, 64, 16
    m.vision_encoder = _bare(R["renc"].ImageEncoderCLIP)
    m.vision_encoder.in_size, m.vision_encoder.patch_size = 64, 16
    m.vision_encoder.enc = R["CLIPVisionModel"](vcfg)
    m.text_encoder = _bare(R["renc"].TextEncoderCLIP)
    m.text_encoder.patch_size = 16
    m.text_encoder.enc = R["CLIPTextModel"](tcfg)
    protos = torch.load(REF / "model" / "ade20k_prototypes.pt", weights_only=True)
    m.class_prototypes = nn.Parameter(protos.clone(), requires_grad=True)
    layer = R["rdec"].DecoderLayer(d_model=128, d_kv=64, nhead=2, dim_feedforward=128, dropout=0, batch_first=True,
                                   norm_first=True)
    m.vision_decoder = R["rdec"].DecoderBlock(decoder_layer=layer, num_layers=1)
    m.pixel_patch = R["rtp"].TextToPatch(out=64, img_in=128, text_in=512)
    m.train()

    g = torch.Generator().manual_seed(7)
    B, L = 2, 8
    pixel_values = torch.randn(B, 3, 64, 64, generator=g)
    input_ids = torch.randint(1, 509, (B, L), generator=g)
    input_ids[:, 0] = 510
    input_ids[0, 5] = 511
    input_ids[0, 6:] = 511
    input_ids[1, 7] = 511
    attention_mask = torch.ones(B, L, dtype=torch.int64)
    attention_mask[0, 6:] = 0
    labels = torch.randint(0, 151, (B, 16, 16), generator=g)
    inputs = dict(pixel_values=pixel_values, input_ids=input_ids, attention_mask=attention_mask)

    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    feature_t, feature_v, logits = m(inputs)
    loss = nn.CrossEntropyLoss()(logits, labels)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    opt.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    opt.step()
    sd1 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # keep the fixture small: all grads as per-tensor (sum, abs-sum, first 8 values) + a few full tensors
    keep_full = ["class_prototypes", "pixel_patch.visual.weight", "vision_decoder.layers.0.multihead_attn.k_proj_weight",
                 "vision_encoder.enc.encoder.layers.0.self_attn.q_proj.weight",
                 "vision_encoder.enc.embeddings.position_embedding.weight",
                 "text_encoder.enc.encoder.layers.1.mlp.fc1.weight", "vision_encoder.enc.pre_layrnorm.bias"]
    fx = dict(
        state_dict=sd0, pixel_values=pixel_values, input_ids=input_ids, attention_mask=attention_mask, labels=labels,
        feature_t=feature_t.detach(), feature_v=feature_v.detach(), logits=logits.detach(), loss=loss.detach(),
        grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items()},
        grad_full={k: grads[k] for k in keep_full},
        after_step={k: sd1[k] for k in keep_full}, lr=torch.tensor(0.05),
    )
    torch.save(fx, OUT / "base_tiny.pt")
    print("base_tiny: loss", float(loss), "logits", tuple(logits.shape), "n_grads", len(grads))


def make_contrastive(R):
    """ContrastiveModel (model/model.py:58-103) + ContrastiveLoss (model/loss.py:39-64, 151 classes hard-coded) at tiny
    dims: forward, loss, backward."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights
    torch.manual_seed(1024)
    vcfg = R["CLIPVisionConfig"](hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                 image_size=64, patch_size=16)
    tcfg = R["CLIPTextConfig"](vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                               num_attention_heads=1, max_position_embeddings=77, eos_token_id=511, bos_token_id=510,
                               pad_token_id=511)
    m = _bare(R["rmodel"].ContrastiveModel)
    m.patch_size, m.in_size, m.out_size = 16, 64, 16
    m.vision_encoder = _bare(R["renc"].ImageEncoderCLIP)
    m.vision_encoder.in_size, m.vision_encoder.patch_size = 64, 16
    m.vision_encoder.enc = R["CLIPVisionModel"](vcfg)
    m.text_encoder = _bare(R["renc"].TextEncoderCLIPPooler)
    m.text_encoder.patch_size = 16
    m.text_encoder.enc = R["CLIPTextModel"](tcfg)
    m.pixel_patch = R["rtp"].TextToPatch(out=64, img_in=128, text_in=64)
    m.train()
    params = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in params.items()}
    w = make_weights(shapes, 71)
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(w[k])
    g = torch.Generator().manual_seed(72)
    B, Nt, L = 2, 151, 8
    pixel_values = torch.randn(B, 3, 64, 64, generator=g)
    input_ids = torch.randint(1, 509, (Nt, L), generator=g)
    input_ids[:, 0] = 510
    eos = torch.randint(2, L, (Nt,), generator=g)
    for i in range(Nt):
        input_ids[i, eos[i]:] = 511
    labels = torch.randint(0, 151, (B, 16, 16), generator=g)
    feature_t, feature_v, logits = m(dict(pixel_values=pixel_values, input_ids=input_ids))
    loss, lv, lt = R["rloss"].ContrastiveLoss()(outputs=logits, labels=labels)
    loss.backward()
    keep = [k for k in params if params[k].grad is not None and (k.startswith("pixel_patch.") or "layers.0.self_attn.q_proj.weight" in k
                                                                  or k.endswith("final_layer_norm.weight") or k.endswith("pre_layrnorm.bias"))]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(71), pixel_values=pixel_values,
              input_ids=input_ids, labels=labels, feature_t=feature_t.detach(), feature_v=feature_v.detach()[:, ::7].clone(),
              logits=logits.detach(), loss=loss.detach(), loss_visual=lv.detach(), loss_textual=lt.detach(),
              grad_stats={k: (torch.stack([p.grad.sum(), p.grad.abs().sum()]) if p.grad is not None else torch.zeros(2))
                          for k, p in params.items()},
              no_grad=torch.tensor([i for i, k in enumerate(params) if params[k].grad is None]),
              grad_full={k: params[k].grad.clone() for k in keep})
    torch.save(fx, OUT / "contrastive_tiny.pt")
    print("contrastive:", tuple(logits.shape), float(loss), "params without grad:", len(fx["no_grad"]))


def make_decoder_d96(R):
    """DecoderBlock/DecoderLayer (model/decoder.py:9-21) with head_dim 96, key padding mask, 2 layers."""
    torch.manual_seed(11)
    layer = R["rdec"].DecoderLayer(d_model=192, d_kv=128, nhead=2, dim_feedforward=128, dropout=0, batch_first=True,
                                   norm_first=True)
    blk = R["rdec"].DecoderBlock(decoder_layer=layer, num_layers=2)
    for p in blk.parameters():  # _get_clones deep-copies: make the two layers differ
        with torch.no_grad():
            p.add_(0.02 * torch.randn_like(p))
    g = torch.Generator().manual_seed(12)
    tgt = torch.randn(2, 16, 192, generator=g, requires_grad=True)
    mem = torch.randn(2, 7, 128, generator=g, requires_grad=True)
    kpm = torch.zeros(2, 7, dtype=torch.bool)
    kpm[0, 4:] = True
    out = blk(tgt=tgt, memory=mem, memory_key_padding_mask=kpm)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    fx = dict(state_dict={k: v.detach().clone() for k, v in blk.state_dict().items()}, tgt=tgt.detach(),
              memory=mem.detach(), kpm=kpm, out=out.detach(), dout=dout, dtgt=tgt.grad.clone(), dmem=mem.grad.clone(),
              grads={k: p.grad.clone() for k, p in blk.named_parameters()})
    torch.save(fx, OUT / "decoder_d96.pt")
    print("decoder_d96: out", tuple(out.shape), "keys", len(fx["state_dict"]))


def make_ops(R):
    """Per-op vectors: interpolation, losses, pooled text output, pos-embedding resize."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(21)
    fx = {}
    x = torch.randn(2, 5, 6, 6, generator=g)
    fx["interp_in"] = x
    fx["bicubic_x4"] = F.interpolate(x, mode="bicubic", scale_factor=4)
    fx["bilinear_x2"] = F.interpolate(x, mode="bilinear", scale_factor=2)
    fx["bilinear_x4"] = F.interpolate(x, mode="bilinear", scale_factor=4)
    fx["bilinear_size20"] = F.interpolate(x, mode="bilinear", size=20)
    fx["bicubic_size9"] = F.interpolate(x, mode="bicubic", size=9)
    # losses (model/loss.py)
    logits = torch.randn(2, 151, 8, 8, generator=g)
    labels = torch.randint(0, 151, (2, 8, 8), generator=g)
    fx["ce_logits"], fx["ce_labels"] = logits, labels
    fx["ce"] = nn.CrossEntropyLoss()(logits, labels)
    low = torch.randn(2, 151, 4, 4, generator=g)
    lab16 = torch.randint(0, 151, (2, 16, 16), generator=g)
    fx["aux_in"], fx["aux_labels"] = low, lab16
    fx["aux"] = R["rloss"].AuxiliaryLoss()(low, lab16)
    outs = torch.randn(2, 64, 151, generator=g)
    fx["con_in"] = outs
    c = R["rloss"].ContrastiveLoss()(outs, labels)
    fx["con"] = torch.stack([c[0], c[1], c[2]])
    a, p, n = torch.randn(6, 16, generator=g), torch.randn(3, 16, generator=g), torch.randn(5, 16, generator=g)
    fx["np_x"], fx["np_pos"], fx["np_neg"] = a, p, n
    fx["npair"] = R["rloss"].NPairLoss()(a, p, n)
    # pooled text (TextEncoderCLIPPooler, model/encoder.py:104-116)
    torch.manual_seed(5)
    tcfg = R["CLIPTextConfig"](vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                               num_attention_heads=1, max_position_embeddings=77, eos_token_id=299, bos_token_id=298,
                               pad_token_id=299)
    te = _bare(R["renc"].TextEncoderCLIPPooler)
    te.patch_size = 16
    te.enc = R["CLIPTextModel"](tcfg)
    te.eval()
    ids = torch.randint(1, 297, (3, 9), generator=g)
    ids[:, 0] = 298
    ids[0, 4] = 299
    ids[1, 8] = 299
    ids[2, 2] = 299
    mask = torch.ones(3, 9, dtype=torch.int64)
    mask[0, 5:] = 0
    mask[2, 3:] = 0
    with torch.no_grad():
        fx["pool_sd"] = {k: v.clone() for k, v in te.state_dict().items()}
        fx["pool_ids"], fx["pool_mask"] = ids, mask
        fx["pool_out"] = te(input_ids=ids, attention_mask=mask)
    # pos-embedding resize (ImageEncoderCLIP.pos_emebedding_interpolate, model/encoder.py:32-44), 14x14 -> 8x8
    torch.manual_seed(6)
    vcfg = R["CLIPVisionConfig"](hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1,
                                 image_size=224, patch_size=16)
    ve = _bare(R["renc"].ImageEncoderCLIP)
    ve.in_size, ve.patch_size = 128, 16
    inner = R["CLIPVisionModel"](vcfg)

    class _Legacy(nn.Module):  # the reference addresses `enc.vision_model.embeddings` (pre-5.x layout)
        def __init__(self, vm):
            super().__init__()
            self.vision_model = vm

    ve.enc = _Legacy(inner)
    with torch.no_grad():
        fx["posint_in"] = inner.embeddings.position_embedding.weight.detach().clone()
        fx["posint_out"] = ve.pos_emebedding_interpolate(tgt_size=8).detach().clone()
    torch.save(fx, OUT / "ops.pt")
    print("ops:", sorted(fx.keys()))


def _patch_sr_is_causal(cls):
    """torch 2.10 calls `_sa_block(x, mask, kpm, is_causal)`; the reference's overrides take 3 arguments
    (SURVEY.md §2 drift #2).  Harness-side adapter: accept and drop `is_causal`; the method body stays the reference's."""
    orig = cls._sa_block
    if getattr(orig, "_adapted", False):
        return

    def adapted(self, x, attn_mask, key_padding_mask, is_causal=False):
        return orig(self, x, attn_mask, key_padding_mask)

    adapted._adapted = True
    cls._sa_block = adapted


def make_hier(R):
    """HierarchicalCrossA / HierarchicalSelfA (model/hierarchical.py) and FTNDecoder (model/decoder.py:36-94) at
    reduced dims, weights from tests/golden_util.make_weights (not stored), forward + backward."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights
    import model.hierarchical as rh
    for cls in (rh.SRTransformerCrossA, rh.SRTransformerSelfA, R["rdec"].SRTransformerDecoder):
        _patch_sr_is_causal(cls)
    in_dims, dim, nhead = [64, 128, 192, 256], 128, 2
    g = torch.Generator().manual_seed(31)
    B, K = 2, 10
    visual = [torch.randn(B, p, c, generator=g) for p, c in zip((256, 64, 16, 4), in_dims)]
    textual = torch.randn(B, K, dim, generator=g)
    fx = dict(visual0=visual[0], visual1=visual[1], visual2=visual[2], visual3=visual[3], textual=textual)
    specs = dict(
        cross=(lambda: rh.HierarchicalCrossA(in_dims, [2, 1, 1], dim, nhead=nhead, dropout=0, batch_first=True), True, 41),
        selfa=(lambda: rh.HierarchicalSelfA(in_dims, [1, 1, 2], dim, nhead=nhead, dropout=0, batch_first=True), False, 42),
        ftn=(lambda: R["rdec"].FTNDecoder(in_dims, dim, dropout=0), True, 43),
    )
    for name, (ctor, uses_text, seed) in specs.items():
        m = ctor().train()
        params = dict(m.named_parameters())  # shared layers appear once
        shapes = {k: list(v.shape) for k, v in params.items()}
        w = make_weights(shapes, seed)
        with torch.no_grad():
            for k, p in params.items():
                p.copy_(w[k])
        vis = [v.clone().requires_grad_(True) for v in visual]
        txt = textual.clone().requires_grad_(True)
        out = m(vis, txt) if uses_text else m(vis)
        dout = torch.randn(out.shape, generator=g)
        out.backward(dout)
        fx[name] = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, seed=torch.tensor(seed), out=out.detach(),
                        dout=dout, dvisual0=vis[0].grad.clone(), dvisual3=vis[3].grad.clone(),
                        dvisual1_absmax=vis[1].grad.abs().max() if vis[1].grad is not None else torch.tensor(0.0),
                        dtextual=txt.grad.clone() if uses_text else torch.zeros(1),
                        grad_stats={k: torch.stack([p.grad.sum(), p.grad.abs().sum()]) for k, p in params.items()},
                        grad_full={k: params[k].grad.clone() for k in list(params)[:0] +
                                   [kk for kk in params if kk.endswith("sr.weight") or kk.endswith("linear_stage_3.weight")
                                    or kk.endswith("norm.weight")][:6]})
    # score-map tail (model/final.py:350-356; final.py itself cannot be imported — un-vendored DenseCLIP — so the
    # same torch calls are issued here)
    import torch.nn.functional as F
    from einops import rearrange
    ve = torch.randn(2, 64, 128, generator=g, requires_grad=True)
    te = torch.randn(2, 10, 128, generator=g, requires_grad=True)
    v = rearrange(ve, "b (h w) c -> b c h w", h=8)
    v = F.normalize(v, dim=1, p=2)
    t = F.normalize(te, dim=2, p=2)
    sm = F.interpolate(torch.einsum("bchw,bkc->bkhw", v, t), mode="bilinear", scale_factor=4)
    lab = torch.randint(0, 10, (2, 32, 32), generator=g)
    loss = nn.CrossEntropyLoss()(sm, lab)
    loss.backward()
    fx["tail"] = dict(ve=ve.detach(), te=te.detach(), score=sm.detach(), labels=lab, loss=loss.detach(), dve=ve.grad.clone(),
                      dte=te.grad.clone())
    torch.save(fx, OUT / "hier_tiny.pt")
    print("hier:", {k: (tuple(v["out"].shape) if isinstance(v, dict) and "out" in v else None) for k, v in fx.items()})


def make_ftn(R):
    """ftn.Decoder (model/ftn.py:67-129) at its hard-coded dims (Swin-base widths, grids 128/64/32/16), B=1, eval mode
    (the reference hard-codes torch's default dropout 0.1), forward + backward.  Inputs, weights and the output
    gradient are regenerated from seeds by the tests; the fixture keeps every 61st row of the big tensors."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import ftn_inputs, make_weights
    import model.ftn as rftn
    m = rftn.Decoder().eval()
    params = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in params.items()}
    w = make_weights(shapes, 51)
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(w[k])
    xs, dout = ftn_inputs(52)
    xs = [x.requires_grad_(True) for x in xs]
    out = m(xs)
    out.backward(dout)
    st = 61
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(51), xseed=torch.tensor(52),
              stride=torch.tensor(st), out=out.detach()[:, ::st].clone(),
              dx=[x.grad[:, ::st].clone() for x in xs],
              grad_stats={k: (torch.stack([p.grad.sum(), p.grad.abs().sum()]) if p.grad is not None else torch.zeros(2))
                          for k, p in params.items()},
              no_grad=[k for k, p in params.items() if p.grad is None],
              grad_full={k: params[k].grad.clone() for k in params
                         if params[k].grad is not None and params[k].numel() <= 1024 and ("norm" in k or k.endswith("bias"))})
    fx["no_grad"] = torch.tensor([list(params).index(k) for k in fx["no_grad"]])
    torch.save(fx, OUT / "ftn_decoder.pt")
    print("ftn:", tuple(out.shape), "params without grad:", len(fx["no_grad"]), "fixture rows:", fx["out"].shape[1])


def make_swin(R):
    """SwinTransformer.forward (model/encoder.py:121-131) with a config-built tiny SwinModel (hub constructor cannot run
    offline): embed 32, heads [1,2,4,8] (head_dim 32 like every Swin), window 5 on a 44x44 grid so that padding to the
    window (44->45, 22->25, 11->15), the cyclic shift + region mask, and the odd-size patch merging (11->12) all occur.
    Forward + backward through hidden_states[:4]."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights
    from transformers import SwinConfig, SwinModel
    cfg = SwinConfig(image_size=176, patch_size=4, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=5,
                     drop_path_rate=0.0)
    cfg._attn_implementation = "eager"
    m = _bare(R["renc"].SwinTransformer)
    m.encoder = SwinModel(cfg)
    m.train()
    params = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in params.items()}
    w = make_weights(shapes, 61)
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(w[k])
    g = torch.Generator().manual_seed(62)
    x = torch.randn(2, 3, 176, 176, generator=g)
    outs = m(x)                                   # the reference's own forward
    assert len(outs) == 4
    douts = [torch.randn(o.shape, generator=g) * 0.2 for o in outs]
    sum((o * d).sum() for o, d in zip(outs, douts)).backward()
    names = list(params)
    keep = [k for k in names if params[k].grad is not None and
            (k.endswith("relative_position_bias_table") or "norm" in k or k.endswith("bias") or ".blocks.0.attention.q_proj.weight" in k
             or k.endswith("reduction.weight") or k.endswith("projection.weight") or ".blocks.1.mlp.fc1.weight" in k)]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(61), pixel_values=x,
              outs=[o.detach() for o in outs], douts=douts,
              grad_stats={k: (torch.stack([p.grad.sum(), p.grad.abs().sum()]) if p.grad is not None else torch.zeros(2))
                          for k, p in params.items()},
              no_grad=torch.tensor([i for i, k in enumerate(names) if params[k].grad is None]),
              grad_full={k: params[k].grad.clone() for k in keep})
    torch.save(fx, OUT / "swin_tiny.pt")
    print("swin:", [tuple(o.shape) for o in outs], "params without grad:", len(fx["no_grad"]), "full grads:", len(keep))


def make_swin_droppath(R):
    """The same tiny SwinModel as make_swin in TRAINING mode with drop_path_rate = 0.2 and KNOWN per-sample decisions:
    SwinDropPath.forward is replaced, for the reference's own forward, by a multiply with the next prepared [B] vector of
    keep / (1 - p_block) (p_block from the module's own drop_prob).  Pins where stochastic depth acts and its rates."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights
    from transformers import SwinConfig, SwinModel
    from transformers.models.swin import modeling_swin as MS
    cfg = SwinConfig(image_size=176, patch_size=4, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=5,
                     drop_path_rate=0.2)
    cfg._attn_implementation = "eager"
    m = _bare(R["renc"].SwinTransformer)
    m.encoder = SwinModel(cfg)
    m.train()
    params = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in params.items()}
    w = make_weights(shapes, 61)
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(w[k])
    from golden_util import swin_droppath_inputs
    x, keeps, douts = swin_droppath_inputs(63, 4)              # keeps: one row per SwinDropPath call (40 % dropped)
    rates, used = [], []
    real = MS.SwinDropPath.forward

    def fake(self, hidden_states):
        i = len(used)
        used.append(float(self.drop_prob))
        return hidden_states * (keeps[i] / (1.0 - self.drop_prob)).view(-1, 1, 1)

    MS.SwinDropPath.forward = fake
    try:
        outs = m(x)
        assert [tuple(o.shape) for o in outs] == [tuple(d.shape) for d in douts]
        sum((o * d).sum() for o, d in zip(outs, douts)).backward()
    finally:
        MS.SwinDropPath.forward = real
    keep_g = ["encoder.encoder.layers.0.blocks.1.attention.q_proj.weight", "encoder.encoder.layers.2.blocks.1.attention.o_proj.weight",
              "encoder.embeddings.patch_embeddings.projection.weight", "encoder.encoder.layers.1.blocks.0.mlp.fc1.weight"]
    fx = dict(iseed=torch.tensor(63), ncalls=torch.tensor(len(used)), rates=torch.tensor(used), outs=[o.detach() for o in outs],
              grad_full={k: params[k].grad.clone() for k in keep_g})
    torch.save(fx, OUT / "swin_droppath.pt")
    print("swin drop-path: calls", len(used), "rates", [round(r, 4) for r in used])



def make_prompt(R):
    """PromptDecoder(PromptLayer(512, 1024, 8, batch_first=True), 2) — the reference's construction at model/model.py:183
    (post-norm, dropout 0.1 default, d_kv != d_model), here 2 layers deep — in EVAL mode (dropout inactive; identical
    numbers to a dropout=0 training forward): forward + backward on tgt [2,150,512] (K text embeddings) and memory
    [2,256,1024] (visual tokens).  Weights and inputs are regenerated from seeds (tests/golden_util.py), so the fixture
    holds only the reference's outputs and gradients."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights, prompt_inputs
    torch.manual_seed(31)
    dec = R["rdec"].PromptDecoder(R["rdec"].PromptLayer(d_model=512, d_kv=1024, nhead=8, batch_first=True), num_layers=2)
    named = dict(dec.named_parameters())
    shapes = {k: list(v.shape) for k, v in named.items()}
    w = make_weights(shapes, 32)
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    dec.eval()
    assert dec.layers[0].dropout.p == 0.1 and dec.layers[0].norm_first is False
    tgt, mem, dout = prompt_inputs(33)
    tgt.requires_grad_(True)
    mem.requires_grad_(True)
    out = dec(tgt=tgt, memory=mem)
    out.backward(dout)
    grads = {k: p.grad.detach().clone() for k, p in named.items()}
    keep = ["layers.0.norm1.weight", "layers.1.norm3.weight", "layers.0.multihead_attn.in_proj_bias",
            "layers.1.multihead_attn.out_proj.bias", "layers.1.self_attn.out_proj.weight",
            "layers.0.multihead_attn.q_proj_weight"]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(32), iseed=torch.tensor(33),
              out=out.detach().clone(), dtgt=tgt.grad.clone(), dmem_rows=mem.grad[:, ::4].clone(),
              grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items()},
              grad_full={k: grads[k] for k in keep})
    torch.save(fx, OUT / "prompt_decoder.pt")
    print("prompt_decoder: out", tuple(out.shape), "params", len(shapes), sorted(shapes)[:4])


def make_clip_full(R):
    """ImageEncoderCLIPFull.forward (model/encoder.py:67-68: last_hidden_state WITH the CLS row) at tiny dims: forward +
    backward with an upstream gradient that is non-zero on the CLS row."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import clip_full_inputs, make_weights
    torch.manual_seed(41)
    vcfg = R["CLIPVisionConfig"](hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                 image_size=64, patch_size=16)
    enc = _bare(R["renc"].ImageEncoderCLIPFull)
    enc.in_size, enc.patch_size = 64, 16
    enc.enc = R["CLIPVisionModel"](vcfg)
    named = dict(enc.named_parameters())
    shapes = {k: list(v.shape) for k, v in named.items()}
    w = make_weights(shapes, 42)
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(w[k])
    enc.train()
    pix, dout = clip_full_inputs(43)
    out = enc(pixel_values=pix)
    assert tuple(out.shape) == (2, 17, 128)
    out.backward(dout)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in named.items()}
    keep = ["enc.embeddings.class_embedding", "enc.embeddings.position_embedding.weight",
            "enc.embeddings.patch_embedding.weight", "enc.encoder.layers.0.self_attn.k_proj.weight",
            "enc.encoder.layers.1.mlp.fc2.weight", "enc.pre_layrnorm.weight"]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(42), iseed=torch.tensor(43),
              out=out.detach().clone(),
              grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items() if v is not None},
              no_grad=[k for k, v in grads.items() if v is None],
              grad_full={k: grads[k] for k in keep})
    torch.save(fx, OUT / "clip_full_tiny.pt")
    print("clip_full: out", tuple(out.shape), "no grad:", fx["no_grad"])



def make_dropout(R):
    """Training-mode dropout of the reference's decoder layers (PromptLayer: post-norm, model/decoder.py:24-28; DecoderLayer
    with norm_first=True, :9-13) with KNOWN masks: torch.nn.functional.dropout is replaced, for the duration of the
    reference's own forward, by a function that multiplies by the next prepared keep/(1-p) tensor; the attention modules
    are asked for need_weights=True so that torch takes its explicit softmax -> dropout -> matmul path
    (torch:nn/functional.py multi_head_attention_forward) instead of the fused SDPA kernel, whose mask is not observable.
    Pins WHERE the six dropouts act; the fixture stores the reference's outputs and gradients only."""
    import torch.nn.functional as F
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import DROP_SITES, dropout_case, make_weights
    fx = {}
    for tag, cls, nf in (("post", R["rdec"].PromptLayer, False), ("pre", R["rdec"].DecoderLayer, True)):
        torch.manual_seed(51)
        layer = cls(d_model=128, d_kv=192, nhead=2, dim_feedforward=256, dropout=0.25, batch_first=True, norm_first=nf)
        dec = (R["rdec"].PromptDecoder if not nf else R["rdec"].DecoderBlock)(layer, num_layers=2)
        named = dict(dec.named_parameters())
        shapes = {k: list(v.shape) for k, v in named.items()}
        w = make_weights(shapes, 52)
        with torch.no_grad():
            for k, p_ in named.items():
                p_.copy_(w[k])
        dec.train()
        tgt, mem, dout, drops = dropout_case(53 + int(nf), nf)
        queue = [d[s_] for d in drops for s_ in DROP_SITES]
        for l in dec.layers:       # explicit-softmax path of torch's MHA (the dropout call is then a Python-level F.dropout)
            for mha in (l.self_attn, l.multihead_attn):
                orig = mha.forward
                mha.forward = (lambda o: (lambda *a, **kw: o(*a, **{**kw, "need_weights": True})))(orig)
        real = F.dropout

        def fake(input, p=0.5, training=True, inplace=False):
            assert training and abs(p - 0.25) < 1e-9
            m = queue.pop(0)
            return input * m.reshape(input.shape)

        F.dropout = fake
        try:
            tgt.requires_grad_(True); mem.requires_grad_(True)
            out = dec(tgt=tgt, memory=mem) if not nf else dec(tgt=tgt, memory=mem)
            out.backward(dout)
        finally:
            F.dropout = real
        assert not queue, len(queue)
        grads = {k: p_.grad.detach().clone() for k, p_ in named.items()}
        fx[tag] = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, out=out.detach().clone(), dtgt=tgt.grad.clone(),
                       dmem=mem.grad.clone(), grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items()},
                       grad_full={k: grads[k] for k in ("layers.0.linear1.weight", "layers.1.multihead_attn.k_proj_weight",
                                                        "layers.0.self_attn.in_proj_weight", "layers.1.norm2.weight")})
        print("dropout", tag, "out", tuple(out.shape))
    torch.save(fx, OUT / "decoder_dropout.pt")




def make_prompt_ftn(R):
    """PromptFTN.forward (model/model.py:186-214) END TO END through the reference's own classes: a config-built tiny Swin
    (embed 32, depths 1-1-1-1, window 7: the 512 x 512 input gives the hard-coded 128 x 128 token grid and stage widths
    32/64/128/256), a config-built 1-layer CLIP text tower of width 512 behind the reference's TextEncoderCLIPPooler (frozen as
    in its constructor), the reference's PromptDecoder(PromptLayer(512, 256, 8, batch_first=True), 2) and FTNDecoder([32,64,128,
    256], 512), both with dropout 0 so that the TRAINING forward/backward is reproducible (the eval forward of the default
    dropout-0.1 construction gives the same numbers).  Only PromptFTN.__init__ is bypassed (it names hub checkpoints).
    Kept: the score map on a stride-8 grid, CE(score_map, labels), and selected gradients."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import make_weights, prompt_ftn_inputs
    from transformers import SwinConfig, SwinModel
    for cls in (R["rdec"].SRTransformerDecoder,):
        _patch_sr_is_causal(cls)
    torch.manual_seed(71)
    m = _bare(R["rmodel"].PromptFTN)
    te = _bare(R["renc"].TextEncoderCLIPPooler)
    te.patch_size = 16
    te.enc = R["CLIPTextModel"](R["CLIPTextConfig"](vocab_size=512, hidden_size=512, intermediate_size=128, num_hidden_layers=1,
                                                    num_attention_heads=8, max_position_embeddings=16, eos_token_id=511,
                                                    bos_token_id=510, pad_token_id=511))
    m.textual_encoder = te
    for prm in m.textual_encoder.parameters():
        prm.requires_grad = False
    ve = _bare(R["renc"].SwinTransformer)
    cfg = SwinConfig(image_size=512, patch_size=4, embed_dim=32, depths=[1, 1, 1, 1], num_heads=[1, 2, 4, 8], window_size=7,
                     drop_path_rate=0.0)
    cfg._attn_implementation = "eager"
    ve.encoder = SwinModel(cfg)
    m.visual_encoder = ve
    m.prompt_decoder = R["rdec"].PromptDecoder(R["rdec"].PromptLayer(d_model=512, d_kv=256, nhead=8, dropout=0.0, batch_first=True),
                                               num_layers=2)
    m.decoder = R["rdec"].FTNDecoder(in_dims=[32, 64, 128, 256], dim=512, dropout=0.0)
    named = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in named.items()}
    w = make_weights(shapes, 72)
    with torch.no_grad():
        for k, prm in named.items():
            prm.copy_(w[k])
    m.train()
    inputs, labels = prompt_ftn_inputs(73)
    none, score_map = m(inputs)                    # the reference's own forward
    assert none is None and tuple(score_map.shape) == (1, 6, 512, 512)
    loss = nn.CrossEntropyLoss()(score_map, labels)
    loss.backward()
    grads = {k: (prm.grad.detach().clone() if prm.grad is not None else None) for k, prm in named.items()}
    keep = ["visual_encoder.encoder.embeddings.patch_embeddings.projection.weight",
            "visual_encoder.encoder.encoder.layers.3.blocks.0.attention.self.query.weight",
            "prompt_decoder.layers.0.multihead_attn.k_proj_weight", "prompt_decoder.layers.1.linear2.weight",
            "decoder.linear2_stage_1.weight", "decoder.linear2_stage_4.weight",
            "decoder.attention_stage_4.2.attention_block.sr.weight", "decoder.attention_stage_2.0.attention_block.linear1.weight"]
    keep = [k for k in keep if grads.get(k) is not None]
    assert len(keep) >= 6, [k for k in named if "stage_4.2" in k][:8]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(72), iseed=torch.tensor(73),
              score_s8=score_map.detach()[:, :, ::8, ::8].clone(), loss=loss.detach().clone(),
              no_grad=[k for k, v in grads.items() if v is None],
              grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items() if v is not None},
              grad_full={k: (grads[k] if grads[k].numel() <= 65536 else grads[k].flatten()[::37].clone()) for k in keep})   # big ones: every 37th element
    torch.save(fx, OUT / "prompt_ftn.pt")
    print("prompt_ftn: loss", float(loss), "params", len(shapes), "frozen", len(fx["no_grad"]), "full grads", keep)


def make_dense_clip(R):
    """DenseClip.forward (model/model.py:122-171) through the reference's own classes at tiny dims: ImageEncoderCLIPFull
    (hidden 256, 2 layers, 64 x 64 / patch 16 -> 17 tokens), frozen TextEncoderCLIPPooler (width 64), TextToPatch(out 128),
    PromptDecoder(PromptLayer(128, 128, 2, dropout 0, batch_first=True), 2) — the reference's ``PromptLayer(d_model=512, nhead=8)``
    lacks the required d_kv and batch_first and cannot be constructed; d_kv = d_model and batch_first=True is the only reading
    its forward admits — and DecoderBlock(DecoderLayer(256, 128, 2, batch_first=True, norm_first=True), 2) (head_dim 64 / 128:
    what the HIP attention kernels take).  Only
    DenseClip.__init__ is bypassed.  Kept: score_map, out, and gradients of loss = <score_map, ds> + <out, do>."""
    sys.path.insert(0, str(ROOT / "tests"))
    from golden_util import dense_clip_inputs, make_weights
    torch.manual_seed(81)
    vcfg = R["CLIPVisionConfig"](hidden_size=256, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                                 image_size=64, patch_size=16)
    tcfg = R["CLIPTextConfig"](vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                               num_attention_heads=1, max_position_embeddings=16, eos_token_id=511, bos_token_id=510,
                               pad_token_id=511)
    m = _bare(R["rmodel"].DenseClip)
    m.patch_size, m.in_size, m.out_size = 16, 64, 16
    m.vision_encoder = _bare(R["renc"].ImageEncoderCLIPFull)
    m.vision_encoder.in_size, m.vision_encoder.patch_size = 64, 16
    m.vision_encoder.enc = R["CLIPVisionModel"](vcfg)
    m.text_encoder = _bare(R["renc"].TextEncoderCLIPPooler)
    m.text_encoder.patch_size = 16
    m.text_encoder.enc = R["CLIPTextModel"](tcfg)
    for prm in m.text_encoder.parameters():
        prm.requires_grad = False
    m.text_patch = R["rtp"].TextToPatch(out=128, img_in=256, text_in=64)
    m.prompt_decoder = R["rdec"].PromptDecoder(R["rdec"].PromptLayer(d_model=128, d_kv=128, nhead=2, dim_feedforward=128, dropout=0.0,
                                                                    batch_first=True), num_layers=2)
    m.vision_decoder = R["rdec"].DecoderBlock(decoder_layer=R["rdec"].DecoderLayer(d_model=256, d_kv=128, nhead=2, dim_feedforward=128,
                                                                                  batch_first=True, norm_first=True), num_layers=2)
    named = dict(m.named_parameters())
    shapes = {k: list(v.shape) for k, v in named.items()}
    w = make_weights(shapes, 82)
    with torch.no_grad():
        for k, prm in named.items():
            prm.copy_(w[k])
    m.train()
    inputs, ds, do = dense_clip_inputs(83)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):          # (the reference's forward prints a shape line, model.py:167)
        none, score_map, out = m(inputs)
    assert none is None and tuple(score_map.shape) == (2, 5, 4, 4) and tuple(out.shape) == (2, 17, 256)
    ((score_map * ds).sum() + (out * do).sum()).backward()
    grads = {k: (prm.grad.detach().clone() if prm.grad is not None else None) for k, prm in named.items()}
    keep = ["vision_encoder.enc.embeddings.class_embedding", "vision_encoder.enc.encoder.layers.1.mlp.fc2.weight",
            "text_patch.visual.weight", "text_patch.textual.weight", "prompt_decoder.layers.1.multihead_attn.in_proj_weight",
            "vision_decoder.layers.0.multihead_attn.k_proj_weight", "vision_decoder.layers.1.linear1.weight"]
    fx = dict(shapes={k: torch.tensor(v) for k, v in shapes.items()}, wseed=torch.tensor(82), iseed=torch.tensor(83),
              score_map=score_map.detach().clone(), out=out.detach().clone(), no_grad=[k for k, v in grads.items() if v is None],
              grad_stats={k: torch.stack([v.sum(), v.abs().sum()]) for k, v in grads.items() if v is not None},
              grad_full={k: grads[k] for k in keep})
    torch.save(fx, OUT / "dense_clip_tiny.pt")
    print("dense_clip: score_map", tuple(score_map.shape), "out", tuple(out.shape), "frozen", len(fx["no_grad"]))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    R = _ref_imports()
    if len(sys.argv) > 1 and sys.argv[1] == "hier":
        make_hier(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ftn":
        make_ftn(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "contrastive":
        make_contrastive(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "swin":
        make_swin(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dropout":
        make_dropout(R)
        make_swin_droppath(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "compose":
        make_prompt_ftn(R)
        make_dense_clip(R)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prompt":
        make_prompt(R)
        make_clip_full(R)
        return
    make_base_tiny(R)
    make_decoder_d96(R)
    make_ops(R)
    make_hier(R)
    make_ftn(R)
    make_swin(R)
    make_contrastive(R)
    make_prompt(R)
    make_clip_full(R)
    make_dropout(R)
    make_swin_droppath(R)
    make_prompt_ftn(R)
    make_dense_clip(R)
    # the reference's only data fixture on this path — copied as-is
    protos = torch.load(REF / "model" / "ade20k_prototypes.pt", weights_only=True)
    torch.save(protos.clone(), OUT / "ade20k_prototypes.pt")
    for f in sorted(OUT.glob("*.pt")):
        print(f.name, f.stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
