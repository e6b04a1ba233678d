"""Edge cases of the hot path against the CPU oracle: batch of one, the longest text (77 tokens), no padding / nearly
all padding, few classes, ignored labels, minimal operator shapes, and the error behaviour the reference has."""
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _tiny(dev, K=151):
    import lc2is_amd.nn as N
    from oracle import ref_cpu as O
    torch.manual_seed(3)
    protos = torch.randn(K, 64) * 0.3
    m = N.BaseModelWithText(16, 64, 16, vision_arch=N.ClipArch(128, 2, 2, 256),
                            text_arch=N.ClipArch(64, 1, 2, 128, vocab=512, eos_token_id=511), nhead=2, dim_feedforward=128,
                            out_dim=64, prototypes=protos)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.BaseCfg(in_size=64, out_size=16, patch=16, vision=O.ClipCfg(128, 2, 2, patch=16),
                    text=O.ClipCfg(64, 1, 2, eos_token_id=511), dec_heads=2, dec_layers=1)
    return m.to(dev).train(), sd, cfg, O


@pytest.mark.parametrize("B,L,pad,K", [(1, 77, 0, 151), (1, 5, 4, 151), (3, 16, 0, 7), (2, 77, 70, 151)])
def test_model_edges_vs_oracle(dev, B, L, pad, K):
    m, sd, cfg, O = _tiny(dev, K)
    g = torch.Generator().manual_seed(B * 100 + L)
    ids = torch.randint(1, 500, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    if pad:
        mask[:, L - pad:] = 0
        ids[:, L - pad:] = 511
    inputs = dict(pixel_values=torch.randn(B, 3, 64, 64, generator=g), input_ids=ids, attention_mask=mask)
    labels = torch.randint(0, K, (B, 16, 16), generator=g)
    labels[:, :2] = -100                                            # ignored pixels (nn.CrossEntropyLoss default)
    _, _, ref = O.base_model_with_text(sd, inputs, cfg)
    ref_loss = O.cross_entropy(ref, labels)
    dinputs = {k: v.to(dev) for k, v in inputs.items()}
    out = m(dinputs)["outputs"]
    assert out.shape == (B, K, 16, 16)
    assert _rel(out, ref) < 2e-2
    loss = m.forward_loss(dinputs, labels.to(dev))
    assert abs(float(loss.detach()) - float(ref_loss)) < 2e-2
    loss.backward()
    assert m.class_prototypes.grad is not None and torch.isfinite(m.class_prototypes.grad).all()


def test_reference_error_behaviour(dev):
    m, _, _, _ = _tiny(dev)
    ok = dict(pixel_values=torch.randn(1, 3, 64, 64, device=dev), input_ids=torch.ones(1, 4, dtype=torch.long, device=dev),
              attention_mask=torch.ones(1, 4, dtype=torch.long, device=dev))
    with pytest.raises(ValueError):                                  # hf CLIPVisionEmbeddings: image size mismatch
        m({**ok, "pixel_values": torch.randn(1, 3, 48, 48, device=dev)})
    with pytest.raises(ValueError):                                  # hf CLIPTextEmbeddings: longer than max_position_embeddings
        m({**ok, "input_ids": torch.ones(1, 78, dtype=torch.long, device=dev), "attention_mask": torch.ones(1, 78, dtype=torch.long, device=dev)})
    with pytest.raises(RuntimeError):                                # no CPU path
        m({k: v.cpu() for k, v in ok.items()})


def test_minimal_operator_shapes(dev):
    from lc2is_amd import ops
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1, 64, generator=g).bfloat16().to(dev)
    w = torch.randn(4, 64, generator=g).bfloat16().to(dev)
    _, of, _ = ops.gemm_nt(a, w, None, out_bf16=None, out_f32=True)                        # M = 1, N = 4
    assert _rel(of, a.double().cpu() @ w.double().cpu().T) < 1e-5
    dw = ops.gemm_tn(torch.randn(1, 8, generator=g).bfloat16().to(dev), torch.randn(1, 8, generator=g).bfloat16().to(dev))
    assert dw.shape == (8, 8) and torch.isfinite(dw).all()                                 # one token
    x = torch.randn(1, 64, generator=g).to(dev)
    y, _, mean, rstd = ops.layernorm_fwd(x, torch.ones(64, device=dev), torch.zeros(64, device=dev), 1e-5)
    assert _rel(y.float(), torch.nn.functional.layer_norm(x.cpu(), (64,))) < 1e-2         # one row
    q = torch.randn(1, 64, generator=g).bfloat16().to(dev)
    o, lse = ops.attention_fwd(q, q, q, 1, 1, 1, 1, 64, 0.125)                              # one query, one key
    assert _rel(o.float(), q.float()) < 1e-2
    for bad in (lambda: ops.gemm_nt(a, torch.zeros(4, 32, dtype=torch.bfloat16, device=dev)),            # K mismatch
                lambda: ops.gemm_nt(a.float(), w),                                                        # wrong dtype
                lambda: ops.layernorm_fwd(torch.randn(2, 6, device=dev), torch.ones(6, device=dev), None)):  # C % 4
        with pytest.raises(RuntimeError):
            bad()
