"""Full-dimension pin of the CPU oracle: the REFERENCE's BaseModelWithText at the real ViT-B/16 + CLIP-text
dims (157 M parameters, config-built and seeded — the hub constructors cannot run offline), in_size 128, B=1, against
oracle.base_model_with_text on the same state_dict.  Runs in the build container only (needs /root/reference); the weights
are too large to commit, so this is a script with a printed verdict, not a fixture.

    python tools/check_full_dims.py        ->  logits max|diff|, rel-L2, CE loss of both
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import torch  # noqa: E402
from torch import nn  # noqa: E402

from make_golden import REF, _bare, _ref_imports  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402


def main():
    R = _ref_imports()
    torch.manual_seed(1024)
    in_size, out_size = 128, 32
    m = _bare(R["rmodel"].BaseModelWithText)
    m.patch_size, m.in_size, m.out_size = 16, in_size, out_size
    m.vision_encoder = _bare(R["renc"].ImageEncoderCLIP)
    m.vision_encoder.in_size, m.vision_encoder.patch_size = in_size, 16
    m.vision_encoder.enc = R["CLIPVisionModel"](R["CLIPVisionConfig"](image_size=in_size, patch_size=16))   # ViT-B/16 defaults
    m.text_encoder = _bare(R["renc"].TextEncoderCLIP)
    m.text_encoder.patch_size = 16
    m.text_encoder.enc = R["CLIPTextModel"](R["CLIPTextConfig"]())                                         # CLIP-B text defaults
    m.class_prototypes = nn.Parameter(torch.load(REF / "model" / "ade20k_prototypes.pt", weights_only=True).clone())
    layer = R["rdec"].DecoderLayer(d_model=768, d_kv=512, nhead=8, dropout=0, batch_first=True, norm_first=True)
    m.vision_decoder = R["rdec"].DecoderBlock(decoder_layer=layer, num_layers=1)
    m.pixel_patch = R["rtp"].TextToPatch(out=512, img_in=768, text_in=512)
    m.eval()
    print("reference parameters: %.2f M" % (sum(p.numel() for p in m.parameters()) / 1e6))
    g = torch.Generator().manual_seed(1000)
    inputs = dict(pixel_values=torch.randn(1, 3, in_size, in_size, generator=g),
                  input_ids=torch.cat([torch.tensor([[49406]]), torch.randint(1, 49405, (1, 10), generator=g),
                                       torch.full((1, 5), 49407)], dim=1),
                  attention_mask=torch.cat([torch.ones(1, 12, dtype=torch.long), torch.zeros(1, 4, dtype=torch.long)], dim=1))
    labels = torch.randint(0, 151, (1, out_size, out_size), generator=g)
    t0 = time.time()
    with torch.no_grad():
        _, _, ref = m(inputs)
    t_ref = time.time() - t0
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    cfg = O.BaseCfg(in_size=in_size, out_size=out_size, patch=16, vision=O.ClipCfg(768, 12, 12, patch=16),
                    text=O.ClipCfg(512, 8, 12), dec_heads=8, dec_layers=1)
    t0 = time.time()
    with torch.no_grad():
        _, _, mine = O.base_model_with_text(sd, inputs, cfg)
    t_or = time.time() - t0
    rel = float((mine - ref).norm() / ref.norm())
    print(f"logits {tuple(ref.shape)}: max|diff| {float((mine - ref).abs().max()):.3e}  rel-L2 {rel:.3e}   "
          f"(reference {t_ref:.1f}s, oracle {t_or:.1f}s)")
    ce_ref, ce_or = float(nn.CrossEntropyLoss()(ref, labels)), float(O.cross_entropy(mine, labels))
    print(f"cross-entropy: reference {ce_ref:.6f}  oracle {ce_or:.6f}")
    ok = rel < 1e-4 and abs(ce_ref - ce_or) < 1e-4
    print("PINNED at full dims" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
