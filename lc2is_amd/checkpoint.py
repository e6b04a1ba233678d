"""Checkpoint interchange with the reference.

* ``Engine.save`` (engine.py:186-190) writes ``torch.save(model.state_dict(), out_dir + "checkpoints/step-N.pt")``:
  :func:`save_checkpoint` / :func:`load_checkpoint` use the same file name and a plain state dict, so files move both
  ways (the modules keep the reference's parameter names; legacy ``enc.vision_model.`` / ``enc.text_model.`` prefixes,
  224-pixel position tables and transformers-4.x Swin names are rewritten by the modules' own load hooks).
* The reference builds its towers with ``from_pretrained("openai/clip-vit-base-patch16")`` etc. (model/encoder.py:19-21,
  94-96, 126-127) — a network fetch.  :func:`load_pretrained_dir` fills the same towers from a LOCAL directory in the
  hub layout (``model.safetensors`` or ``pytorch_model.bin``), nothing is downloaded.

Every file is read with loaders that execute nothing from it (safetensors, ``torch.load(weights_only=True)``).
"""
from __future__ import annotations

import json
from pathlib import Path

import torch
from torch import nn


def _read_state(path: Path) -> dict:
    path = Path(path)
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise TypeError(f"{path}: expected a state dict")
    return sd.get("state_dict", sd) if all(isinstance(k, str) for k in sd) else sd


def save_checkpoint(model: nn.Module, out_dir: str | Path, train_step: int) -> Path:
    """engine.py:186-190 — ``<out_dir>/checkpoints/step-<N>.pt`` holding ``model.state_dict()``."""
    d = Path(out_dir) / "checkpoints"
    d.mkdir(parents=True, exist_ok=True)
    f = d / f"step-{train_step}.pt"
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
    # the dropout stream is not part of the reference's file (it would break the interchange): a JSON sidecar carries it, so a
    # resumed run draws the masks the uninterrupted run would have drawn
    from .nn.base import DropoutRng
    if DropoutRng.get_state() is not None:
        (d / f"step-{train_step}.rng.json").write_text(json.dumps({"dropout_rng_state": DropoutRng.get_state()}))
    return f


def load_checkpoint(model: nn.Module, path: str | Path, strict: bool = True):
    """Load a reference ``step-N.pt`` (or one written by :func:`save_checkpoint`) into a drop-in model."""
    result = model.load_state_dict(_read_state(Path(path)), strict=strict)
    side = Path(path).with_suffix(".rng.json")
    if side.is_file():
        from .nn.base import DropoutRng
        DropoutRng.set_state(json.loads(side.read_text())["dropout_rng_state"])
    for m in model.modules():                      # bf16 weight shadows are rebuilt on the next forward
        if hasattr(m, "invalidate_shadows"):
            m.invalidate_shadows()
    return result


def _hub_file(directory: Path) -> Path:
    for name in ("model.safetensors", "pytorch_model.bin"):
        if (directory / name).is_file():
            return directory / name
    raise FileNotFoundError(f"{directory}: no model.safetensors / pytorch_model.bin (hub snapshot layout expected)")


def load_pretrained_dir(tower: nn.Module, directory: str | Path, strict: bool = True):
    """Fill one tower from a local hub-format snapshot.

    ``ImageEncoderCLIP`` / ``ImageEncoderCLIPFull`` take the ``vision_model.*`` keys of a CLIP checkpoint,
    ``TextEncoderCLIP`` / ``TextEncoderCLIPPooler`` the ``text_model.*`` keys (projection heads and ``logit_scale`` are
    not part of the reference's towers and are skipped), ``SwinTransformer`` the ``SwinModel`` keys (with or without the
    ``swin.`` prefix of classification checkpoints; the classifier head is skipped)."""
    from .nn.clip import ImageEncoderCLIP, TextEncoderCLIP
    from .nn.swin import SwinTransformer
    sd = _read_state(_hub_file(Path(directory)))
    if isinstance(tower, SwinTransformer):
        out = {}
        for k, v in sd.items():
            k = k[len("swin."):] if k.startswith("swin.") else k
            if k.startswith(("classifier.", "pooler.")):
                continue
            out["encoder." + k] = v
    elif isinstance(tower, (ImageEncoderCLIP, TextEncoderCLIP)):
        want = "vision_model." if isinstance(tower, ImageEncoderCLIP) else "text_model."
        out = {"enc." + k: v for k, v in sd.items() if k.startswith(want)}      # the modules' hooks strip the inner prefix
        if not out:
            raise KeyError(f"{directory}: no '{want}*' keys — not a CLIP checkpoint?")
        out = {k: v for k, v in out.items() if not k.endswith("position_ids")}
    else:
        raise TypeError(f"load_pretrained_dir: unsupported tower {type(tower).__name__}")
    result = tower.load_state_dict(out, strict=strict)
    if hasattr(tower, "invalidate_shadows"):
        tower.invalidate_shadows()
    return result
