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


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _sidecar(f: Path, rank: int, world: int) -> Path:
    """One dropout-stream sidecar PER RANK: ``step-N.rng.json`` in a single process, ``step-N.rng.rank<r>.json`` under
    data parallelism (each rank writes and reads only its own: no collective anywhere in save or load)."""
    return f.with_suffix(".rng.json") if world == 1 else f.with_name(f"{f.stem}.rng.rank{rank}.json")


def save_checkpoint(model: nn.Module, out_dir: str | Path, train_step: int, write_model: bool | None = None) -> Path:
    """engine.py:186-190 — ``<out_dir>/checkpoints/step-<N>.pt`` holding ``model.state_dict()``.

    A plain write like the reference's ``Engine.save`` — **no collective, no barrier**: the usual ``if rank == 0:
    save_checkpoint(...)`` of a data-parallel caller works, and so does calling it on every rank.  The replicas hold the same
    parameters, so only rank 0 writes ``step-N.pt`` (``write_model`` overrides: True / False); the dropout stream, however,
    differs per rank (``DropoutRng`` derives it from the torch seed AND the rank), so every rank THAT CALLS writes its own
    small JSON sidecar.  A rank that did not call has no sidecar and resumes on the stream derived from its seed and rank."""
    _, rank, world = _dist()
    d = Path(out_dir) / "checkpoints"
    f = d / f"step-{train_step}.pt"
    d.mkdir(parents=True, exist_ok=True)
    if write_model if write_model is not None else rank == 0:
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
    # the dropout stream is not part of the reference's file (it would break the interchange): a JSON sidecar carries it, so a
    # resumed run draws the masks the uninterrupted run would have drawn — on every rank that saved
    from .nn.base import DropoutRng
    side, st = _sidecar(f, rank, world), DropoutRng.get_state()
    if st is not None:
        side.write_text(json.dumps({"train_step": int(train_step), "world_size": world, "rank": rank,
                                    "dropout_rng_state": {str(rank): st}}))
    elif side.exists():
        side.unlink()                          # never leave a stale stream beside a fresh step-N.pt
    return f


def _sidecar_state(side: Path, stem: str, rank: int, world: int):
    """The dropout-stream state a sidecar holds for (this rank, this world size, this train step), or None.  Accepted forms:
    the current one; round 4's single file with one entry per rank; and — in a single process only — the first format, a bare
    ``{"dropout_rng_state": <int>}`` without step or world size.  Anything malformed or stale is ignored."""
    try:
        meta = json.loads(side.read_text())
        st = meta["dropout_rng_state"]
        if isinstance(st, int):                                    # first format: no step / world size recorded
            return st if (world == 1 and "train_step" not in meta) else None
        if stem != f"step-{int(meta['train_step'])}" or int(meta["world_size"]) != world:
            return None
        return int(st[str(rank)])
    except (OSError, ValueError, KeyError, TypeError):
        return None


def load_checkpoint(model: nn.Module, path: str | Path, strict: bool = True):
    """Load a reference ``step-N.pt`` (or one written by :func:`save_checkpoint`) into a drop-in model.  This rank's
    dropout-stream sidecar is adopted only when it names the same train step as the file and this world size (a sidecar left
    behind by another run beside a reference ``step-N.pt`` is ignored: the stream then stays the one derived from the torch
    seed and the rank).  No collective: every rank reads only files."""
    path = Path(path)
    result = model.load_state_dict(_read_state(path), strict=strict)
    _, rank, world = _dist()
    from .nn.base import DropoutRng
    for side in (_sidecar(path, rank, world), path.with_suffix(".rng.json")):   # own file first, then round 4's combined file
        if side.is_file():
            st = _sidecar_state(side, path.stem, rank, world)
            if st is not None:
                DropoutRng.set_state(st)
                break
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
