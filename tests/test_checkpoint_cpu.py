"""CPU: checkpoint interchange — Engine.save-style files and local hub-format snapshots load into the
drop-in modules with the reference's / transformers' key names (host logic only, no kernels)."""
from pathlib import Path

import torch

import lc2is_amd.nn as N
from lc2is_amd import checkpoint as C
from lc2is_amd.nn.clip import ClipArch

ROOT = Path(__file__).resolve().parent.parent


def _tiny_arches():
    v = ClipArch(64, 1, 1, 128)
    t = ClipArch(64, 1, 1, 128, vocab=64, max_pos=16, eos_token_id=63)
    return v, t


def test_engine_style_checkpoint_roundtrip(tmp_path):
    v, t = _tiny_arches()
    protos = torch.randn(5, 64)
    m = N.BaseModelWithText(16, 64, 16, vision_arch=v, text_arch=t, nhead=1, dim_feedforward=128, out_dim=64, prototypes=protos)
    f = C.save_checkpoint(m, tmp_path, 7)
    assert f == tmp_path / "checkpoints" / "step-7.pt"                          # engine.py:189 naming
    m2 = N.BaseModelWithText(16, 64, 16, vision_arch=v, text_arch=t, nhead=1, dim_feedforward=128, out_dim=64, prototypes=torch.zeros(5, 64))
    res = C.load_checkpoint(m2, f)
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_local_hub_snapshot_into_towers(tmp_path):
    from safetensors.torch import save_file
    v, t = _tiny_arches()
    vis = N.ImageEncoderCLIP(in_size=64, patch_size=16, arch=v)
    txt = N.TextEncoderCLIP(patch_size=16, arch=t)
    hub = {}
    for k, p in vis.state_dict().items():
        hub["vision_model." + k[len("enc."):]] = p.clone() + 1
    for k, p in txt.state_dict().items():
        hub["text_model." + k[len("enc."):]] = p.clone() + 2
    hub["logit_scale"] = torch.tensor(1.0)
    hub["visual_projection.weight"] = torch.zeros(4, 64)
    d = tmp_path / "clip"
    d.mkdir()
    save_file(hub, str(d / "model.safetensors"))
    ref_v = {k: p.clone() for k, p in vis.state_dict().items()}
    ref_t = {k: p.clone() for k, p in txt.state_dict().items()}
    C.load_pretrained_dir(vis, d)
    C.load_pretrained_dir(txt, d)
    assert all(torch.equal(vis.state_dict()[k], ref_v[k] + 1) for k in ref_v)
    assert all(torch.equal(txt.state_dict()[k], ref_t[k] + 2) for k in ref_t)
    # Swin: classification-checkpoint layout ("swin." prefix + classifier head) in pytorch_model.bin
    sw = N.SwinTransformer(N.SwinArch(32, (2, 2, 2, 2), (1, 2, 4, 8), 5), drop_path_rate=0.0)
    ref_s = {k: p.clone() for k, p in sw.state_dict().items()}
    hub_s = {"swin." + k[len("encoder."):]: p.clone() - 1 for k, p in ref_s.items()}
    hub_s["classifier.weight"] = torch.zeros(3, 256)
    d2 = tmp_path / "swin"
    d2.mkdir()
    torch.save(hub_s, d2 / "pytorch_model.bin")
    C.load_pretrained_dir(sw, d2)
    assert all(torch.equal(sw.state_dict()[k], ref_s[k] - 1) for k in ref_s)


def test_dropout_rng_follows_torch_seed_and_travels_with_the_checkpoint(tmp_path):
    """a later torch.manual_seed restarts the dropout stream; the stream state is saved beside step-N.pt
    (JSON sidecar: the reference's file stays a plain state dict) and restored on load."""
    from lc2is_amd.nn.base import DropoutRng
    torch.manual_seed(5)
    a = [DropoutRng.next_seed() for _ in range(3)]
    torch.manual_seed(6)
    b = [DropoutRng.next_seed() for _ in range(3)]
    torch.manual_seed(5)
    c = [DropoutRng.next_seed() for _ in range(3)]
    assert a == c and a != b and len(set(a)) == 3
    v, t = _tiny_arches()
    m = N.BaseModelWithText(16, 64, 16, vision_arch=v, text_arch=t, nhead=1, dim_feedforward=128, out_dim=64, prototypes=torch.zeros(5, 64))
    f = C.save_checkpoint(m, tmp_path, 3)
    nxt = [DropoutRng.next_seed() for _ in range(2)]
    DropoutRng.manual_seed(999)
    C.load_checkpoint(m, f)
    assert [DropoutRng.next_seed() for _ in range(2)] == nxt
    assert isinstance(torch.load(f, weights_only=True), dict)                   # still the reference's format


def test_packed_cross_attention_key_of_equal_width_layers_roundtrips():
    """d_kv == d_model: torch's MultiheadAttention (hence a reference checkpoint) holds ONE ``multihead_attn.in_proj_weight``;
    the drop-in keeps q / k / v apart and splits / merges the key on load / save (DenseClip's prompt layers)."""
    layer = N.PromptLayer(d_model=128, d_kv=128, nhead=2, dim_feedforward=64, batch_first=True)
    dec = N.PromptDecoder(layer, num_layers=2)
    sd = dec.state_dict()
    assert "layers.0.multihead_attn.in_proj_weight" in sd and "layers.0.multihead_attn.q_proj_weight" not in sd
    assert tuple(sd["layers.0.multihead_attn.in_proj_weight"].shape) == (384, 128)
    ref = {k: torch.randn_like(v) for k, v in sd.items()}
    res = dec.load_state_dict(ref, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ca = dec.layers[1].multihead_attn
    assert torch.equal(torch.cat([ca.q_proj_weight, ca.k_proj_weight, ca.v_proj_weight], 0), ref["layers.1.multihead_attn.in_proj_weight"])
    # unequal widths keep the three separate keys
    sd2 = N.PromptDecoder(N.PromptLayer(d_model=128, d_kv=256, nhead=2, dim_feedforward=64, batch_first=True), 1).state_dict()
    assert "layers.0.multihead_attn.k_proj_weight" in sd2 and "layers.0.multihead_attn.in_proj_weight" not in sd2


def _rng_worker(rank, world, port, tmp, q):
    import os
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lc2is_amd import checkpoint as Cc
    from lc2is_amd.nn.base import DropoutRng
    torch.manual_seed(11)                                   # the same torch seed everywhere: the rank makes the streams differ
    first = [DropoutRng.next_seed() for _ in range(2)]
    m = torch.nn.Linear(4, 4)
    f = Cc.save_checkpoint(m, tmp, 7)                       # every rank calls; rank 0 writes step-7.pt, each rank its own sidecar
    if rank == 0:                                           # the usual DP pattern must not hang: a plain write, no collective
        Cc.save_checkpoint(m, tmp + "/rank0_only", 8)
    dist.barrier()                                          # (test-side: rank 1 loads what rank 0 wrote)
    nxt = [DropoutRng.next_seed() for _ in range(3)]       # what the uninterrupted run draws next on THIS rank
    DropoutRng.manual_seed(12345)                           # a resumed process starts somewhere else
    Cc.load_checkpoint(m, f)
    resumed = [DropoutRng.next_seed() for _ in range(3)]
    q.put((rank, first, nxt, resumed))
    dist.destroy_process_group()


def test_dropout_rng_sidecar_is_per_rank_under_data_parallelism(tmp_path):
    """under DP each rank's dropout stream is derived from seed + rank, so after a resume rank r must
    continue ITS stream — not rank 0's.  ADVICE round 4 (medium): save_checkpoint is a plain write like Engine.save (no
    collective): every rank writes its own sidecar, and `if rank == 0: save_checkpoint(...)` returns."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 777) % 2000
    procs = [ctx.Process(target=_rng_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    (r0, first0, nxt0, res0), (r1, first1, nxt1, res1) = res
    assert first0 != first1 and nxt0 != nxt1                # the ranks draw different masks
    assert res0 == nxt0 and res1 == nxt1                    # and each resumes its own stream
    import json
    for r in range(2):
        meta = json.loads((tmp_path / "checkpoints" / f"step-7.rng.rank{r}.json").read_text())
        assert meta["train_step"] == 7 and meta["world_size"] == 2 and set(meta["dropout_rng_state"]) == {str(r)}
    only = tmp_path / "rank0_only" / "checkpoints"
    assert (only / "step-8.pt").is_file() and (only / "step-8.rng.rank0.json").is_file()
    assert not (only / "step-8.rng.rank1.json").exists()


def test_older_sidecar_formats(tmp_path):
    """Round 4's combined file (one entry per rank) and the first format (a bare state, single process only) are still read;
    malformed files are ignored instead of raising."""
    import json
    from lc2is_amd.nn.base import DropoutRng
    m = torch.nn.Linear(3, 3)
    torch.manual_seed(3)
    DropoutRng.next_seed()
    f = C.save_checkpoint(m, tmp_path, 5)
    side = f.with_suffix(".rng.json")
    side.write_text(json.dumps({"train_step": 5, "world_size": 1, "dropout_rng_state": {"0": 1234}}))   # round 4
    C.load_checkpoint(m, f)
    assert DropoutRng.get_state() == 1234
    side.write_text(json.dumps({"dropout_rng_state": 777}))                                              # first format
    C.load_checkpoint(m, f)
    assert DropoutRng.get_state() == 777
    for bad in ('{"train_step": 5, "world_size": "x", "dropout_rng_state": {"0": 1}}', '{"dropout_rng_state": {"0": 1}}',
                'not json', '{"train_step": 5, "world_size": 1, "dropout_rng_state": {"0": "y"}}'):
        side.write_text(bad)
        DropoutRng.manual_seed(42)
        C.load_checkpoint(m, f)
        assert DropoutRng.get_state() == 42


def test_stale_dropout_sidecar_is_ignored(tmp_path):
    """A sidecar that names another train step (left by an earlier run beside a reference step-N.pt) must not be adopted."""
    import json
    from lc2is_amd.nn.base import DropoutRng
    m = torch.nn.Linear(3, 3)
    torch.manual_seed(3)
    DropoutRng.next_seed()
    f = C.save_checkpoint(m, tmp_path, 5)
    side = f.with_suffix(".rng.json")
    meta = json.loads(side.read_text())
    meta["train_step"] = 4
    side.write_text(json.dumps(meta))
    DropoutRng.manual_seed(42)
    before = DropoutRng.get_state()
    C.load_checkpoint(m, f)
    assert DropoutRng.get_state() == before
