"""world_size-2 gloo (CPU) tests of the data-parallel reduction logic, plus the DP correctness oracle:
the mean of per-rank gradients equals the gradient on the concatenated global batch."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lc2is_amd.dp import GradReducer
    red = GradReducer(max_chunk_elems=1000)
    flat = torch.arange(5000, dtype=torch.float32) * (rank + 1)
    red.begin_step()
    red.reduce_ranges(flat, [(0, 1500), (2048, 5000)])
    for w in red._pending:
        w.wait()
    params = torch.full((16,), float(rank))
    red.broadcast_params(params, src=0)
    q.put((rank, flat, params, red.world_size))
    dist.destroy_process_group()


def test_grad_reducer_ranges_and_broadcast_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = torch.arange(5000, dtype=torch.float32)
    for rank, flat, params, ws in res:
        assert ws == 2
        assert torch.equal(flat[:1500], base[:1500] * 3) and torch.equal(flat[2048:], base[2048:] * 3)
        assert torch.equal(flat[1500:2048], base[1500:2048] * (rank + 1))  # outside the ranges: untouched
        assert torch.equal(params, torch.zeros(16))


def test_merge_of_arena_ranges():
    from lc2is_amd.dp import GradReducer
    assert GradReducer._merge([(0, 100), (128, 300), (1024, 2000)]) == [(0, 300), (1024, 2000)]


def test_dp_mean_of_rank_grads_equals_global_batch_grad():
    """Pure-math check on the oracle: with equal per-rank batches, averaging rank gradients == one step on the
    concatenated batch (losses are means over B*H*W; no BatchNorm on the path)."""
    from oracle import ref_cpu as O
    fx = torch.load(ROOT / "tests" / "golden" / "base_tiny.pt", weights_only=True)
    cfg = O.BaseCfg(in_size=64, out_size=16, patch=16, vision=O.ClipCfg(128, 2, 2, patch=16),
                    text=O.ClipCfg(64, 1, 2, eos_token_id=511), dec_heads=2, dec_layers=1)
    inputs = {k: fx[k] for k in ("pixel_values", "input_ids", "attention_mask")}
    _, _, g_all, _ = O.train_step_sgd(fx["state_dict"], inputs, fx["labels"], cfg, 0.1)
    gs = []
    for r in range(2):
        sl = {k: v[r:r + 1] for k, v in inputs.items()}
        _, _, g, _ = O.train_step_sgd(fx["state_dict"], sl, fx["labels"][r:r + 1], cfg, 0.1)
        gs.append(g)
    for k in ("class_prototypes", "vision_decoder.layers.0.linear1.weight", "text_encoder.enc.final_layer_norm.weight"):
        mean = (gs[0][k] + gs[1][k]) / 2
        assert torch.allclose(mean, g_all[k], atol=1e-5 + 1e-4 * float(g_all[k].abs().max())), k


def test_subtract_ranges():
    from lc2is_amd.dp import GradReducer
    assert GradReducer._subtract([(0, 100)], [(10, 20), (50, 60)]) == [(0, 10), (20, 50), (60, 100)]
    assert GradReducer._subtract([(0, 100), (200, 300)], [(0, 100), (250, 400)]) == [(200, 250)]
    assert GradReducer._subtract([(0, 100)], []) == [(0, 100)]


def _layer_worker(rank, world, port, q):
    """Per-layer bucketed reduction: layers report in reverse order while 'backward' is still running; every element of
    the module's slice must be reduced exactly once."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch import nn
    from lc2is_amd.dp import GradReducer

    class Tower(nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = nn.Linear(8, 8)
            self.layers = nn.ModuleList([nn.Linear(8, 8) for _ in range(5)])
            self._grad_ready_cb = self._part_ready_cb = None

    class Arena:
        pass

    tower = Tower()
    params = list(tower.parameters())
    arena, off = Arena(), 0
    arena.ranges = {}
    for p in params:
        arena.ranges[id(p)] = (off, off + p.numel())
        off += (p.numel() + 63) // 64 * 64
    arena.grad = torch.full((off,), float(rank + 1))
    for p in params:
        lo, hi = arena.ranges[id(p)]
        p._lc2is_grad = arena.grad[lo:hi].view(p.shape)
        p.grad = p._lc2is_grad
    red = GradReducer(bucket_elems=100)          # 72 elements per layer -> a bucket closes every second layer
    red._flat, red._arena = arena.grad, arena
    red._module_ranges[id(tower)] = (tower, red._merge([arena.ranges[id(p)] for p in params]), params)
    red.begin_step()
    calls = []
    orig = red.reduce_ranges
    red.reduce_ranges = lambda flat, ranges: (calls.append(list(ranges)), orig(flat, ranges))[1]
    for layer in reversed(tower.layers):
        red._on_part_done(tower, layer)
    red._on_module_done(tower)
    red.finish_step()
    q.put((rank, arena.grad.clone(), calls, {k: v for k, v in arena.ranges.items()}.values().__len__()))
    dist.destroy_process_group()


def test_per_layer_bucketed_reduction_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_layer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grad, calls, nparams in res:
        assert nparams == 12
        assert len(calls) >= 3                                    # two early buckets (+ the remainder at module end)
        # every parameter element was summed over both ranks exactly once: 1 + 2 = 3
        covered = torch.zeros_like(grad, dtype=torch.bool)
        for ranges in calls:
            for lo, hi in ranges:
                assert not covered[lo:hi].any(), "a slice was reduced twice"
                covered[lo:hi] = True
        assert torch.equal(grad[covered], torch.full_like(grad[covered], 3.0))
        assert covered.sum() >= 6 * 72
