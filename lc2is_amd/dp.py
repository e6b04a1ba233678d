"""Data-parallel gradient reduction over RCCL/xGMI (NEW functionality — the reference is single-device,
SURVEY.md §2 row 19 / §8e).

One process per GPU; every rank holds a full replica and its own B images.  Gradients live in ONE flat fp32
arena (``ParamArena.grad``); ``GradReducer`` all-reduces (sum) contiguous slices of it as soon as the module
that owns a slice has finished its backward — head -> decoder -> vision/text — so the collective for the
late layers' gradients runs on RCCL's stream while the earlier layers' backward kernels still execute
(overlap without a tracing compiler: the backward order is ours to know).  The 1/world_size average is folded
into the fused optimizer kernel's ``grad_scale``.

xGMI note: 157 M fp32 gradients = 628 MB; slices are whole modules (vision 343 MB, text 253 MB, decoder 30 MB,
head 3 MB), i.e. few large collectives — per-link-bound rings want large messages, not 25 MB DDP buckets.

The class only needs ``torch.distributed`` and flat tensors, so the same code runs under ``gloo`` on CPU
tensors (tests/test_dp_gloo.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, process_group=None, max_chunk_elems: int = 64 * 1024 * 1024,
                 bucket_elems: int = 12 * 1024 * 1024) -> None:
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.max_chunk = max_chunk_elems
        self._pending = []
        self._flat = None
        self._done = set()
        self._module_ranges = {}
        self.bucket_elems = bucket_elems     # layers are reduced in buckets of >= this many elements (48 MB fp32): xGMI
        self._bucket = {}                    # rings are per-link bound and want large messages, not one per layer
        self._early = {}
        # CUs left to RCCL's channel kernels while collectives are in flight (LC2IS_DP_CU_RESERVE, default 0 = plan on all 256): every
        # large-tile GEMM block takes a whole CU, so a CU held by a collective turns "one round of tiles" into two.  Set from the
        # first all_reduce of a step to finish_step (lc2is_set_cu_budget; NT GEMM plans are bitwise equal under any budget, the
        # weight-gradient plans split M differently, i.e. sum their fp32 partials in another order: last-bit differences).
        import os
        self.cu_reserve = max(0, min(128, int(os.environ.get("LC2IS_DP_CU_RESERVE", "0"))))
        self._budget_on = False
        self.collectives_last_step = 0       # all_reduce calls issued by the most recent finished step

    # -- wiring ------------------------------------------------------------------------------------------------
    def attach(self, model, arena) -> None:
        """Register per-module slices of the arena and the modules' backward-complete callbacks."""
        from .nn.base import HipModule
        self._flat = arena.grad
        self._arena = arena
        tops = [m for m in model.children() if isinstance(m, HipModule)]
        owned = set()
        for m in tops:
            ps = list(m.parameters())
            owned.update(id(p) for p in ps)
            self._module_ranges[id(m)] = (m, self._merge([arena.ranges[id(p)] for p in ps]), ps)
            m._grad_ready_cb = self._on_module_done
            m._part_ready_cb = self._on_part_done
        rest = [p for p in model.parameters() if id(p) not in owned]  # e.g. class_prototypes of the composition
        self._module_ranges[id(model)] = (model, self._merge([arena.ranges[id(p)] for p in rest]), rest)
        model._grad_ready_cb = self._on_module_done

    @staticmethod
    def _merge(ranges):
        out = []
        for lo, hi in sorted(ranges):
            lo_al = lo
            if out and lo_al - out[-1][1] < 64:  # arena padding between neighbours
                out[-1][1] = hi
            else:
                out.append([lo_al, hi])
        return [(a, b) for a, b in out]

    # -- per step ------------------------------------------------------------------------------------------------
    def begin_step(self) -> None:
        self._pending.clear()
        self._done.clear()
        self._bucket.clear()
        self._early.clear()

    def _zero_missing(self, params) -> None:
        for p in params:  # parameters the graph never reached still need defined (zero) gradients
            if p.grad is None:
                p._lc2is_grad.zero_()
                p.grad = p._lc2is_grad

    def _on_part_done(self, module, part) -> None:
        """A layer of `module` finished its backward while the rest of the module is still running: its slice of the
        arena joins the module's bucket, and a full bucket is all-reduced now, under the remaining backward."""
        key = id(module)
        if key not in self._module_ranges or key in self._done:
            return
        ps = [p for p in part.parameters() if id(p) in self._arena.ranges]
        self._zero_missing(ps)
        b = self._bucket.setdefault(key, [])
        b.extend(self._arena.ranges[id(p)] for p in ps)
        if sum(hi - lo for lo, hi in b) >= self.bucket_elems:
            self._flush_bucket(key)

    def _flush_bucket(self, key) -> None:
        b = self._bucket.pop(key, None)
        if b:
            merged = self._merge(b)
            self.reduce_ranges(self._flat, merged)
            self._early.setdefault(key, []).extend(merged)

    @staticmethod
    def _subtract(ranges, done):
        """ranges minus the already reduced sub-ranges (both lists of half-open intervals)."""
        out = []
        for lo, hi in ranges:
            cur = lo
            for a, b in sorted(done):
                if b <= cur or a >= hi:
                    continue
                if a > cur:
                    out.append((cur, a))
                cur = max(cur, b)
            if cur < hi:
                out.append((cur, hi))
        return out

    def _on_module_done(self, module) -> None:
        key = id(module)
        if key in self._done or key not in self._module_ranges:
            return
        self._flush_bucket(key)
        self._done.add(key)
        _, ranges, params = self._module_ranges[key]
        self._zero_missing(params)
        self.reduce_ranges(self._flat, self._subtract(ranges, self._early.get(key, [])))

    def _set_budget(self, on: bool) -> None:
        if self.cu_reserve and on != self._budget_on and self._flat is not None and self._flat.is_cuda:
            from . import ops
            ops.set_cu_budget(256 - self.cu_reserve if on else 0)
            self._budget_on = on

    def reduce_ranges(self, flat: torch.Tensor, ranges) -> None:
        if ranges:
            self._set_budget(True)
        for lo, hi in ranges:
            for a in range(lo, hi, self.max_chunk):
                b = min(a + self.max_chunk, hi)
                self._pending.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_step(self) -> None:
        """Reduce whatever has not been reduced yet, then make the compute stream wait for every collective."""
        for key, (_, ranges, _) in self._module_ranges.items():
            if key not in self._done:
                self._flush_bucket(key)
                self._done.add(key)
                self.reduce_ranges(self._flat, self._subtract(ranges, self._early.get(key, [])))
        try:
            for w in self._pending:
                w.wait()
        finally:                             # (an exception must not leave the planners on a reduced CU budget)
            self.collectives_last_step = len(self._pending)
            self._pending.clear()
            self._set_budget(False)

    def broadcast_params(self, flat_params: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(flat_params, src=src, group=self.group)
