"""Shared machinery of the drop-in modules: bf16 weight shadows, gradient buffers, the flat arena.

Design (MI355X-first):
  * parameters stay fp32 ``nn.Parameter``s under the reference's names (``state_dict`` interchange);
  * every GEMM weight has two bf16 *shadows* in HBM — row-major [N,K] for the forward product and the
    transposed [K,N] for the dgrad product (so both are NT MFMA GEMMs) — refreshed for the whole module by
    ONE kernel launch whenever a parameter's version counter moved (optimizer step, ``load_state_dict``);
  * gradients are written by the HIP backward kernels straight into per-parameter buffers (views of one
    flat arena when ``ParamArena`` is used), never returned through autograd;
  * the module-level ``torch.autograd.Function``s only tie the modules together.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


import os as _os

# Experiment (LC2IS_WGRAD_STREAM=1): weight gradients are consumed only by the optimizer, so they can run on a side HIP
# stream beside the dgrad chain (joined in HipModule._grads_ready).  Measured +0.3 % images/s on one box: the big GEMMs
# saturate the chip either way, and sharing CUs stretches every co-running kernel — off by default.
_WGRAD_SIDE = _os.environ.get("LC2IS_WGRAD_STREAM", "0") == "1"
_wgrad_streams: dict = {}


def wgrad_stream(device) -> torch.cuda.Stream:
    s = _wgrad_streams.get(device)
    if s is None:
        s = _wgrad_streams[device] = torch.cuda.Stream(device)
    return s


class _on_wgrad_stream:
    """Context: run the enclosed launches on the wgrad stream after everything queued so far on the current one; the
    operand tensors are kept from being recycled by the caching allocator until the side stream has read them."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if not _WGRAD_SIDE or torch.cuda.is_current_stream_capturing():
            self.ctx = None
            return self
        dev = self.tensors[0].device
        side = wgrad_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        for t in self.tensors:
            t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def join_wgrad_stream(device) -> None:
    if _WGRAD_SIDE and device in _wgrad_streams:
        torch.cuda.current_stream(device).wait_stream(_wgrad_streams[device])


class DropoutRng:
    """Seeds for the counter-based dropout kernels (csrc/common.h).  Every dropout SITE of every forward draws one 64-bit
    seed; the site's backward re-uses it, so no mask is ever stored.  The stream is a splitmix64 sequence started from
    ``torch.initial_seed()`` and the process' rank: ``torch.manual_seed`` makes training runs repeatable, and a LATER
    ``torch.manual_seed`` restarts the stream (the torch seed is compared at every draw).  ``manual_seed`` sets the stream
    directly; ``get_state`` / ``set_state`` travel with a checkpoint (checkpoint.save_checkpoint / load_checkpoint)."""

    _state = None
    _torch_seed = None  # the torch.initial_seed() the stream was derived from (or that was current at manual_seed / set_state)
    last = {}           # site name -> (seed, p) of the most recent forward, for tests that re-create the masks.  Two blocks of
                        # one model share site names ("layers.0.d1"): give each a distinct `rng_name` attribute (a string
                        # prefix, default "") to keep their entries apart

    @classmethod
    def manual_seed(cls, seed: int) -> None:
        cls._state = seed & 0xFFFFFFFFFFFFFFFF
        cls._torch_seed = torch.initial_seed()

    @classmethod
    def get_state(cls) -> int | None:
        return cls._state

    @classmethod
    def set_state(cls, state: int | None) -> None:
        cls._state = None if state is None else int(state) & 0xFFFFFFFFFFFFFFFF
        cls._torch_seed = torch.initial_seed()

    @classmethod
    def next_seed(cls, site: str | None = None, p: float = 0.0) -> int:
        ts = torch.initial_seed()
        if cls._state is None or ts != cls._torch_seed:
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
            cls._state = (ts * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
            cls._torch_seed = ts
        cls._state = (cls._state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = cls._state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        if site is not None:
            cls.last[site] = (z, p)
        return z


def assign_rng_names(root) -> None:
    """Give every submodule of `root` its dotted path as the prefix of its dropout-site names (``DropoutRng.last`` keys,
    the active-site count of ``TrainStep.capture``): two decoder blocks of one composition both log ``layers.0.d1`` otherwise
    and overwrite each other's entries.  Compositions call this at the end of ``__init__``; an outer composition overwrites
    the names an inner one assigned (full paths win).  The seeds themselves do not depend on the names."""
    for name, mod in root.named_modules():
        mod.rng_name = (name + ".") if name else ""


class DropSites:
    """The dropout sites of ONE layer in ONE training forward: rate + the seed drawn for each site (kept with the saved
    activations, so the backward regenerates the same decisions).  ``None`` stands for "dropout inactive" (eval / p == 0)."""

    def __init__(self, p: float, prefix: str = "") -> None:
        self.p, self.prefix, self.seeds = float(p), prefix, {}

    def seed(self, site: str) -> int:
        s = self.seeds.get(site)
        if s is None:
            s = self.seeds[site] = DropoutRng.next_seed(self.prefix + site, self.p)
        return s

    @staticmethod
    def make(module_training: bool, p: float, prefix: str = ""):
        return DropSites(p, prefix) if (module_training and p > 0.0) else None


def drop_branch_add(ds: DropSites | None, site: str, branch32: torch.Tensor, resid32: torch.Tensor) -> torch.Tensor:
    """resid + dropout(branch): the `x + self.dropoutN(sublayer(x))` of torch's Transformer layers (fp32 rows)."""
    return ops.dropout_rows_f32(branch32, ds.p, ds.seed(site), resid=resid32)[0]


def drop_branch_grad16(ds: DropSites | None, site: str, g32: torch.Tensor, g16: torch.Tensor | None) -> torch.Tensor:
    """bf16 gradient wrt a dropped branch output from the fp32 gradient wrt `resid + dropout(branch)`."""
    if ds is None:
        return g16 if g16 is not None else ops.cast_bf16(g32)
    return ops.dropout_rows_f32(g32, ds.p, ds.seeds[site], out_f32=None, out_bf16=True)[1]


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"lc2is_amd: {what} is on {t.device}; the MI355X path has no CPU fallback — move the module and its "
            "inputs to a HIP device (the CPU oracle lives in oracle/ and is test-only)")


class HipModule(nn.Module):
    """Base of the top-level drop-in modules (one shadow table per instance)."""

    def __init__(self) -> None:
        super().__init__()
        self._sh = None           # namespace of shadow tensors
        self._sh_table = None
        self._sh_versions = None
        self._sh_device = None
        self._sh_ptrs = []
        self._grad_ready_cb = None  # set by dp.GradReducer: called when this module's backward is complete
        self._part_ready_cb = None  # set by dp.GradReducer: called when a sub-module's (a layer's) gradients are complete

    # subclasses: return (namespace_dict, entries) where entries = [(param, dst, dstT)]
    def _build_shadows(self, device):  # pragma: no cover - abstract
        raise NotImplementedError

    def _params_for_version(self):
        return list(self.parameters(recurse=True))

    def _ensure_ready(self):
        ps = self._params_for_version()
        dev = ps[0].device
        require_cuda(ps[0], f"{type(self).__name__} parameters")
        if self._sh is None or self._sh_device != dev or any(p.data_ptr() != q for p, q in zip(ps, self._sh_ptrs)):
            ns, entries = self._build_shadows(dev)
            self._sh = ns
            self._sh_table = ops.ShadowTable(entries, dev) if entries else None
            self._sh_device = dev
            self._sh_ptrs = [p.data_ptr() for p in ps]
            self._sh_versions = None
        vers = [p._version for p in ps]
        if vers != self._sh_versions:
            if self._sh_table is not None:
                self._sh_table.refresh()
            self._post_refresh()
            self._sh_versions = vers
        return self._sh

    def _post_refresh(self):
        pass

    def _grads_ready(self):
        if self._sh_device is not None:
            join_wgrad_stream(self._sh_device)     # this module's weight gradients are complete on the current stream
        if self._grad_ready_cb is not None:
            self._grad_ready_cb(self)

    def _part_grads_ready(self, part: nn.Module):
        """A layer of this module has finished its backward (all its gradient kernels are queued on the current stream)."""
        if self._part_ready_cb is not None:
            if self._sh_device is not None:
                join_wgrad_stream(self._sh_device)
            self._part_ready_cb(self, part)

    def invalidate_shadows(self):
        """Call after changing parameters behind torch's back (the fused optimizer kernels do)."""
        self._sh_versions = None

    def _apply(self, fn, *args, **kwargs):  # .to()/.cuda() invalidate the shadows
        self._sh = None
        return super()._apply(fn, *args, **kwargs)


def grad_buf(p: torch.Tensor):
    """(buffer, accumulate) for writing the gradient of parameter ``p`` from a HIP kernel."""
    if p.grad is None:
        g = getattr(p, "_lc2is_grad", None)
        if g is None or g.shape != p.shape or g.device != p.device:
            g = torch.empty_like(p)
        p.grad = g
        return g, False
    return p.grad, True


class WgradBatch:
    """``with WgradBatch():`` defers the weight-gradient GEMMs requested by ``linear_bwd_params`` inside the block and
    launches them together at exit (``ops.gemm_tn_grouped``: one grid + one ordered reduce for a whole transformer layer).
    The operands are referenced by the batch until then.  Problems the grouped kernel does not take (N or K not a
    multiple of 256, few tokens, mixed accumulate state) are launched right away as before."""

    _active = None

    def __init__(self, enabled: bool = True):
        self.enabled = enabled and _os.environ.get("LC2IS_WGRAD_GROUPED", "1") != "0"
        self.items = []

    def __enter__(self):
        if self.enabled:
            self._prev, WgradBatch._active = WgradBatch._active, self
            self._prev_ln = ops.ln_defer_begin()   # the LayerNorm dgamma / dbeta reductions of the block leave together too
        return self

    def __exit__(self, exc_type, *exc):
        if self.enabled:
            WgradBatch._active = self._prev
            if exc_type is None:
                self.flush()
                ops.ln_defer_end(self._prev_ln)
            else:
                ops._ln_defer = self._prev_ln
        return False

    def flush(self):
        ops.ln_defer_flush()
        items, self.items = self.items, []
        gmax = ops.GROUP_MAX
        for i in range(0, len(items), gmax):
            chunk = items[i:i + gmax]
            if len(chunk) == 1:
                dy, x, g, gb, acc = chunk[0]
                ops.gemm_tn(dy, x, g, accumulate=acc, db=gb)
            else:
                ops.gemm_tn_grouped(chunk)


def linear_bwd_params(dy_bf16: torch.Tensor, x_bf16: torch.Tensor, weight: nn.Parameter, bias: nn.Parameter | None):
    """dW = dy^T x, db = colsum(dy), written into the parameters' gradient buffers."""
    want_b = bias is not None and bias.requires_grad
    if weight.requires_grad:
        g, acc = grad_buf(weight)
        gb, accb = grad_buf(bias) if want_b else (None, False)
        batch = WgradBatch._active
        if batch is not None and (gb is None or accb == acc) and ops.gemm_tn_groupable(dy_bf16, x_bf16):
            batch.items.append((dy_bf16, x_bf16, g.reshape(g.shape[0], -1), gb, acc))
            return
        with _on_wgrad_stream(dy_bf16, x_bf16):
            if gb is not None and accb != acc:      # mixed gradient state: fall back to the separate column-sum launch
                ops.colsum(dy_bf16, gb, accumulate=accb)
                gb = None
            ops.gemm_tn(dy_bf16, x_bf16, g.reshape(g.shape[0], -1), accumulate=acc, db=gb)
    elif want_b:
        g, acc = grad_buf(bias)
        with _on_wgrad_stream(dy_bf16):
            ops.colsum(dy_bf16, g, accumulate=acc)


def vec_grad(p: nn.Parameter | None):
    if p is None or not p.requires_grad:
        return None, False
    return grad_buf(p)


class ParamArena:
    """Flat fp32 storage for all parameters (and their gradients) of a model: one fused optimizer launch,
    one contiguous all-reduce payload, 256-byte aligned views.  Build AFTER moving the model to the GPU."""

    ALIGN = 64  # elements

    def __init__(self, module: nn.Module):
        params = []
        seen = set()
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise RuntimeError("ParamArena: module has no parameters")
        dev = params[0].device
        require_cuda(params[0], "ParamArena parameters")
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, self.offsets = params, offs
        self.ranges = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._lc2is_grad = self.grad[o:o + n].view(p.shape)
                p.grad = None
                self.ranges[id(p)] = (o, o + n)
        self.numel = total

    def attach_grads(self):
        for p in self.params:
            p.grad = p._lc2is_grad

    def zero_grad(self, set_to_none: bool = True):
        """set_to_none: the next backward overwrites (no memset needed); else zero the flat buffer."""
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.grad.zero_()
            self.attach_grads()

    def finalize_grads(self):
        """After backward: parameters without a gradient — frozen ones (requires_grad=False, e.g. the reference's frozen
        text towers, model/model.py:115-117) and ones the graph never reached (CLIP's post_layernorm) — get a zero
        gradient so the flat buffer is fully defined for the all-reduce.  Returns the LIVE segments [(lo, hi)] of the
        arena (maximal runs of parameters that did receive a gradient, alignment padding included): torch.optim skips
        parameters whose grad is None, so an optimizer with weight decay must only touch these."""
        live, run = [], None
        for p, o in zip(self.params, self.offsets):
            end = o + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if p.grad is None:
                p._lc2is_grad.zero_()
                p.grad = p._lc2is_grad
                if run is not None:
                    live.append(run)
                    run = None
            else:
                run = (run[0], end) if run is not None else (o, end)
        if run is not None:
            live.append(run)
        return live

    def module_range(self, module: nn.Module):
        lo, hi = None, None
        for p in module.parameters():
            r = self.ranges.get(id(p))
            if r is None:
                continue
            lo = r[0] if lo is None else min(lo, r[0])
            hi = r[1] if hi is None else max(hi, r[1])
        return lo, hi
