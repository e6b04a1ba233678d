"""Score-map tail of the FTN / hierarchical compositions on MI355X.

Reference (model/final.py:350-356; the same four lines at model/model.py:204-212 and model/ftn.py:56-62):
    visual = rearrange(visual_embeddings, "b (h w) c -> b c h w"); visual = F.normalize(visual, dim=1)
    text   = F.normalize(text_embeddings, dim=-1)
    score_map = einsum('bchw,bkc->bkhw', visual, text);  score_map = F.interpolate(score_map, "bilinear", x4)
followed by nn.CrossEntropyLoss()(score_map, labels) in the engine (engine.py:94).

HIP path: L2-normalise both sides (one wave per row), one strided-batched MFMA GEMM launch for the per-image K class scores at the low
resolution, then the fused bilinear-x4 + softmax-CE kernel: the [B,K,4h,4w] fp32 map (157 MB/img at 512x512,
K=150 — "the" bandwidth-bound tensor of config 5) is only materialised when the caller asks for it.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .base import require_cuda

KPAD = 192


def _scores_lo(visual, text):
    B, P, C = visual.shape
    K = text.shape[1]
    if K > KPAD or C % 64:
        raise NotImplementedError("lc2is_amd ScoreMapTail: at most 192 classes, channel count a multiple of 64")
    v32 = visual.reshape(B * P, C).float().contiguous()
    t32 = text.reshape(B * K, C).float().contiguous()
    _, vn16, vinv = ops.l2norm_fwd(v32, want_f32=False)
    _, tn16, tinv = ops.l2norm_fwd(t32, want_f32=False)
    tpad = torch.zeros(B, KPAD, C, dtype=torch.bfloat16, device=visual.device)
    tpad[:, :K] = tn16.view(B, K, C)
    # per-image class embeddings: ONE strided-batched NT launch (blockIdx.y = image)
    _, scores = ops.gemm_nt_batched(vn16.view(B, P, C), tpad)
    scores = scores.view(B * P, KPAD)
    return scores, dict(v32=v32, t32=t32, vn16=vn16, tpad=tpad, vinv=vinv, tinv=tinv, dims=(B, P, C, K))


def _scores_bwd(ds32, sv):
    """ds32: gradient wrt the low-resolution scores [B*P, KPAD] (fp32)."""
    B, P, C, K = sv["dims"]
    ds16 = ops.cast_bf16(ds32)
    dtn = torch.empty(B, KPAD, C, dtype=torch.float32, device=ds32.device)
    tT = ops.transpose_bf16_batched(sv["tpad"])                                    # [B, C, KPAD]
    _, dvn = ops.gemm_nt_batched(ds16.view(B, P, KPAD), tT)                         # d(normalised visual) [B, P, C]
    dvn = dvn.view(B * P, C)
    vn16 = sv["vn16"]
    probs = [(ds16[b * P:(b + 1) * P], vn16[b * P:(b + 1) * P], dtn[b], None, False) for b in range(B)]
    gmax = ops.GROUP_MAX
    for i in range(0, B, gmax):                                                    # the B weight-gradient-shaped products: one grid
        ops.gemm_tn_grouped(probs[i:i + gmax])
    dv = ops.l2norm_bwd(dvn, sv["v32"], sv["vinv"])
    dt = ops.l2norm_bwd(dtn[:, :K].reshape(B * K, C).contiguous(), sv["t32"], sv["tinv"])
    return dv.view(B, P, C), dt.view(B, K, C)


class _ScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, visual, text, labels, scale, ignore_index, save):
        B, P, C = visual.shape
        K = text.shape[1]
        h = int(round(P ** 0.5))
        scores, sv = _scores_lo(visual, text)
        if labels is None:
            _, _, hi = ops.head_upsample_ce(scores, None, B, h, h, K, scale, ops.INTERP_BILINEAR, want_scores=True,
                                            want_loss=False)
            ctx.sv, ctx.fused, ctx.meta = (sv if save else None), None, (B, h, K, scale)
            return hi
        n = float(B * h * scale * h * scale)
        loss2, dlo, _ = ops.head_upsample_ce(scores, labels.contiguous(), B, h, h, K, scale, ops.INTERP_BILINEAR,
                                             want_grad=save, ignore_index=ignore_index, grad_scale=1.0 / n)
        if save:   # mean over the pixels the kernel counted (negative / ignore_index / out-of-range labels are skipped)
            dlo.mul_(n / loss2[1].clamp_min(1.0))
        ctx.sv, ctx.fused, ctx.meta = (sv if save else None), dlo, (B, h, K, scale)
        return loss2[0] / loss2[1]

    @staticmethod
    def backward(ctx, gout):
        B, h, K, scale = ctx.meta
        if ctx.fused is not None:
            ds = ctx.fused * gout
        else:
            ds = ops.upsample_bwd_nchw(gout.float().contiguous(), B, h, h, K, scale, ops.INTERP_BILINEAR, KPAD)
        dv, dt = _scores_bwd(ds, ctx.sv)
        ctx.sv = ctx.fused = None
        return dv, dt, None, None, None, None


class ScoreMapTail(nn.Module):
    """``forward(visual_embeddings [B,h*w,C], text_embeddings [B,K,C]) -> score_map [B,K,S*h,S*w]`` (the reference's
    return value) and ``loss(visual, text, labels)`` = CrossEntropyLoss()(score_map, labels) without ever writing the map."""

    def __init__(self, scale_factor: int = 4) -> None:
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, visual_embeddings: torch.Tensor, text_embeddings: torch.Tensor) -> torch.Tensor:
        require_cuda(visual_embeddings, "visual_embeddings")
        save = torch.is_grad_enabled() and (visual_embeddings.requires_grad or text_embeddings.requires_grad)
        return _ScoreFn.apply(visual_embeddings, text_embeddings, None, self.scale_factor, -100, save)

    def loss(self, visual_embeddings: torch.Tensor, text_embeddings: torch.Tensor, labels: torch.Tensor,
             ignore_index: int = -100) -> torch.Tensor:
        require_cuda(visual_embeddings, "visual_embeddings")
        save = torch.is_grad_enabled() and (visual_embeddings.requires_grad or text_embeddings.requires_grad)
        return _ScoreFn.apply(visual_embeddings, text_embeddings, labels, self.scale_factor, ignore_index, save)
