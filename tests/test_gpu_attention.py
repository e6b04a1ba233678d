"""GPU parity of the fused attention kernels against explicit fp64 softmax(QK^T)V."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_attention(q, k, v, B, H, Sq, Sk, D, scale, causal, kbias):
    q4 = q.double().reshape(B, Sq, H, D).transpose(1, 2)
    k4 = k.double().reshape(B, Sk, H, D).transpose(1, 2)
    v4 = v.double().reshape(B, Sk, H, D).transpose(1, 2)
    s = q4 @ k4.transpose(-1, -2) * scale
    if kbias is not None:
        s = s + kbias.double()[:, None, None, :]
    if causal:
        s = s + torch.full((Sq, Sk), float("-inf"), dtype=torch.float64, device=q.device).triu(1)
    p = torch.softmax(s, dim=-1)
    o = (p @ v4).transpose(1, 2).reshape(B * Sq, H * D)
    lse = torch.logsumexp(s, dim=-1)
    return o, lse, p


CASES = [
    # B, H, Sq, Sk, D, causal, mask, packed
    (2, 3, 1025, 1025, 64, False, False, True),   # ViT-B/16 @512 token count, packed QKV
    (2, 2, 65, 65, 64, False, False, True),       # config-1 token count
    (3, 2, 16, 16, 64, True, True, True),         # CLIP text: causal ∧ padding
    (2, 2, 77, 77, 64, True, True, False),
    (2, 8, 1024, 1024, 96, False, False, True),   # decoder self-attention
    (2, 8, 1024, 16, 96, False, True, False),     # decoder cross-attention with key padding
    (1, 2, 300, 200, 128, False, False, False),
    (1, 1, 129, 64, 64, False, False, False),
    # round 2: the unrolled interior loops of the forward / dQ kernels and their hand-off to the general body
    (1, 2, 400, 400, 64, True, False, True),      # causal over several query blocks: interior tiles left of the diagonal (2*bx of them)
    (1, 1, 330, 330, 128, True, True, False),     # the same on the two-stage ring (D = 128), with key padding (no interior tiles)
    (1, 2, 256, 512, 64, False, False, False),    # Sk a multiple of 64: every tile is interior, the last one leaves through the general body
    (1, 1, 200, 448, 64, False, False, False),    # 7 tiles: two unrolled triples would overrun, one triple + 4 general tiles
    (1, 1, 100, 832, 96, False, False, False),    # 13 tiles on the two-stage ring
    # round 3: config 5 at its real size — spatial-reduction attention of the 64x64 stage (queries 4096, keys 4096 / 4 after the
    # 2x2 stride-2 conv, model/hierarchical.py:214-219), 8 heads x 64
    (1, 8, 4096, 1024, 64, False, False, False),
    # ragged last tiles of at most 32 rows (S = 64 n + 1: the empty second half of the tile is skipped in all three kernels), with
    # different query / key counts and many heads; and the ViT token count with a key-padding mask
    (2, 5, 257, 385, 64, False, False, False),
    (1, 2, 1025, 1025, 64, False, True, True),
]


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,mask,packed", CASES)
def test_attention_fwd(dev, B, H, Sq, Sk, D, causal, mask, packed):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Sq + D)
    if packed:
        qkv = (torch.randn(B * Sq, 3 * H * D, generator=g)).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    else:
        q = torch.randn(B * Sq, H * D, generator=g).to(torch.bfloat16).to(dev)
        k = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
        v = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
    kbias = None
    if mask:
        valid = torch.randint(1, Sk + 1, (B,), generator=g)
        kbias = torch.zeros(B, Sk)
        for i in range(B):
            kbias[i, valid[i]:] = float("-inf")
        kbias = kbias.to(dev)
    scale = 1.0 / math.sqrt(D)
    o, lse2 = ops.attention_fwd(q, k, v, B, H, Sq, Sk, D, scale, causal=causal, kbias=kbias)
    ro, rlse, _ = _ref_attention(q, k, v, B, H, Sq, Sk, D, scale, causal, kbias)
    err = (o.double() - ro).abs().max().item()
    assert err < 2e-2, err  # P is rounded to bf16 (2^-9) before PV; |V| ~ N(0,1)
    rel = ((o.double() - ro).norm() / ro.norm()).item()
    assert rel < 6e-3, rel
    assert (lse2.double() * math.log(2.0) - rlse).abs().max().item() < 2e-3


def test_attention_fwd_spiked_scores(dev):
    """One key row spikes against one query at a late tile: forces the running-max rescale branch."""
    from lc2is_amd import ops
    B, H, S, D = 1, 1, 512, 64
    g = torch.Generator(device="cpu").manual_seed(11)
    q = torch.randn(S, D, generator=g)
    k = torch.randn(S, D, generator=g)
    v = torch.randn(S, D, generator=g)
    k[400] = q[37] * 6.0
    q, k, v = (t.to(torch.bfloat16).to(dev) for t in (q, k, v))
    o, _ = ops.attention_fwd(q, k, v, B, H, S, S, D, 0.125)
    ro, _, _ = _ref_attention(q, k, v, B, H, S, S, D, 0.125, False, None)
    assert (o.double() - ro).abs().max().item() < 3e-2


@pytest.mark.parametrize("bits", [5.0, 7.0, 9.5, 10.5, 14.0, 60.0, 140.0, 400.0])
@pytest.mark.parametrize("key", [40, 70, 96 + 64 * 5, 1000])
def test_attention_fwd_frame_growth_by_chosen_bits(dev, bits, key):
    """Round 5: interior half steps check the lane's SUM of probabilities (<= 2^10) instead of a per-score maximum; a half whose
    sum fails is recomputed on the classic path, which moves the row's frame.  One key is made to score `bits` (log2 units) above
    everything a chosen query has seen — below the old threshold (6), between the old and the new one (probabilities up to 2^10
    stay in the frame), just past the new one, and far beyond fp32's exp2 range (inf in the light path) — at a key position in
    the first tile's second half, in the second tile, in a late tile of the unrolled loop and in the rolled tail.  Output and
    log-sum-exp against fp64 on the full tensor; the backward runs on the result (its LSE comes from here)."""
    from lc2is_amd import ops
    B, H, S, D = 2, 2, 1025, 64
    g = torch.Generator(device="cpu").manual_seed(int(bits * 10) + key)
    q = torch.randn(B * S, H * D, generator=g)
    k = torch.randn(B * S, H * D, generator=g)
    v = torch.randn(B * S, H * D, generator=g)
    scale = 0.125
    for b in range(B):
        for h in range(H):
            qi = 100 + 37 * h + b          # the query that meets the spike
            qv = q[b * S + qi, h * D:(h + 1) * D]
            base = (q[b * S + qi, h * D:(h + 1) * D] @ k[b * S:(b + 1) * S, h * D:(h + 1) * D].T * scale * 1.4426950408889634).max().item()
            a = (base + bits) / (qv @ qv * scale * 1.4426950408889634).item()
            k[b * S + key, h * D:(h + 1) * D] = qv * a
    q, k, v = (t.to(torch.bfloat16).to(dev) for t in (q, k, v))
    o, lse2 = ops.attention_fwd(q, k, v, B, H, S, S, D, scale)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ro, rlse, _ = _ref_attention(qd, kd, vd, B, H, S, S, D, scale, False, None)
    assert (o.double() - ro.detach()).abs().max().item() < 3e-2
    assert ((o.double() - ro.detach()).norm() / ro.detach().norm()).item() < 6e-3
    assert (lse2.double() * math.log(2.0) - rlse.detach()).abs().max().item() < 2e-2
    do = torch.randn(B * S, H * D, generator=g).to(torch.bfloat16).to(dev)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse2, B, H, S, S, D, scale)
    ro.backward(do.double())
    for name, got, ref in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        rel = ((got.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert rel < 3e-2, (name, rel)


@pytest.mark.parametrize("causal", [False, True])
def test_attention_very_negative_first_keys(dev, causal):
    """A row whose FIRST half tile scores far below -128 in log2 units (q = +8, k[:64] = -8 at D = 64, scale 1: -4096):
    the row's first frame must not be formed by rescaling the empty state (exp2(+4096) = inf, 0 * inf = NaN).  The causal
    variant makes query 0 see only key 0 with a very negative q0.k0.  Forward and backward vs fp64."""
    from lc2is_amd import ops
    B, H, S, D = 1, 1, 192, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    q = torch.randn(S, D, generator=g) * 0.125   # (ordinary rows: the usual 1/sqrt(D) score scale)
    k = torch.randn(S, D, generator=g)
    v = torch.randn(S, D, generator=g)
    q[0] = 8.0
    q[70] = 8.0
    k[:64] = -8.0
    q, k, v = (t.to(torch.bfloat16).to(dev) for t in (q, k, v))
    do = torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev)
    o, lse2 = ops.attention_fwd(q, k, v, B, H, S, S, D, 1.0, causal=causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse2).all()
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ro, rlse, _ = _ref_attention(qd, kd, vd, B, H, S, S, D, 1.0, causal, None)
    assert (o.double() - ro.detach()).abs().max().item() < 3e-2
    assert (lse2.double() * math.log(2.0) - rlse.detach()).abs().max().item() < 2e-2
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse2, B, H, S, S, D, 1.0, causal=causal)
    ro.backward(do.double())
    for name, got, ref in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        assert torch.isfinite(got.float()).all(), name
        rel = ((got.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert rel < 3e-2, (name, rel)


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,mask,packed", CASES)
def test_attention_bwd(dev, B, H, Sq, Sk, D, causal, mask, packed):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 999 + Sq + D)
    if packed:
        qkv = (torch.randn(B * Sq, 3 * H * D, generator=g)).to(torch.bfloat16).to(dev)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    else:
        q = torch.randn(B * Sq, H * D, generator=g).to(torch.bfloat16).to(dev)
        k = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
        v = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B * Sq, H * D, generator=g).to(torch.bfloat16).to(dev)
    kbias = None
    if mask:
        valid = torch.randint(1, Sk + 1, (B,), generator=g)
        kbias = torch.zeros(B, Sk)
        for i in range(B):
            kbias[i, valid[i]:] = float("-inf")
        kbias = kbias.to(dev)
    scale = 1.0 / math.sqrt(D)
    o, lse2 = ops.attention_fwd(q, k, v, B, H, Sq, Sk, D, scale, causal=causal, kbias=kbias)
    if packed:
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:]
        ops.attention_bwd(q, k, v, o, do, lse2, B, H, Sq, Sk, D, scale, causal=causal, kbias=kbias,
                          dq=dq, dk=dk, dv=dv)
    else:
        dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse2, B, H, Sq, Sk, D, scale, causal=causal, kbias=kbias)
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ro, _, _ = _ref_attention(qd, kd, vd, B, H, Sq, Sk, D, scale, causal, kbias)
    ro.backward(do.double())
    for name, got, ref in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        rel = ((got.double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert rel < 1.5e-2, (name, rel)
        assert (got.double() - ref).abs().max().item() < 0.05 * ref.abs().max().item() + 1e-3, name


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal", [(4, 12, 1025, 1025, 64, False), (3, 4, 400, 400, 64, True)])
def test_attention_bwd_is_bitwise_reproducible(dev, B, H, Sq, Sk, D, causal):
    """The two-launch backward has no atomics and a fixed summation order: repeated runs are bitwise equal."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(77 + Sq + D)
    q = torch.randn(B * Sq, H * D, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(B * Sk, H * D, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B * Sq, H * D, generator=g).to(torch.bfloat16).to(dev)
    scale = 1.0 / math.sqrt(D)
    o, lse2 = ops.attention_fwd(q, k, v, B, H, Sq, Sk, D, scale, causal=causal)
    runs = [[t.clone() for t in ops.attention_bwd(q, k, v, o, do, lse2, B, H, Sq, Sk, D, scale, causal=causal)] for _ in range(3)]
    for r in runs[1:]:
        for a, b_ in zip(runs[0], r):
            assert torch.equal(a, b_)
