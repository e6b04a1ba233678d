"""Randomised shape sweeps of the three kernel families every module leans on — NT GEMM (default plan: tile choice, persistent
form, ragged-row split), weight-gradient GEMM, fused attention forward + backward — against fp64 torch on the same bf16 inputs.
Seeds are fixed: the sweep is the same on every run; shapes mix multiples of the tile sizes with ragged ones."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _nt_cases():
    rnd = random.Random(20260403)
    cases = []
    for _ in range(28):
        M = rnd.choice([rnd.randint(1, 300), rnd.randint(300, 5000), 256 * rnd.randint(1, 40), 256 * rnd.randint(1, 40) + rnd.randint(1, 64),
                        1025 * rnd.randint(1, 8)])
        N = rnd.choice([4 * rnd.randint(1, 64), 8 * rnd.randint(8, 400), 256 * rnd.randint(1, 12), 768, 2304, 3072])
        K = 64 * rnd.randint(1, 48)
        epi = rnd.choice(["bf16", "f32", "f32+res", "both", "quick_gelu+aux", "relu"])
        cases.append((M, N, K, epi))
    return cases


@pytest.mark.parametrize("M,N,K,epi", _nt_cases())
def test_gemm_nt_random_shapes(dev, M, N, K, epi):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 31 + N * 7 + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * (K ** -0.5)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    z = a.double() @ w.double().T + bias.double()
    tol32 = 2e-6 * (K ** 0.5) + 1e-6
    if epi == "bf16":
        ob, _, _ = ops.gemm_nt(a, w, bias)
        assert _rel(ob.float(), z) < 4e-3
    elif epi == "f32":
        _, of, _ = ops.gemm_nt(a, w, bias, out_bf16=None, out_f32=True)
        assert _rel(of, z) < tol32
    elif epi == "f32+res":
        resid = torch.randn(M, N, generator=g).to(dev)
        _, of, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=True)
        assert _rel(of, z + resid.double()) < tol32
    elif epi == "both":
        ob, of, _ = ops.gemm_nt(a, w, bias, out_bf16=True, out_f32=True)
        assert _rel(of, z) < tol32 and _rel(ob.float(), z) < 4e-3
    elif epi == "quick_gelu+aux":
        ob, _, aux = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True)
        assert _rel(aux.float(), z) < 4e-3                                     # the saved pre-activation
        assert _rel(ob.float(), z * torch.sigmoid(1.702 * z)) < 6e-3
    else:
        ob, _, _ = ops.gemm_nt(a, w, bias, act=ops.ACT_RELU)
        assert _rel(ob.float(), torch.relu(z)) < 4e-3


def _tn_cases():
    rnd = random.Random(77)
    return [(rnd.choice([rnd.randint(1, 200), rnd.randint(200, 9000), 64 * rnd.randint(1, 100)]),
             8 * rnd.randint(1, 96), 8 * rnd.randint(1, 96), rnd.random() < 0.4) for _ in range(16)] + \
           [(4100, 768, 3072, False), (2050, 2304, 768, True), (33000, 256, 256, False)]


@pytest.mark.parametrize("M,N,K,accumulate", _tn_cases())
def test_gemm_tn_random_shapes(dev, M, N, K, accumulate):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 13 * N + 101 * K)
    dy = _bf(torch.randn(M, N, generator=g)).to(dev)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    dw0 = torch.randn(N, K, generator=g).to(dev)
    db0 = torch.randn(N, generator=g).to(dev)
    dw, db = dw0.clone(), db0.clone()
    ops.gemm_tn(dy, x, dw, accumulate=accumulate, db=db)
    ref = dy.double().T @ x.double() + (dw0.double() if accumulate else 0)
    refb = dy.double().sum(0) + (db0.double() if accumulate else 0)
    assert _rel(dw, ref) < 3e-6 * (M ** 0.5) + 1e-6
    assert _rel(db, refb) < 3e-6 * (M ** 0.5) + 1e-6


def _attn_cases():
    rnd = random.Random(5)
    cases = []
    for _ in range(14):
        D = rnd.choice([64, 64, 96, 128])
        B, H = rnd.randint(1, 3), rnd.randint(1, 4)
        causal = rnd.random() < 0.3
        Sq = rnd.choice([rnd.randint(1, 130), rnd.randint(130, 700), 64 * rnd.randint(1, 9), 64 * rnd.randint(1, 9) + 1])
        Sk = Sq if causal else rnd.choice([Sq, rnd.randint(1, 600), 64 * rnd.randint(1, 8) + rnd.choice([0, 1, 33])])
        cases.append((B, H, Sq, Sk, D, causal, rnd.random() < 0.4))
    return cases


@pytest.mark.parametrize("B,H,Sq,Sk,D,causal,pad", _attn_cases())
def test_attention_random_shapes(dev, B, H, Sq, Sk, D, causal, pad):
    """forward and the three gradients vs fp64 softmax attention; `pad` masks a random suffix of every batch's keys (additive
    -inf key bias, as the padding masks of the text tower / decoder do) but never all of them."""
    from lc2is_amd import ops
    C = H * D
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Sq * 7 + Sk)
    q = _bf(torch.randn(B * Sq, C, generator=g)).to(dev)
    k = _bf(torch.randn(B * Sk, C, generator=g)).to(dev)
    v = _bf(torch.randn(B * Sk, C, generator=g)).to(dev)
    do = _bf(torch.randn(B * Sq, C, generator=g) * 0.5).to(dev)
    kbias = None
    if pad:
        keep = torch.randint(1, Sk + 1, (B,), generator=g)
        kbias = torch.zeros(B, Sk)
        for b in range(B):
            kbias[b, int(keep[b]):] = float("-inf")
        kbias = kbias.to(dev)
    sc = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, B, H, Sq, Sk, D, sc, causal=causal, kbias=kbias)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, B, H, Sq, Sk, D, sc, causal=causal, kbias=kbias)
    qq = q.double().view(B, Sq, H, D).transpose(1, 2).requires_grad_(True)
    kk = k.double().view(B, Sk, H, D).transpose(1, 2).requires_grad_(True)
    vv = v.double().view(B, Sk, H, D).transpose(1, 2).requires_grad_(True)
    s = qq @ kk.transpose(-1, -2) * sc
    if kbias is not None:
        s = s + kbias.double()[:, None, None, :]
    if causal:
        s = s + torch.full((Sq, Sk), float("-inf"), dtype=torch.float64, device=dev).triu(1)
    ro = torch.softmax(s, -1) @ vv
    ro.backward(do.double().view(B, Sq, H, D).transpose(1, 2))
    back = lambda t, S_: t.transpose(1, 2).reshape(B * S_, C)
    assert _rel(o.float(), back(ro.detach(), Sq)) < 8e-3
    assert _rel(dq.float(), back(qq.grad, Sq)) < 1.5e-2
    assert _rel(dk.float(), back(kk.grad, Sk)) < 1.5e-2
    assert _rel(dv.float(), back(vv.grad, Sk)) < 1.5e-2
