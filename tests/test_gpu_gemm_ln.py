"""GPU parity of the dense-layer and LayerNorm kernels against plain fp32/fp64 PyTorch (through the C ABI)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K,cfg", [
    (300, 192, 128, 1), (300, 192, 128, 2), (300, 192, 128, 3), (1025, 768, 768, 0), (4100, 2304, 768, 0),
    (129, 152, 512, 0), (77, 64, 64, 3), (2050, 768, 3072, 2), (2050, 768, 3072, 4), (300, 192, 128, 4),
    (515, 384, 192, 6), (4100, 768, 768, 4),
])
def test_gemm_nt_plain(dev, M, N, K, cfg):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ob, of, _ = ops.gemm_nt(a, w, bias, out_bf16=True, out_f32=True, tile_cfg=cfg)
    ref = a.double() @ w.double().T + bias.double()
    assert _rel(of, ref) < 2e-6 * (K ** 0.5)  # fp32 accumulate of exact bf16 products
    assert (of.double() - ref).abs().max().item() < 1e-3
    assert _rel(ob.float(), ref) < 4e-3  # bf16 output rounding (2^-9 relative)


@pytest.mark.parametrize("M,N,K,cfg", [(1025, 768, 768, 4), (300, 196, 128, 4), (515, 328, 192, 6), (4100, 768, 3072, 0),
                                         (70, 64, 64, 4)])
@pytest.mark.parametrize("mode", ["resid", "plain", "inplace"])
def test_gemm_nt_fp32_only_output(dev, M, N, K, cfg, mode):
    """fp32-only outputs of the LDS-DMA kernels leave through the staged fp32 epilogue (whole-line buffer stores):
    ragged rows / columns, padded leading dimension, missing residual, residual aliased with the output."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 11 + N + cfg)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    big = torch.full((M + 3, N + 8), 7.0, device=dev)     # guard rows / columns around a strided output view
    out = big[1:M + 1, 4:N + 4]
    resid = torch.randn(M, N, generator=g).to(dev)
    ref = a.double() @ w.double().T + bias.double()
    if mode == "resid":
        _, of, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=out, tile_cfg=cfg)
        ref = ref + resid.double()
    elif mode == "inplace":
        out.copy_(resid)
        _, of, _ = ops.gemm_nt(a, w, bias, resid=out, out_bf16=None, out_f32=out, tile_cfg=cfg)
        ref = ref + resid.double()
    else:
        _, of, _ = ops.gemm_nt(a, w, bias, out_bf16=None, out_f32=out, tile_cfg=cfg)
    assert of.data_ptr() == out.data_ptr()
    assert _rel(of, ref) < 2e-6 * (K ** 0.5)
    assert (of.double() - ref).abs().max().item() < 1e-3
    guard = big.clone()
    guard[1:M + 1, 4:N + 4] = 7.0
    assert torch.equal(guard, torch.full_like(big, 7.0))    # nothing written outside the view


def test_gemm_nt_asymmetric_identity(dev):
    """A = I against an asymmetric W catches a transposed C write (guide §3)."""
    from lc2is_amd import ops
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16, device=dev)
    w = _bf(torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 125.0).to(dev)
    _, of, _ = ops.gemm_nt(a, w, None, out_bf16=None, out_f32=True)
    assert torch.equal(of, w.float().T.contiguous())


@pytest.mark.parametrize("cfg", [0, 4, 6])
@pytest.mark.parametrize("act", ["quick_gelu", "relu"])
def test_gemm_nt_activation_and_backward_epilogue(dev, act, cfg):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 515, 384, 192
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    code = ops.ACT_QUICK_GELU if act == "quick_gelu" else ops.ACT_RELU
    ob, of, z = ops.gemm_nt(a, w, bias, act=code, resid=resid, out_bf16=True, out_f32=True, aux_out=True, tile_cfg=cfg)
    zr = a.double() @ w.double().T + bias.double()
    ar = zr * torch.sigmoid(1.702 * zr) if act == "quick_gelu" else zr.clamp_min(0)
    assert _rel(z.float(), zr) < 4e-3
    assert _rel(of, ar + resid.double()) < 1e-5
    assert _rel(ob.float(), ar + resid.double()) < 4e-3
    # backward epilogue: dz = (dy @ w2) * act'(saved)
    dy = _bf(torch.randn(M, K, generator=g)).to(dev)
    saved = z if act == "quick_gelu" else ob  # relu saves its output
    dcode = ops.ACT_DQUICK_GELU if act == "quick_gelu" else ops.ACT_DRELU
    dz, _, _ = ops.gemm_nt(dy, w, None, act=dcode, aux_in=saved, tile_cfg=cfg)
    s = saved.double()
    if act == "quick_gelu":
        sg = torch.sigmoid(1.702 * s)
        d = sg * (1 + 1.702 * s * (1 - sg))
    else:
        d = (s > 0).double()
    ref = (dy.double() @ w.double().T) * d
    assert _rel(dz.float(), ref) < 5e-3


@pytest.mark.parametrize("M,N,K", [(2050, 768, 768), (300, 200, 128), (66000, 512, 64), (70, 64, 64), (32800, 768, 192),
                                   (33000, 512, 320), (4100, 1024, 3072)])
def test_gemm_nt_ping_pong_is_bitwise_equal(dev, M, N, K):
    """tile_cfg 15 — the persistent kernel whose two groups of four waves run half a K step apart (one group computes alone on the
    matrix pipe while its SIMD partners request the next K tile by inline-asm LDS-DMA; counted `s_waitcnt vmcnt(n)`, nothing waits
    for vmcnt(0) in the steady state) — against tile_cfg 4: same accumulation order, so the outputs are bitwise equal.  K = 64 ..
    3072 covers one K tile, two / three (only the sequence-counted waits), five (odd count: the last K tile sits in buffer 0 when
    the next output tile's first requests go out) and 48; several output tiles per block (66000 x 512 = 516 tiles) exercise the
    seam where the epilogue's stores sit between the DMA requests in the in-order vmcnt queue.  A wait that is too weak reads a K
    tile before its DMA landed; a request made too early overwrites fragments the other wave group still reads."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    saved = _bf(torch.randn(M, N, generator=g)).to(dev)
    for rep in range(3):   # (warm caches on the later runs: a different DMA timing)
        o4, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=4)
        o15, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=15)
        assert torch.equal(o4, o15)
        o4, _, z4 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=4)
        o15, _, z15 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=15)
        assert torch.equal(o4, o15) and torch.equal(z4, z15)
        d4, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=4)
        d15, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=15)
        assert torch.equal(d4, d15)
        r4, _, _ = ops.gemm_nt(a, w, bias, act=ops.ACT_RELU, tile_cfg=4)          # (round 5: the decoders' linear1 and its dgrad)
        r15, _, _ = ops.gemm_nt(a, w, bias, act=ops.ACT_RELU, tile_cfg=15)
        assert torch.equal(r4, r15)
        d4, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DRELU, aux_in=saved, tile_cfg=4)
        d15, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DRELU, aux_in=saved, tile_cfg=15)
        assert torch.equal(d4, d15)
    ref = a.double() @ w.double().T + bias.double()
    assert _rel(ops.gemm_nt(a, w, bias, tile_cfg=15)[0].float(), ref) < 4e-3


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (2050, 768, 128), (300, 384, 64), (4100, 1152, 320), (32768, 768, 192)])
def test_gemm_nt_w384_is_bitwise_equal(dev, M, N, K):
    """tile_cfg 16 — 256x384 tiles (8 waves x 128x96, fp32-only output through LDS in 384-byte row segments; the whole 160 KiB of
    LDS for the two stages) — against tile_cfg 4 on the residual-stream form out = resid + a.w^T + bias: same K order, so bitwise
    equal; ragged rows (2050, 300, 4100) exercise the range-checked residual loads / stores, N = 384 / 1152 one and three column
    tiles, K = 64 a single K tile.  Also without bias / residual (the dgrads into the fp32 stream)."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    for rep in range(2):
        _, f4, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True, tile_cfg=4)
        _, f16, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True, tile_cfg=16)
        assert torch.equal(f4, f16)
        _, g4, _ = ops.gemm_nt(a, w, None, out_bf16=False, out_f32=True, tile_cfg=4)
        _, g16, _ = ops.gemm_nt(a, w, None, out_bf16=False, out_f32=True, tile_cfg=16)
        assert torch.equal(g4, g16)
        b4, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=4)     # bf16-only output (the dgrads that feed LayerNorm / attention backward)
        b16, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=16)
        assert torch.equal(b4, b16)
        b4, _, _ = ops.gemm_nt(a, w, None, tile_cfg=4)
        b16, _, _ = ops.gemm_nt(a, w, None, tile_cfg=16)
        assert torch.equal(b4, b16)
    inplace = resid.clone()   # the residual stream updated in place (out_f32 is resid)
    ops.gemm_nt(a, w, bias, resid=inplace, out_bf16=False, out_f32=inplace, tile_cfg=16)
    assert torch.equal(inplace, f4)
    ref = a.double() @ w.double().T + bias.double() + resid.double()
    assert _rel(f16, ref) < 2e-6 * (K ** 0.5)  # fp32 accumulate of exact bf16 products


@pytest.mark.parametrize("M,N,K", [(32, 768, 3072), (64, 768, 768), (17, 200, 64), (1, 16, 320), (48, 1152, 2304), (32, 3072, 768)])
def test_gemm_nt_rows_tail_is_bitwise_equal(dev, M, N, K):
    """tile_cfg 17 — the ragged-row tail of the exact-round plans: one wave per 16 x 16 outputs and the whole K, operands straight
    from global memory in the MFMA fragment layout, 16 K chunks in flight — against tile_cfg 4 (same MFMA, same operand roles, same
    sequential K order: bitwise equal).  K = 64 and 320 are not multiples of the 512-element prefetch window (chunks past K must not be
    accumulated), N = 200 has a ragged last column group, M = 1 / 17 ragged row fragments."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    _, f4, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True, tile_cfg=4)
    _, f17, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True, tile_cfg=17)
    assert torch.equal(f4, f17)
    _, g4, _ = ops.gemm_nt(a, w, None, out_bf16=False, out_f32=True, tile_cfg=4)
    _, g17, _ = ops.gemm_nt(a, w, None, out_bf16=False, out_f32=True, tile_cfg=17)
    assert torch.equal(g4, g17)
    # the bf16 epilogues it takes over from the 64x64 kernel in front of the persistent forms (fc1: quick_gelu + saved z; dfc2: x gelu'(z))
    saved = _bf(torch.randn(M, N, generator=g)).to(dev)
    o4, _, z4 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=4)
    o17, _, z17 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=17)
    assert torch.equal(o4, o17) and torch.equal(z4, z17)
    d4, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=4)
    d17, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=17)
    assert torch.equal(d4, d17)


def test_gemm_nt_auto_plan_picks_exact_round_and_matches(dev):
    """M = 32 800, N = 768 (the residual-stream GEMMs of the headline step): the automatic plan is 256 tiles of 256x384 + the row
    tail; bitwise equal to the single-launch 256x256 plan, residual updated in place as the modules do."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 32800, 768, 768
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    _, f4, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True, tile_cfg=4)
    x = resid.clone()
    ops.gemm_nt(a, w, bias, resid=x, out_bf16=False, out_f32=x, tile_cfg=0)
    assert torch.equal(x, f4)
    b4, _, _ = ops.gemm_nt(a, w, None, tile_cfg=4)   # bf16-only: 256 tiles of 256x384 + the row tail as well
    b0, _, _ = ops.gemm_nt(a, w, None, tile_cfg=0)
    assert torch.equal(b4, b0)


@pytest.mark.parametrize("N,K", [(3072, 128), (3072, 768), (2304, 192)])
def test_gemm_nt_auto_plan_persistent_with_folded_tail(dev, N, K):
    """M = 64*256 + 32 rows with bf16 outputs: the automatic plan is the persistent ping-pong kernel (or 256x384 tiles for N = 2304)
    over the whole tiles with the 32 ragged rows FOLDED into the same launch (fragment jobs after each block's last tile; the
    derivative epilogue keeps a separate tail launch) — bitwise equal to the single-launch 256x256 plan for the three epilogues of
    the tower: bias, quick_gelu + saved pre-activation, x quick_gelu'(saved)."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(N + K)
    M = 64 * 256 + 32
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    saved = _bf(torch.randn(M, N, generator=g)).to(dev)
    for rep in range(2):
        o4, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=4)
        o0, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=0)
        assert torch.equal(o4, o0)
        o4, _, z4 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=4)
        o0, _, z0 = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, aux_out=True, tile_cfg=0)
        assert torch.equal(o4, o0) and torch.equal(z4, z0)
        d4, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=4)
        d0, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_DQUICK_GELU, aux_in=saved, tile_cfg=0)
        assert torch.equal(d4, d0)


@pytest.mark.parametrize("N,K", [(3072, 128), (768, 192), (2304, 128)])
def test_gemm_nt_auto_plan_ragged_rows(dev, N, K):
    """M = 64*256 + 32 rows (the B x 1025-token shape): the automatic plan peels the ragged rows into a small-tile
    launch and / or picks 128x384 tiles; every output must match the single-launch 256x256 plan."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(N + K)
    M = 64 * 256 + 32
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    outs = {}
    for cfg in (4, 0):
        ob, of, d = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU_GRAD, resid=resid, out_bf16=True, out_f32=True,
                                aux_out=True, tile_cfg=cfg)
        dz, _, _ = ops.gemm_nt(a, w, None, act=ops.ACT_MUL_AUX, aux_in=d, tile_cfg=cfg)
        outs[cfg] = (ob.float(), of, d.float(), dz.float())
    for x, y in zip(outs[0], outs[4]):
        assert _rel(x, y.double()) < 1e-6
    ref = a[-40:].double() @ w.double().T + bias.double()
    sg = torch.sigmoid(1.702 * ref)
    assert _rel(outs[0][1][-40:], ref * sg + resid[-40:].double()) < 1e-5      # the peeled rows themselves


@pytest.mark.parametrize("cfg", [0, 3, 4])
def test_gemm_nt_gelu_derivative_pair(dev, cfg):
    """Codes 5/6: the forward epilogue saves quick_gelu'(z) (bf16) next to the activation; the backward epilogue
    multiplies by it.  cfg 3 = direct epilogue, 4 = LDS-staged epilogue."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(6)
    M, N, K = 515, 384, 192
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ob, _, d = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU_GRAD, aux_out=True, tile_cfg=cfg)
    zr = a.double() @ w.double().T + bias.double()
    sg = torch.sigmoid(1.702 * zr)
    assert _rel(ob.float(), zr * sg) < 4e-3
    dref = sg * (1 + 1.702 * zr * (1 - sg))
    assert _rel(d.float(), dref) < 4e-3
    ob2, _, _ = ops.gemm_nt(a, w, bias, act=ops.ACT_QUICK_GELU, tile_cfg=cfg)       # same activation as code 1
    assert _rel(ob.float(), ob2.float().double()) < 1e-6
    dy = _bf(torch.randn(M, K, generator=g)).to(dev)
    dz, _, _ = ops.gemm_nt(dy, w, None, act=ops.ACT_MUL_AUX, aux_in=d, tile_cfg=cfg)
    assert _rel(dz.float(), (dy.double() @ w.double().T) * d.double()) < 4e-3
    with pytest.raises(RuntimeError):
        ops.gemm_nt(dy, w, None, act=ops.ACT_MUL_AUX, tile_cfg=cfg)                    # aux_in is required


@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (1025, 768, 768), (4100, 2304, 768), (2050, 152, 512),
                                   (64, 64, 64), (5000, 3072, 768), (33, 8, 8)])
def test_gemm_tn(dev, M, N, K):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dy = _bf(torch.randn(M, N, generator=g)).to(dev)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    dw = ops.gemm_tn(dy, x)
    ref = dy.double().T @ x.double()
    assert _rel(dw, ref) < 1e-5
    dw2 = ops.gemm_tn(dy, x, dw.clone(), accumulate=True)
    assert _rel(dw2, 2 * ref) < 1e-5
    # bitwise reproducible
    assert torch.equal(ops.gemm_tn(dy, x), dw)
    # fused bias gradient
    db = torch.full((N,), 7.0, device=dev)
    dw3 = ops.gemm_tn(dy, x, torch.empty_like(dw), db=db)
    assert torch.equal(dw3, dw) and _rel(db, dy.double().sum(0)) < 1e-5
    db2 = db.clone()
    ops.gemm_tn(dy, x, dw3, accumulate=True, db=db2)
    assert _rel(db2, 2 * dy.double().sum(0)) < 1e-5


def test_gemm_tn_asymmetric(dev):
    from lc2is_amd import ops
    M = 128
    dy = torch.eye(M, dtype=torch.bfloat16, device=dev)[:, :64].contiguous()  # [M,64]: dW[n] = x[n]
    x = _bf((torch.arange(M * 192, dtype=torch.float32).reshape(M, 192) % 253) - 126.0).to(dev)
    dw = ops.gemm_tn(dy, x)
    assert torch.equal(dw, x.float()[:64])


def test_colsum(dev):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    dy = _bf(torch.randn(4100, 2304, generator=g)).to(dev)
    db = ops.colsum(dy)
    assert _rel(db, dy.double().sum(0)) < 1e-5
    v = dy[:, 768:1536]  # strided view
    assert _rel(ops.colsum(v), v.double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,C", [(1025, 768), (37, 64), (513, 512), (260, 1024), (100, 192), (9, 2048)])
def test_layernorm_fwd_bwd(dev, M, C):
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(C + M)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    yb, yf, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5, out_bf16=True, out_f32=True)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (C,), gd, bd, 1e-5)
    assert (yf.double() - ref).abs().max().item() < 2e-5
    assert _rel(yb.float(), ref) < 4e-3
    # no-bias variant (torch 2.10 bias=False drift of DecoderLayer norms)
    _, yf2, _, _ = ops.layernorm_fwd(x, gamma, None, 1e-5, out_bf16=None, out_f32=True)
    assert (yf2.double() - (ref - bd)).abs().max().item() < 2e-5
    dy = torch.randn(M, C, generator=g).to(dev)
    dres = torch.randn(M, C, generator=g).to(dev)
    ref.backward(dy.double())
    dxf, dxb, dg, db = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres)
    assert _rel(dxf, xd.grad + dres.double()) < 1e-5
    assert _rel(dxb.float(), xd.grad + dres.double()) < 4e-3
    assert _rel(dg, gd.grad) < 1e-5
    assert _rel(db, bd.grad) < 1e-5
    # bf16 dy path
    dyb = _bf(dy)
    dxf2, _, _, _ = ops.layernorm_bwd(dyb, x, gamma, mean, rstd, want_bf16=False)
    xd.grad = None
    torch.nn.functional.layer_norm(xd, (C,), gd, bd, 1e-5).backward(dyb.double())
    assert _rel(dxf2, xd.grad) < 1e-5


@pytest.mark.parametrize("M,N,K", [(2050 + 32, 768, 768), (4100, 768, 3072), (300, 512, 128), (2080, 1024, 256), (66000, 512, 64), (70, 64, 64)])
def test_gemm_nt_bf16_residual_epilogue(dev, M, N, K):
    """Round 5: `out = bf16(a @ w.T + bias + resid)` with a BF16 residual (the bf16 residual stream; LC2IS_ACT_ADD_AUX).  fp32 add,
    one rounding: against fp64 of the same bf16 inputs the result is exact to bf16 rounding; every tile plan that takes the
    problem — the default dispatch (256x384 tiles + folded ragged rows at N = 768, the persistent ping-pong form, 128x128 tiles)
    and the forced single-kernel plans — gives bitwise the same tensor."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = _bf(torch.randn(M, N, generator=g) * 3).to(dev)
    ref = a.double() @ w.double().T + bias.double() + resid.double()
    out0, of, _ = ops.gemm_nt(a, w, bias, resid=resid)
    assert of is None and out0.dtype == torch.bfloat16
    assert _rel(out0.float(), ref) < 3e-3
    assert (out0.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-6   # one bf16 rounding of the fp32 sum
    for cfg in (4, 6, 1, 3, 15, 16):
        try:
            o, _, _ = ops.gemm_nt(a, w, bias, resid=resid, tile_cfg=cfg)
        except RuntimeError:
            continue          # a plan that does not take this shape (16: N % 384, 15: ...)
        assert torch.equal(o, out0), cfg
    # the spelled-out form of the same epilogue
    o2, _, _ = ops.gemm_nt(a, w, bias, act=ops.ACT_ADD_AUX, aux_in=resid)
    assert torch.equal(o2, out0)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(a, w, bias, resid=resid, act=ops.ACT_RELU)


@pytest.mark.parametrize("M,C", [(300, 768), (77, 1024), (1025, 64)])
def test_layernorm_bf16_streams(dev, M, C):
    """Round 5: the residual stream (x) and the gradient stream (dres) may arrive in bf16.  The kernels widen them on the way
    in and otherwise run the fp32 arithmetic, so against fp64 of the SAME bf16-valued inputs the fp32 outputs are exact to
    fp32 rounding, and feeding the widened tensors through the fp32 entry gives bitwise the same result."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(C * 3 + M)
    xb = _bf((torch.randn(M, C, generator=g) * 2 + 0.5).to(dev))
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    yb, yf, mean, rstd = ops.layernorm_fwd(xb, gamma, beta, 1e-5, out_bf16=True, out_f32=True)
    yb2, yf2, mean2, rstd2 = ops.layernorm_fwd(xb.float(), gamma, beta, 1e-5, out_bf16=True, out_f32=True)
    assert torch.equal(yf, yf2) and torch.equal(yb, yb2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    xd = xb.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (C,), gd, bd, 1e-5)
    assert (yf.double() - ref).abs().max().item() < 2e-5
    dyb = _bf(torch.randn(M, C, generator=g).to(dev))
    dresb = _bf(torch.randn(M, C, generator=g).to(dev))
    ref.backward(dyb.double())
    want = xd.grad + dresb.double()
    for x_in, r_in in ((xb, dresb), (xb.float(), dresb), (xb, dresb.float())):
        dxf, dxb, dg, db = ops.layernorm_bwd(dyb, x_in, gamma, mean, rstd, dres=r_in)
        assert _rel(dxf, want) < 1e-5
        assert _rel(dxb.float(), want) < 4e-3
        assert _rel(dg, gd.grad) < 1e-5 and _rel(db, bd.grad) < 1e-5
    a = ops.layernorm_bwd(dyb, xb, gamma, mean, rstd, dres=dresb)
    b = ops.layernorm_bwd(dyb, xb.float(), gamma, mean, rstd, dres=dresb.float())
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    # bf16-only output (what the towers' backward uses): no fp32 tensor is written
    # bf16-only output (what the towers' backward uses: the lean kernel): no fp32 tensor is written.  Same formulas as the generic
    # kernel; hipcc contracts multiply-adds differently in the two bodies, so a few outputs may land on the neighbouring bf16 value
    dxf3, dxb3, _, _ = ops.layernorm_bwd(dyb, xb, gamma, mean, rstd, dres=dresb, want_f32=False)
    assert dxf3 is None and _rel(dxb3.float(), a[1].float()) < 1e-3
    assert (dxb3 != a[1]).float().mean().item() < 0.02


@pytest.mark.parametrize("M,C", [(32800, 768), (2416, 512), (4100, 1024), (16208, 1024), (333, 64), (3, 768)])
def test_layernorm_bwd_lean_kernel(dev, M, C):
    """Round 5: the towers' backward (bf16 dy / residual-path gradient / output) runs a lean kernel with next-row prefetch on a
    smaller grid.  Same per-element formulas as the generic kernel (reached through the fp32-dres entry with the widened tensor): dx
    agrees to the last bf16 bit on all but a few elements, dgamma / dbeta to fp32 summation order, and everything agrees with fp64;
    x in fp32 and in bf16; with and without a residual-path gradient; more rows than one grid round, fewer rows than a block."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x32 = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    dyb = _bf(torch.randn(M, C, generator=g).to(dev))
    dresb = _bf(torch.randn(M, C, generator=g).to(dev))
    for x in (x32, _bf(x32)):
        _, _, mean, rstd = ops.layernorm_fwd(x, gamma, None, 1e-5)
        for dres_lean, dres_gen in ((dresb, dresb.float()), (None, None)):
            lean = ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dres=dres_lean, want_f32=False)              # lean kernel
            gen = ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dres=dres_gen, want_f32=True)                  # generic kernel
            # (same formulas; hipcc contracts multiply-adds differently in the two bodies: a few outputs sit on the neighbouring bf16 value)
            assert lean[0] is None and _rel(lean[1].float(), gen[1].float()) < 1e-3 and (lean[1] != gen[1]).float().mean().item() < 0.02
            assert _rel(lean[2], gen[2]) < 2e-6 and _rel(lean[3], gen[3]) < 2e-6
        xd = x.double().requires_grad_(True)
        gd = gamma.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xd, (C,), gd, None, 1e-5).backward(dyb.double())
        dxb, dg = ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dres=dresb, want_f32=False)[1:3]
        assert _rel(dxb.float(), xd.grad + dresb.double()) < 4e-3
        assert _rel(dg, gd.grad) < 1e-5


def test_layernorm_bwd_deferred_param_grads_are_bitwise_equal(dev):
    """The dgamma / dbeta reductions of several LayerNorm backward calls in one grouped launch (ops.ln_defer_begin/flush,
    opened by nn.base.WgradBatch): same fixed-order sums as the per-call second launch, incl. accumulate and a vector that
    two calls of the scope write (the second is reduced in a later launch)."""
    from lc2is_amd import ops
    torch.manual_seed(5)
    shapes = [(4100, 768), (333, 768), (2050, 512), (77, 64)]
    cases = []
    for M, C in shapes:
        x = torch.randn(M, C, device=dev)
        gamma = torch.randn(C, device=dev)
        _, _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros(C, device=dev))
        cases.append((torch.randn(M, C, device=dev).bfloat16(), x, gamma, mean, rstd))
    base_g = [torch.randn(C, device=dev) for _, C in shapes]
    base_b = [torch.randn(C, device=dev) for _, C in shapes]

    def run(deferred):
        dgs, dbs = [t.clone() for t in base_g], [t.clone() for t in base_b]
        prev = ops.ln_defer_begin() if deferred else None
        outs = []
        for i, (dy, x, gamma, mean, rstd) in enumerate(cases):
            acc = i % 2 == 1
            outs.append(ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=dgs[i], dbeta=dbs[i], accumulate=acc)[0])
        # a second call into the vectors of case 0 (a module used twice inside one scope)
        dy, x, gamma, mean, rstd = cases[0]
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=dgs[0], dbeta=dbs[0], accumulate=True)
        if deferred:
            assert len(ops._ln_defer) == len(cases) + 1
            ops.ln_defer_end(prev)
            assert ops._ln_defer is prev
        torch.cuda.synchronize()
        return dgs, dbs, outs

    g0, b0, o0 = run(False)
    g1, b1, o1 = run(True)
    for a, b in zip(g0 + b0 + o0, g1 + b1 + o1):
        assert torch.equal(a, b)
    ref = (cases[0][0].float() * ((cases[0][1] - cases[0][3][:, None]) * cases[0][4][:, None])).sum(0) * 2
    assert ((g1[0] - ref).norm() / ref.norm()).item() < 1e-3


def test_gemm_nt_default_plan_matches_single_kernel_on_the_step_shapes(dev):
    """The default dispatch at the headline row count (32 x 1025 = 128 row tiles + 32 ragged rows: persistent form for the
    bf16 outputs, ragged rows peeled into a small-tile launch for the fp32 ones) against the one-kernel 256x256 plan, for
    one layer's GEMM shapes."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    M = 32 * 1025
    for (N, K) in [(2304, 768), (768, 768), (768, 2304)]:
        a = _bf(torch.randn(M, K, generator=g)).to(dev)
        w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        o0, _, _ = ops.gemm_nt(a, w, bias)
        o4, _, _ = ops.gemm_nt(a, w, bias, tile_cfg=4)
        assert torch.equal(o0, o4), (N, K)
        resid = torch.randn(M, N, generator=g).to(dev)
        _, f0, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=True)
        _, f4, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=True, tile_cfg=4)
        assert torch.equal(f0, f4), (N, K)


def test_gemm_tn_grouped(dev):
    """One grid for several weight gradients == the per-problem launches (fp32, reproducible; accumulate + fused bias)."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    M = 4100
    shapes = [(256, 768), (768, 256), (256, 256), (512, 256)]
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dy = _bf(torch.randn(M, N + 64, generator=g)).to(dev)[:, :N]            # strided view (ldy > N), like dqkv slices
        x = _bf(torch.randn(M, K, generator=g)).to(dev)
        acc = i % 2 == 1
        dw = torch.full((N, K), 2.0, device=dev)
        db = torch.full((N,), 3.0, device=dev) if i != 2 else None
        probs.append((dy, x, dw, db, acc))
        ref_w = dy.double().T @ x.double() + (2.0 if acc else 0.0)
        ref_b = dy.double().sum(0) + (3.0 if acc else 0.0)
        refs.append((ref_w, ref_b))
    ops.gemm_tn_grouped(probs)
    for (dy, x, dw, db, acc), (rw, rb) in zip(probs, refs):
        assert _rel(dw, rw) < 1e-5
        if db is not None:
            assert _rel(db, rb) < 1e-5
    again = [(dy, x, torch.full_like(dw, 2.0), None if db is None else torch.full_like(db, 3.0), acc) for dy, x, dw, db, acc in probs]
    ops.gemm_tn_grouped(again)
    for a, b in zip(again, probs):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))   # bitwise reproducible
    with pytest.raises(RuntimeError):
        ops.gemm_tn_grouped([(probs[0][0][:, :204], probs[0][1], torch.empty(204, 768, device=dev), None, False)] * 2)   # N % 8


def test_gemm_tn_grouped_small_tiles(dev):
    """Groups with N or K not a multiple of 256 (Swin: 96 / 192 / 384 channels, 4x MLP) run on the 128x128 body: same results
    contract as the per-problem launches, ragged tiles included."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(29)
    M = 9000
    shapes = [(96, 96), (288, 96), (384, 96), (96, 384), (200, 136), (768, 192)]
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dy = _bf(torch.randn(M, N + 8, generator=g)).to(dev)[:, :N]
        x = _bf(torch.randn(M, K, generator=g)).to(dev)
        acc = i % 2 == 0
        dw = torch.full((N, K), 2.0, device=dev)
        db = torch.full((N,), 3.0, device=dev) if i != 1 else None
        probs.append((dy, x, dw, db, acc))
        refs.append((dy.double().T @ x.double() + (2.0 if acc else 0.0), dy.double().sum(0) + (3.0 if acc else 0.0)))
    assert all(ops.gemm_tn_groupable(p[0], p[1]) for p in probs)
    ops.gemm_tn_grouped(probs)
    for (dy, x, dw, db, acc), (rw, rb) in zip(probs, refs):
        assert _rel(dw, rw) < 1e-5
        if db is not None:
            assert _rel(db, rb) < 1e-5
    again = [(dy, x, torch.full_like(dw, 2.0), None if db is None else torch.full_like(db, 3.0), acc) for dy, x, dw, db, acc in probs]
    ops.gemm_tn_grouped(again)
    for a, b in zip(again, probs):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))


def test_gemm_tn_grouped_whole_tower_table_and_tail_split(dev):
    """More than 16 problems go through the device-side descriptor table; with > 256 output tiles and a ragged last round
    the planner keeps the bulk as full-length blocks and splits the smallest problems (slabs + ordered reduce for those
    only).  59 problems x (1..6 tiles) = 297 tiles, 41 past one round: the smallest problems are split, the rest run full length."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(23)
    M = 4160
    x_by_k = {K: _bf(torch.randn(M, K, generator=g)).to(dev) for K in (256, 512, 768)}
    shapes = [(256, 256)] * 5 + [(256, 512)] * 4 + [(512, 256)] * 4 + [(768, 256)] * 4 + [(512, 512)] * 3 + \
             [(512, 768)] * 3 + [(768, 512)] * 36
    assert len(shapes) > 16 and sum((n // 256) * (k // 256) for n, k in shapes) > 256
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dy = _bf(torch.randn(M, N, generator=g)).to(dev)
        x = x_by_k[K]
        acc = i % 3 == 1
        dw = torch.full((N, K), 2.0, device=dev)
        db = torch.full((N,), 3.0, device=dev) if i % 4 != 2 else None
        probs.append((dy, x, dw, db, acc))
        refs.append((dy.double().T @ x.double() + (2.0 if acc else 0.0), dy.double().sum(0) + (3.0 if acc else 0.0)))
    ops.gemm_tn_grouped(probs)
    for (dy, x, dw, db, acc), (rw, rb) in zip(probs, refs):
        assert _rel(dw, rw) < 1e-5
        if db is not None:
            assert _rel(db, rb) < 1e-5
    again = [(dy, x, torch.full_like(dw, 2.0), None if db is None else torch.full_like(db, 3.0), acc) for dy, x, dw, db, acc in probs]
    ops.gemm_tn_grouped(again)
    for a, b in zip(again, probs):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3]))   # bitwise reproducible
    with pytest.raises(RuntimeError):
        ops.gemm_tn_grouped(probs * 3)      # 177 problems > LC2IS_TN_GROUP_MAX


def test_cu_budget_changes_plans_not_results(dev):
    """lc2is_set_cu_budget(n): the tile planners count rounds over n CUs (the persistent kernels launch n blocks, the exact-round
    256x384 plan is dropped where it no longer saves a round, the grouped weight-gradient plan fills rounds of n) — what the
    data-parallel reducer sets while RCCL's channels hold CUs.  Every NT plan is bitwise equal, so those results must not move; the
    weight gradient's M-split follows its plan, so it may move by fp32 rounding."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 32800, 768, 256
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    w3 = _bf(torch.randn(3072, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    dy = _bf(torch.randn(M, N, generator=g)).to(dev)
    assert ops.get_cu_budget() == 0

    def run():
        _, f, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=False, out_f32=True)
        b16, _, _ = ops.gemm_nt(a, w, bias)
        z, _, zz = ops.gemm_nt(a, w3, None, act=ops.ACT_QUICK_GELU, aux_out=True)
        db = torch.empty(N, device=dev)
        dw = ops.gemm_tn(dy, a, db=db)
        return f, b16, z, zz, dw, db

    ref = run()
    try:
        for budget in (240, 200, 256):
            ops.set_cu_budget(budget)
            assert ops.get_cu_budget() == budget
            out = run()
            for x, y in zip(out[:4], ref[:4]):
                assert torch.equal(x, y)
            for x, y in zip(out[4:], ref[4:]):   # the weight gradient's M-split (hence its fp32 summation order) follows the plan
                assert _rel(x, y) < 1e-5
    finally:
        ops.set_cu_budget(0)
    with pytest.raises(Exception):
        ops.set_cu_budget(300)


@pytest.mark.parametrize("M,N,K", [(32800, 768, 768), (32800, 768, 3072), (512, 768, 128), (300, 768, 192), (2048 + 64, 384, 256),
                                   (1024, 768, 64)])
@pytest.mark.parametrize("with_resid", [True, False])
def test_gemm_nt_ln_fused_matches_the_two_launches(dev, M, N, K, with_resid):
    """The GEMM + LayerNorm launch (256x384 tiles whose two column tiles exchange row statistics; round 5): the fp32 output is
    BITWISE what gemm_nt writes, mean / rstd agree with the LayerNorm kernel to fp32 rounding (another summation order of the same
    two-pass variance), the bf16 rows agree with it to one bf16 ulp, and against fp64 LayerNorm of that output they meet the
    LayerNorm test's bound.  Run three times on the same exchange buffer (it must come back all zero), with the ragged
    32 / 44 / 64 rows of B x 1025-token inputs and N = 384 (one column tile: no exchange)."""
    from lc2is_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 13 + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = (torch.randn(M, N, generator=g) * 3.0 + 0.7).to(dev) if with_resid else None
    if resid is not None:
        resid[:, 5] += 40.0          # an outlier channel, as CLIP's residual stream has
    gamma = (torch.rand(N, generator=g) + 0.5).to(dev)
    beta = torch.randn(N, generator=g).to(dev)
    assert ops.gemm_nt_ln_ok(32800, 768, K) and not ops.gemm_nt_ln_ok(8200, 768, K)   # (the modules fuse only where the tiles fill the chip)
    _, x_ref, _ = ops.gemm_nt(a, w, bias, resid=resid, out_bf16=None, out_f32=True)
    h_ref, _, m_ref, r_ref = ops.layernorm_fwd(x_ref, gamma, beta, 1e-5)
    for rep in range(3):
        x, h, mean, rstd = ops.gemm_nt_ln(a, w, bias, resid, gamma, beta, 1e-5)
        torch.cuda.synchronize()
        assert torch.equal(x, x_ref), rep
        assert torch.isfinite(mean).all() and torch.isfinite(rstd).all()
        assert (mean - m_ref).abs().max().item() <= 2e-6 * m_ref.abs().max().item() + 1e-6
        assert ((rstd - r_ref).abs() / r_ref).max().item() < 5e-6
        d = (h.float() - h_ref.float()).abs()
        assert (d <= h_ref.float().abs() * 2.0 ** -7 + 1e-6).all(), rep                      # at most one bf16 ulp apart
        assert (d > 0).float().mean().item() < 2e-2, rep                                     # ... and only where a rounding boundary sits
        xd = x.double()
        ln = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * gamma.double() + beta.double()
        assert _rel(h.float(), ln) < 4e-3
    for buf in ops._xchg_cache.values():
        assert int(buf.view(torch.int64).ne(0).sum().item()) == 0   # every granule consumed and cleared


def test_gemm_nt_ln_refuses_what_it_does_not_take(dev):
    from lc2is_amd import ops
    a = torch.zeros(400, 64, dtype=torch.bfloat16, device=dev)
    gamma = torch.ones(768, device=dev)
    with pytest.raises(RuntimeError):   # 144 ragged rows: more than the fragment jobs take
        ops.gemm_nt_ln(a, torch.zeros(768, 64, dtype=torch.bfloat16, device=dev), None, None, gamma, None)
    with pytest.raises(RuntimeError):   # N = 512
        ops.gemm_nt_ln(a[:256], torch.zeros(512, 64, dtype=torch.bfloat16, device=dev), None, None, gamma[:512], None)


def test_vision_tower_with_fused_layernorm_launches_matches_the_separate_ones(dev, monkeypatch):
    """The ViT tower's forward AND backward with out-proj / fc2 writing the LayerNorm that follows them (ops.gemm_nt_ln), against the same
    tower with the LayerNorm kernel: the modules take the fused launch only from 224 tiles on (B >= 28 at 512 x 512), so the row
    threshold is lowered here — 3 layers, B = 2 (M = 2050: 8 row tiles + 2 ragged rows).  The statistics differ in the last fp32 bits
    and a bf16 row element by at most one ulp; tokens and gradients agree far inside the bf16 noise of either form."""
    import lc2is_amd.nn as N
    from lc2is_amd import ops
    torch.manual_seed(7)
    base = N.ImageEncoderCLIP(512, 16, arch=N.ClipArch(768, 12, 3, 3072))
    sd = {k: v.detach().clone() for k, v in base.state_dict().items()}
    pix = torch.randn(2, 3, 512, 512, generator=torch.Generator().manual_seed(3)).to(dev)
    gout = None
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, "_LN_FUSE", fused)
        monkeypatch.setattr(ops, "_LN_FUSE_MIN_ROWS", 256)
        assert ops.gemm_nt_ln_ok(2050, 768, 768) == fused
        enc = N.ImageEncoderCLIP(512, 16, arch=N.ClipArch(768, 12, 3, 3072))   # (a fresh module: gradient buffers accumulate)
        enc.load_state_dict(sd)
        enc = enc.to(dev).train()
        out = enc(pixel_values=pix)
        out = out if torch.is_tensor(out) else out[0]
        if gout is None:
            gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).to(dev)
        out.backward(gout)
        res[fused] = (out.detach().clone(), {k: v.grad.detach().clone() for k, v in enc.named_parameters() if v.grad is not None})
    o0, g0 = res[False]
    o1, g1 = res[True]
    assert _rel(o1, o0) < 1e-3, _rel(o1, o0)
    # (k_proj.bias: its true gradient is zero — softmax ignores a per-query constant — so both forms hold rounding noise only)
    rels = {k: _rel(g1[k], g0[k]) for k in g0 if not k.endswith("k_proj.bias")}
    worst = max(rels.values())
    assert set(g0) == set(g1) and worst < 1e-2, sorted(rels.items(), key=lambda kv: -kv[1])[:6]   # measured 4.8e-3 (layer_norm1.weight of layer 0)
