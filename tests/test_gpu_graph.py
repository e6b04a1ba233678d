"""hipGraph replay of the train step (lc2is_amd/step.py ``TrainStep.capture``): the captured step — text-tower fork / join, the
tower-wide weight-gradient grid with its device-side descriptor table, fused SGD — must leave the same parameters as the
eager step (reference: one ``Engine.train_loop`` iteration, engine.py:84-104)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _model(dev, dropout=0.0):
    import lc2is_amd.nn as N
    torch.manual_seed(7)
    # 4 vision layers = 24 weight-gradient problems: more than the 16 that travel as kernel arguments, so the grouped launch
    # takes the descriptor-table path whose upload must survive capture
    m = N.BaseModelWithText(16, 64, 16, vision_arch=N.ClipArch(128, 2, 4, 256),
                            text_arch=N.ClipArch(64, 1, 2, 128, vocab=512, eos_token_id=511), nhead=2,
                            dim_feedforward=128, out_dim=64, **({"dropout": dropout} if dropout else {}))
    return m.to(dev).train()


def _batch(dev, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 500, (2, 8), generator=g)
    ids[:, 0], ids[:, -1] = 510, 511
    return ({"pixel_values": torch.randn(2, 3, 64, 64, generator=g).to(dev), "input_ids": ids.to(dev),
             "attention_mask": torch.ones(2, 8, dtype=torch.long).to(dev)},
            torch.randint(0, 151, (2, 16, 16), generator=g).to(dev))


def test_graph_replay_matches_eager(dev):
    """Eager and replayed steps from the same parameters on the same batches.  The two are the same arithmetic except for the
    weight-gradient plan (whole tower in one grid eagerly, per-layer groups under capture: another split-K count, fp32 sums in
    another order), so losses agree to fp32 rounding of an O(10) loss and the parameter UPDATES to 1e-4 of their size.
    The learning rate keeps the random-init model in the regime training runs in (round 4 trained it at lr = 0.05 into a loss
    of 2e4 — a diverged trajectory amplifies last-bit differences and says nothing about the capture)."""
    from lc2is_amd.step import TrainStep
    batches = [_batch(dev, s) for s in range(4)]
    m_e, m_g = _model(dev), _model(dev)
    m_g.load_state_dict(m_e.state_dict())
    lr = 1e-3
    ts_e, ts_g = TrainStep(m_e, optimizer="sgd", lr=lr), TrainStep(m_g, optimizer="sgd", lr=lr)
    start = ts_e.arena.flat.clone()
    # capture() runs 2 real warm-up steps on the first batch, then records (does not run) the captured one
    for _ in range(2):
        ts_e.step(*batches[0])
    run = ts_g.capture(*batches[0])
    torch.cuda.synchronize()
    d0 = (ts_e.arena.flat - ts_g.arena.flat).abs().max().item()
    assert d0 < 1e-6, d0
    losses_e, losses_g = [], []
    for inp, lab in batches[1:]:
        losses_e.append(ts_e.step(inp, lab).item())
        losses_g.append(run(inp, lab).item())
    torch.cuda.synchronize()
    assert max(losses_e) < 20.0, losses_e                       # still a sane cross-entropy over 151 classes
    assert losses_e == pytest.approx(losses_g, abs=1e-5), (losses_e, losses_g)
    d = (ts_e.arena.flat - ts_g.arena.flat).abs().max().item()
    assert d < 1e-6, d
    upd_e, upd_g = ts_e.arena.flat - start, ts_g.arena.flat - start
    assert upd_e.norm().item() > 0
    assert ((upd_e - upd_g).norm() / upd_e.norm()).item() < 1e-4
    assert ts_g.t == ts_e.t


def test_capture_refuses_active_dropout(dev):
    from lc2is_amd.step import TrainStep
    m = _model(dev, dropout=0.1)
    ts = TrainStep(m, optimizer="sgd", lr=0.05)
    with pytest.raises(RuntimeError, match="dropout"):
        ts.capture(*_batch(dev, 0))


def test_grouped_table_launch_survives_capture(dev):
    """> 16 problems: the descriptor table is uploaded by a captured memcpy node that re-reads its host image at every replay;
    the image belongs to the captured call alone, so eager launches between replays (which cycle the pinned ring) cannot
    change what the graph uploads."""
    from lc2is_amd import ops
    torch.manual_seed(3)
    n, M, N, K = 20, 512, 64, 128
    dys = [torch.randn(M, N, device=dev).bfloat16() for _ in range(n)]
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(n)]
    dws = [torch.zeros(N, K, device=dev) for _ in range(n)]
    dbs = [torch.zeros(N, device=dev) for _ in range(n)]
    probs = [(dy, x, dw, db, False) for dy, x, dw, db in zip(dys, xs, dws, dbs)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm_tn_grouped(probs)      # warm-up: workspace, attributes
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ops.gemm_tn_grouped(probs)
    # eager table launches on OTHER buffers cycle all four ring slots
    other = [(torch.randn(M, N, device=dev).bfloat16(), torch.randn(M, K, device=dev).bfloat16(),
              torch.zeros(N, K, device=dev), None, False) for _ in range(n)]
    for _ in range(5):
        ops.gemm_tn_grouped(other)
    for rep in range(2):
        for dy, x in zip(dys, xs):
            dy.copy_(torch.randn(M, N, device=dev)); x.copy_(torch.randn(M, K, device=dev))
        g.replay()
        torch.cuda.synchronize()
        for dy, x, dw, db in zip(dys, xs, dws, dbs):
            ref = dy.float().t() @ x.float()
            assert ((dw - ref).norm() / ref.norm()).item() < 2e-3
            assert ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item() < 2e-3
