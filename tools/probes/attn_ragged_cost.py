"""Upper bound on what the 1025th token costs the attention kernels: the vision-tower shape (B.H = 384, D = 64) at S = 1024
(whole 128-query / 128-key blocks only) against S = 1025 (one more block per (batch, head) holding a single token).
usage: python tools/probes/attn_ragged_cost.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


B, H, D = 32, 12, 64
for S in (1024, 1025, 1088):
    g = torch.Generator(device=dev).manual_seed(1)
    C = H * D
    q = torch.randn(B * S, C, device=dev, generator=g).bfloat16()
    k = torch.randn(B * S, C, device=dev, generator=g).bfloat16()
    v = torch.randn(B * S, C, device=dev, generator=g).bfloat16()
    do = (torch.randn(B * S, C, device=dev, generator=g) * 0.5).bfloat16()
    sc = D ** -0.5
    o, lse = ops.attention_fwd(q, k, v, B, H, S, S, D, sc)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    tf = min(timeit(lambda: ops.attention_fwd(q, k, v, B, H, S, S, D, sc)) for _ in range(3))
    tb = min(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, D, sc, dq=dq, dk=dk, dv=dv)) for _ in range(3))
    print(f"S={S}: forward {tf:7.1f} us, backward (dq + dk/dv) {tb:7.1f} us", flush=True)
