"""Per-FLOP efficiency of the 256x384 tile against 256x256 where BOTH fill whole rounds and the K loop dominates
(N = 3072: 6 rounds of 256x256 = 4 rounds of 256x384; fp32 output, no residual): does 22 % fewer LDS fragment bytes per FLOP
buy time in the power-limited regime?  usage: python tools/probes/gemm_w384_longk.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, it=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for (M, N, K) in [(32768, 3072, 3072), (32768, 3072, 768), (65536, 1536, 4096)]:
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.03).bfloat16()
    of = torch.empty(M, N, device=dev)
    r = []
    for rep in range(3):
        r.append((timeit(lambda: ops.gemm_nt(x, w, None, out_bf16=False, out_f32=of, tile_cfg=4)),
                  timeit(lambda: ops.gemm_nt(x, w, None, out_bf16=False, out_f32=of, tile_cfg=16))))
    m = [min(c) for c in zip(*r)]
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: 256x256 {m[0]:8.1f} us {fl / m[0] / 1e6:6.0f} TF/s | 256x384 {m[1]:8.1f} us {fl / m[1] / 1e6:6.0f} TF/s", flush=True)
