import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from lc2is_amd import ops
from bench_kernels import timeit
dev = torch.device("cuda:0")
a = torch.randn(1024, 768, device=dev).bfloat16(); w = (torch.randn(768, 768, device=dev) * 0.05).bfloat16()
b = torch.randn(768, device=dev)
r4 = ops.gemm_nt(a, w, b, tile_cfg=4)[0]; r12 = ops.gemm_nt(a, w, b, tile_cfg=12)[0]
print("cfg12 vs cfg4 maxdiff", (r4.float() - r12.float()).abs().max().item())
for (M, N, K) in [(32768, 3072, 768), (32768, 768, 3072), (32768, 768, 768), (32768, 2304, 768)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {}
    for rnd in range(3):
        for cfg in (4, 12, 41, 48):
            t = timeit(lambda: ops.gemm_nt(a, w, None, out_bf16=out, tile_cfg=cfg), iters=10, warm=2)
            res.setdefault(cfg, []).append(t)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"cfg{c}={min(v)*1e6:6.1f}us({2*M*N*K/min(v)/1e12:5.0f}TF)" for c, v in res.items()), flush=True)
