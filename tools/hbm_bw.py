import torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from bench_kernels import timeit
from lc2is_amd import ops
dev = torch.device("cuda:0")
for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    t = min(timeit(lambda: b.copy_(a), iters=20) for _ in range(3))
    print(f"copy {mb} MB: {t*1e6:.1f} us  {2*n*4/t/1e12:.2f} TB/s (r+w)")
    t = min(timeit(lambda: a.sum(), iters=20) for _ in range(3))
    print(f"sum  {mb} MB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s (read)")
    t = min(timeit(lambda: b.fill_(1.0), iters=20) for _ in range(3))
    print(f"fill {mb} MB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s (write)")
M, C = 32800, 768
x = torch.randn(M, C, device=dev); g = torch.ones(C, device=dev); bb = torch.zeros(C, device=dev)
t = min(timeit(lambda: ops.layernorm_fwd(x, g, bb, 1e-5, save_stats=True), iters=20) for _ in range(3))
print(f"ln_fwd: {t*1e6:.1f} us  {(M*C*6)/t/1e12:.2f} TB/s")
