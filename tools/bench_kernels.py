"""Micro-benchmarks of single kernels on the GPU box (HIP-event timed). Usage: python tools/bench_kernels.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from lc2is_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_attention(dev):
    for (B, H, S, D) in [(32, 12, 1025, 64), (32, 8, 1024, 96)]:
        qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
        o = torch.empty(B * S, H * D, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: ops.attention_fwd(q, k, v, B, H, S, S, D, D ** -0.5, out=o))
        print(f"attention_fwd B={B} H={H} S={S} D={D}: {t*1e6:8.1f} us  {4*B*H*S*S*D/t/1e12:7.1f} TF/s", flush=True)
        _, lse = ops.attention_fwd(q, k, v, B, H, S, S, D, D ** -0.5, out=o)
        do = torch.randn_like(o)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :H * D], dqkv[:, H * D:2 * H * D], dqkv[:, 2 * H * D:]
        t = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, B, H, S, S, D, D ** -0.5, dq=dq, dk=dk, dv=dv))
        print(f"attention_bwd B={B} H={H} S={S} D={D}: {t*1e6:8.1f} us  {10*B*H*S*S*D/t/1e12:7.1f} TF/s (5-matmul count)", flush=True)


def main():
    dev = torch.device("cuda:0")
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        return bench_attention(dev)
    M = 32 * 1025
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for cfg in (4, 13):
            t = timeit(lambda: ops.gemm_nt(a, w, bias, out_bf16=out, tile_cfg=cfg))
            print(f"gemm_nt cfg{cfg} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        dy = torch.randn(M, N, device=dev).bfloat16()
        dw = torch.empty(N, K, device=dev)
        t = timeit(lambda: ops.gemm_tn(dy, a, dw))
        print(f"gemm_tn      M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        dbv = torch.empty(N, device=dev)
        t = timeit(lambda: ops.gemm_tn(dy, a, dw, db=dbv))
        print(f"gemm_tn+db   M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        t = timeit(lambda: ops.colsum(dy))
        print(f"colsum       M={M} N={N}: {t*1e6:8.1f} us  {M*N*2/t/1e9:7.1f} GB/s", flush=True)
    x = torch.randn(M, 768, device=dev)
    g = torch.ones(768, device=dev)
    b = torch.zeros(768, device=dev)
    y = torch.empty(M, 768, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, out_bf16=y))
    print(f"layernorm_fwd M={M} C=768: {t*1e6:8.1f} us  {M*768*6/t/1e9:7.1f} GB/s", flush=True)
    _, _, mean, rstd = ops.layernorm_fwd(x, g, b, out_bf16=y)
    dy = torch.randn(M, 768, device=dev).bfloat16()
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=x))
    print(f"layernorm_bwd M={M} C=768: {t*1e6:8.1f} us  {M*768*(2+4+4+4+2)/t/1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
