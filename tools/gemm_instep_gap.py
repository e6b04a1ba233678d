#!/usr/bin/env python3
"""Why is fc1 (M = 32 800, N = 3072, K = 768, quick_gelu + saved z) ~190 us inside the step and ~170 us in a back-to-back loop?
Variants of the loop: the same buffers every time; buffers rotating over 12 'layers' (fresh addresses, as in the step);
an HBM-bound kernel (LayerNorm forward) between the GEMMs; both.  usage: python tools/gemm_instep_gap.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lc2is_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, N, K, L = 32 * 1025, 3072, 768, 12
    g = torch.Generator(device="cpu").manual_seed(0)
    a = [(torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev) for _ in range(L)]
    w = [(torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev) for _ in range(L)]
    bias = torch.randn(N, generator=g).to(dev)
    out = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(L)]
    z = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(L)]
    x32 = [torch.randn(M, K, generator=g).to(dev) for _ in range(2)]
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def run(rot, ln, n=120, warm=24):
        evs = []
        for it in range(warm + n):
            i = it % L if rot else 0
            if ln:
                ops.layernorm_fwd(x32[it & 1], gamma, beta, 1e-5, out_bf16=a[i], save_stats=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm_nt(a[i], w[i], bias, act=ops.ACT_QUICK_GELU, out_bf16=out[i], aux_out=z[i])
            e1.record()
            if it >= warm:
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
        return ts[len(ts) // 2], ts[0]

    for rot in (False, True):
        for ln in (False, True):
            med, mn = run(rot, ln)
            print(f"rotating buffers={int(rot)}  layernorm between={int(ln)}:  fc1 median {med:6.1f} us  min {mn:6.1f} us")


if __name__ == "__main__":
    main()
