"""Persistent 256x256 GEMM (cfg 11) with staggered block starts (LC2IS_GEMM_STAGGER = 10-ns ticks per phase, 8 phases) against the
plain kernel (cfg 4) at the vision tower's shapes.  usage: LC2IS_GEMM_STAGGER=<ticks> python tools/gemm_stagger.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit

dev = torch.device("cuda:0")
M = 32768
for name, (N, K) in dict(qkv=(2304, 768), proj=(768, 768), fc1=(3072, 768), fc2=(768, 3072), dqkv=(768, 2304)).items():
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = {}
    for rnd in range(3):
        for cfg in (4, 11):
            t = timeit(lambda: ops.gemm_nt(a, w, None, out_bf16=out, tile_cfg=cfg), iters=10, warm=2)
            res.setdefault(cfg, []).append(t)
    print(f"stagger={os.environ.get('LC2IS_GEMM_STAGGER','0'):>5} {name:5s} " + "  ".join(f"cfg{c}={min(v)*1e6:6.1f}us({2*M*N*K/min(v)/1e12:5.0f}TF)" for c, v in res.items()), flush=True)
