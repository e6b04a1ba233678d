"""Compare GEMM NT tile configurations on the shapes / epilogues of the ViT-B/16 step (bitwise check vs cfg 4)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops
from bench_kernels import timeit

cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 9]
dev = torch.device("cuda:0")
import os
M = int(os.environ.get("GEMM_M", "32800"))
cases = [("qkv", 2304, 768, "plain"), ("proj", 768, 768, "resid"), ("fc1", 3072, 768, "act"), ("fc2", 768, 3072, "resid"),
         ("fc1g", 3072, 768, "act5"), ("dfc2", 3072, 768, "dact"), ("dfc2m", 3072, 768, "dmul"), ("dfc1", 768, 3072, "plain"),
         ("dqkv", 768, 2304, "plain")]
for name, N, K, kind in cases:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    def run(cfg, outs):
        if kind == "plain":
            ops.gemm_nt(a, w, bias, out_bf16=outs["o"], tile_cfg=cfg)
        elif kind == "resid":
            ops.gemm_nt(a, w, bias, resid=outs["r"], out_bf16=None, out_f32=outs["f"], tile_cfg=cfg)
        elif kind == "act":
            ops.gemm_nt(a, w, bias, out_bf16=outs["o"], aux_out=outs["z"], act=ops.ACT_QUICK_GELU, tile_cfg=cfg)
        elif kind == "act5":
            ops.gemm_nt(a, w, bias, out_bf16=outs["o"], aux_out=outs["z"], act=ops.ACT_QUICK_GELU_GRAD, tile_cfg=cfg)
        elif kind == "dmul":
            ops.gemm_nt(a, w, None, aux_in=outs["z"], out_bf16=outs["o"], act=ops.ACT_MUL_AUX, tile_cfg=cfg)
        else:
            ops.gemm_nt(a, w, None, aux_in=outs["z"], out_bf16=outs["o"], act=ops.ACT_DQUICK_GELU, tile_cfg=cfg)
    def mk():
        return {"o": torch.zeros(M, N, dtype=torch.bfloat16, device=dev), "f": torch.zeros(M, N, device=dev),
                "r": torch.ones(M, N, device=dev), "z": torch.full((M, N), 0.5, dtype=torch.bfloat16, device=dev)}
    ref = mk(); run(cfgs[0], ref)
    line = []
    for cfg in cfgs:
        o = mk()
        try:
            run(cfg, o)
        except RuntimeError:
            line.append(f"cfg{cfg}=   n/a")
            continue
        same = all(torch.equal(o[k], ref[k]) for k in ("o", "f") ) and (kind not in ("act", "act5") or torch.equal(o["z"], ref["z"]))
        t = min(timeit(lambda: run(cfg, o), iters=10, warm=2) for _ in range(3))
        line.append(f"cfg{cfg}={t*1e6:6.1f}us({2*M*N*K/t/1e12:5.0f}TF){'' if same else ' MISMATCH'}")
    print(f"{name:5s} N={N:4d} K={K:4d} {kind:5s}: " + "  ".join(line), flush=True)
