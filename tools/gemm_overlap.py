"""Does a weight-gradient GEMM (TN, compute-heavy, tiny output) overlap with an epilogue-heavy NT GEMM on a second stream?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from lc2is_amd import ops

dev = torch.device("cuda:0")
M = 32800
dy = torch.randn(M, 768, device=dev).bfloat16()          # grad wrt fc2 output
act = torch.randn(M, 3072, device=dev).bfloat16()        # saved fc2 input
z = torch.randn(M, 3072, device=dev).bfloat16()
w2t = (torch.randn(3072, 768, device=dev) * 0.05).bfloat16()
dact = torch.empty(M, 3072, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(768, 3072, device=dev); db = torch.zeros(768, device=dev)
s2 = torch.cuda.Stream()

def nt(): ops.gemm_nt(dy, w2t, None, aux_in=z, out_bf16=dact, act=ops.ACT_DQUICK_GELU)
def tn(): ops.gemm_tn(dy, act, dw, False, db)
def serial(): nt(); tn()
def overlapped():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2): tn()
    nt()
    torch.cuda.current_stream().wait_stream(s2)

def wall(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for name, fn in (("nt", nt), ("tn", tn), ("serial", serial), ("overlapped", overlapped)):
    print(f"{name:10s} {min(wall(fn) for _ in range(3))*1e6:7.1f} us", flush=True)
