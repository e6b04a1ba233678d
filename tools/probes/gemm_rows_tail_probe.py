import sys; sys.path.insert(0,'/root/repo')
import torch
from lc2is_amd import ops
dev=torch.device('cuda:0'); g=torch.Generator(device=dev).manual_seed(0)
def timeit(fn,it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/it
big=torch.empty(256*1024*1024//4,device=dev)
for (M,N,K) in [(32,768,3072),(32,768,2304),(32,768,768),(32,3072,768)]:
    x=torch.randn(M,K,device=dev,generator=g).bfloat16(); w=(torch.randn(N,K,device=dev,generator=g)*0.03).bfloat16()
    ob=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    t17=timeit(lambda: ops.gemm_nt(x,w,None,out_bf16=ob,tile_cfg=17))
    t3=timeit(lambda: ops.gemm_nt(x,w,None,out_bf16=ob,tile_cfg=3))
    def cold():
        big.zero_(); ops.gemm_nt(x,w,None,out_bf16=ob,tile_cfg=17)
    def coldz():
        big.zero_()
    tc=timeit(cold,10)-timeit(coldz,10)
    print(f"M={M} N={N} K={K}: rows kernel {t17:.1f} us back-to-back (hot), {tc:.1f} us after a 256-MB fill (cold); 64x64-tile kernel {t3:.1f} us",flush=True)
