import sys; sys.path.insert(0,'.')
import torch
from lc2is_amd import ops
from tools.attn_bench import timeit
dev=torch.device('cuda:0')
B,H,Sq,D=32,12,1025,64; C=H*D
g=torch.Generator(device=dev).manual_seed(1)
q=torch.randn(B*Sq,C,device=dev,generator=g).bfloat16()
for Sk in (64,128,256,512,1025,2048):
    k=torch.randn(B*Sk,C,device=dev,generator=g).bfloat16(); v=torch.randn(B*Sk,C,device=dev,generator=g).bfloat16()
    o=torch.empty(B*Sq,C,device=dev,dtype=torch.bfloat16)
    t=min(timeit(lambda: ops.attention_fwd(q,k,v,B,H,Sq,Sk,D,0.125,out=o),20) for _ in range(3))
    print(f"Sk={Sk:5d} tiles={(Sk+63)//64:3d}  {t*1e6:8.1f} us", flush=True)
