"""40 training steps of the headline configuration: allocated / peak / reserved HBM must be flat (no per-step growth) and the
loss must fall.  Run on the GPU box: python tools/memcheck.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, lc2is_amd.nn as N
from lc2is_amd.step import TrainStep
from bench import synth_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = N.BaseModelWithText(16, 512, 128).to(dev).train()
ts = TrainStep(m, optimizer='sgd', lr=1e-5)
inputs, labels = synth_batch(32, 512, 128, 16, 2, dev)
for i in range(41):
    loss = ts.step(inputs, labels)
    if i in (4, 20, 40):
        torch.cuda.synchronize()
        print(i, 'alloc GB', round(torch.cuda.memory_allocated()/2**30, 2), 'max', round(torch.cuda.max_memory_allocated()/2**30, 2), 'reserved', round(torch.cuda.memory_reserved()/2**30, 2), 'loss', float(loss))
