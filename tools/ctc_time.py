import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'aes-lac-2018_amd')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, numpy as np
from codes.ctc import ctc_costs_and_grad
T,B,A=405,10,29
torch.manual_seed(0)
acts=torch.randn(T,B,A,device='cuda')
lens=torch.full((B,),T,dtype=torch.int32); ll=torch.full((B,),110,dtype=torch.int32)
labels=torch.randint(1,A,(int(ll.sum()),),dtype=torch.int32)
for _ in range(3): ctc_costs_and_grad(acts,labels,lens,ll)
torch.cuda.synchronize(); ts=[]
for _ in range(10):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); c,g=ctc_costs_and_grad(acts,labels,lens,ll); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print('ctc total %.3f ms (T=%d)'%(np.median(ts),T), float(c.sum()))
