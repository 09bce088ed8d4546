import os, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/aes-lac-2018_amd')
import numpy as np, torch
import bench
from codes.engine import Trainer
from codes.model import DeepSpeech
from codes.transforms import BatchSpectrogram
dev=torch.device('cuda')
torch.manual_seed(42)
model=DeepSpeech().to(dev); opt=torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
trainer=Trainer(model,opt,device=dev,max_norm=400); front=BatchSpectrogram(device=dev)
if os.environ.get('PRE10'):
    plan=bench.bin_plan(10,24); res=[bench.make_resident(bench.make_bin(p),dev) for p in plan[:8]]
    for i in range(8):
        f,o,l,n=res[i]; x,pct=front(f,o); trainer.update((x,l,pct,n))
p32=bench.bin_plan(32,8,seed=43)
bins=[bench.make_bin((900+k,p[1])) for k,p in enumerate(p32)]
res=[bench.make_resident(b,dev) for b in bins]
for rep in range(3):
    ts=[]
    for i in range(8):
        f,o,l,n=res[i]; torch.cuda.synchronize(); t0=time.time(); x,pct=front(f,o); trainer.update((x,l,pct,n)); torch.cuda.synchronize(); ts.append(time.time()-t0)
    fr=sum(bench.frames_of(b) for b in bins)
    print('rep',rep,'frames/s %.0f'%(fr/sum(ts)), ' '.join('%.1f'%(t*1e3) for t in ts), 'T_in', [max(1+len(w)//160 for w in b[0]) for b in bins])
