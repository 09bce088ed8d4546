import os, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/aes-lac-2018_amd')
import numpy as np, torch
import bench
from codes.model import DeepSpeech
from codes.transforms import BatchSpectrogram
from codes.decoder import GreedyDecoder
dev=torch.device('cuda')
plan=bench.bin_plan(10,24)
mine=[bench.make_bin(p) for p in plan[:12]]
res=[bench.make_resident(b,dev) for b in mine]
model=DeepSpeech().to(dev).eval(); front=BatchSpectrogram(device=dev)
dec=GreedyDecoder(['_',' ',"'"]+[chr(65+i) for i in range(26)])
with torch.no_grad():
    for rep in range(4):
        torch.cuda.synchronize(); t0=time.time(); tf=tm=td=0; fr=0
        for i in range(12):
            a=time.time(); inputs,pct=front(res[i][0],res[i][1]); b=time.time()
            probs=model(inputs); c=time.time()
            sizes=(pct*probs.shape[1]).int(); dec.decode(probs,sizes); d=time.time()
            tf+=b-a; tm+=c-b; td+=d-c; fr+=bench.frames_of(mine[i])
        torch.cuda.synchronize(); dt=time.time()-t0
        print('rep',rep,'frames/s %.0f'%(fr/dt),'host ms: front %.1f model %.1f decode %.1f total %.1f'%(tf*1e3,tm*1e3,td*1e3,dt*1e3))
