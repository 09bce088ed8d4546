"""bench.py's loader-fed leg alone (WAV files -> workers -> prefetch stream -> training step), for A/B timing of host / stream
settings (e.g. DS2_PREFETCH_PRIORITY=low)."""
import os, sys, json
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'aes-lac-2018_amd')); sys.path.insert(0, _ROOT)
import torch
import bench
from codes.engine import Trainer
from codes.model import DeepSpeech
dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = DeepSpeech().to(dev)
opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
trainer = Trainer(model, opt, device=dev, max_norm=400)
plan = bench.bin_plan(10, bench.NUM_BINS, world=1)
print(json.dumps(bench.loader_leg(trainer, plan, dev)))
