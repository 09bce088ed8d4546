"""Minibatch assembly (reference ``codes/data.py:106-166``, ``AudioDataLoader._collate_fn``).

Single-task collate only (the multi-task branch is out of scope).  The output
layout is the hot path's input contract: ``inputs (B,T_max,161)`` float32 zero padded,
``targets`` flat int32, ``input_percentages (B)`` float32 = T_i / float(T_max), ``target_sizes (B)`` int32.
"""
import torch


def collate(batch):
    """batch: list of (spect (T_i,F) tensor, labels list[int]) -> the reference's 4-tuple (CPU tensors)."""
    longest = max(batch, key=lambda s: s[0].shape[0])[0]
    t_max, nfreq = longest.shape
    n = len(batch)
    inputs = torch.zeros(n, t_max, nfreq)
    pct = torch.zeros(n, dtype=torch.float)
    sizes = torch.zeros(n, dtype=torch.int)
    flat = []
    for i, (spect, target) in enumerate(batch):
        t_i = spect.shape[0]
        inputs[i, :t_i, :].copy_(spect)
        pct[i] = t_i / float(t_max)
        sizes[i] = len(target)
        flat.extend(target)
    return inputs, torch.tensor(flat, dtype=torch.int), pct, sizes


def collate_audio(batch):
    """batch: list of (wav 1-D tensor, labels) -> (wavs list, targets, target_sizes) for the GPU frontend."""
    wavs = [b[0] for b in batch]
    flat = [int(v) for b in batch for v in b[1]]
    sizes = torch.tensor([len(b[1]) for b in batch], dtype=torch.int)
    return wavs, torch.tensor(flat, dtype=torch.int), sizes
