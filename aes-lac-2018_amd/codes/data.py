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


class AudioDataset(torch.utils.data.Dataset):
    """Manifest CSV ``audio_path,transcript_path,duration`` (reference ``codes/data.py:13-70``; no zip support)."""

    def __init__(self, data_dir, manifest_filepath, transforms=None, target_transforms=None):
        import os
        self.data_dir, self.manifest_filepath = data_dir, manifest_filepath
        with open(manifest_filepath) as f:
            rows = [line.strip().split(',') for line in f if line.strip()]
        self.durations = [float(r[2]) for r in rows]
        self.data = [(os.path.join(data_dir, r[0]), os.path.join(data_dir, r[1])) for r in rows]
        self.transforms, self.target_transforms = transforms, target_transforms

    def __getitem__(self, index):
        audio, target = self.data[index]
        if self.transforms is not None:
            audio = self.transforms(audio)
        if self.target_transforms is not None:
            target = self.target_transforms(target)
        return audio, target

    def __len__(self):
        return len(self.data)


class AudioDataLoader(torch.utils.data.DataLoader):
    """DataLoader whose collate is the reference's (``codes/data.py:96-166``).

    ``raw_audio=True`` keeps clips as raw 1-D waveforms (``collate_audio``) so the spectrogram runs on the GPU
    after collate (``BatchSpectrogram``) instead of per utterance in worker processes."""

    def __init__(self, *args, **kwargs):
        raw = kwargs.pop('raw_audio', False)
        kwargs.pop('num_tasks', None)
        kwargs['collate_fn'] = _collate_raw if raw else _collate_spect
        super().__init__(*args, **kwargs)


def _labels_list(t):
    import numpy as np
    return [int(v) for v in np.asarray(t).reshape(-1)]


def _collate_spect(batch):
    return collate([(s, _labels_list(t)) for s, t in batch])


def _collate_raw(batch):
    wavs, targets, sizes = collate_audio([(w, _labels_list(t)) for w, t in batch])
    return wavs, targets, None, sizes
