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
    """batch: list of (wav, labels) -> (wavs, targets, target_sizes) for the GPU frontend.  ``wav`` is a 1-D float tensor
    (then ``wavs`` is the list of them) or a ``PCMClip`` of int16 samples (then ``wavs`` is ONE ``RawAudioBatch``)."""
    from .transforms import PCMClip, RawAudioBatch
    wavs = [b[0] for b in batch]
    if wavs and isinstance(wavs[0], PCMClip):
        wavs = RawAudioBatch.from_clips(wavs)
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


class DevicePrefetcher(object):
    """Iterate a loader of ``(RawAudioBatch, targets, None, sizes)`` one minibatch AHEAD, on a side stream: while the
    GPU trains on step i, the int16 samples of step i+1 cross PCIe (from page-locked memory when the loader pins) and --
    when a ``frontend`` (``BatchSpectrogram``) is attached -- are decoded, tempo-changed, gain-scaled and turned into
    the log-spectrogram there too, so the training stream never waits for an upload or for the (sequential per clip)
    WSOLA search.  What is yielded carries a ``ready`` event (``RawAudioBatch.ready`` / ``inputs._ds2_ready``) that the
    consumer's stream waits on.  Batches that are not ``RawAudioBatch`` pass through unchanged."""

    def __init__(self, loader, device='cuda', frontend=None):
        self.loader, self.device, self.frontend = loader, torch.device(device), frontend
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    @property
    def batch_sampler(self):
        return self.loader.batch_sampler

    def _stage(self, batch):
        from .transforms import RawAudioBatch
        wavs = batch[0]
        if not isinstance(wavs, RawAudioBatch):
            return batch
        with torch.cuda.stream(self.stream):
            dev = wavs.to(self.device, non_blocking=True)
            dev._host = wavs                         # keep the page-locked source alive until the copy has run
            if self.frontend is not None:
                inputs, pct = self.frontend(dev)
                inputs._ds2_ready = torch.cuda.Event()
                inputs._ds2_ready.record(self.stream)
                inputs._ds2_src = dev
                return (inputs, batch[1], pct, batch[3])
            dev.ready = torch.cuda.Event()
            dev.ready.record(self.stream)
        return (dev,) + tuple(batch[1:])

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        for batch in it:
            cur, nxt = nxt, self._stage(batch)
            yield cur
        yield nxt


def wait_ready(inputs):
    """Make the current stream wait for a tensor a ``DevicePrefetcher`` produced on its side stream (no-op otherwise)."""
    ev = getattr(inputs, '_ds2_ready', None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
        inputs.record_stream(torch.cuda.current_stream())
        inputs._ds2_ready = None
    return inputs
