"""Oracle: log-magnitude spectrogram frontend (TEST INFRASTRUCTURE ONLY).

Restates ``ToSpectrogram.__call__`` with ``librosa_compat=True`` -- the branch
the reference selects in ``codes/utils/training_utils.py:19-23`` -- following
``codes/transforms.py:94-119``:

  S = |librosa.stft(x, n_fft=320, hop_length=160, win_length=320,
                    window=hann(320, periodic=False))|.T      (:96-102)
  S = log1p(S)                                                 (:114)
  S = (S - S.mean()) / (S.std() + eps)   (torch unbiased std)  (:116-117)

librosa itself is absent from the reference tree and un-pinned
(docker/requirements.txt:4); its published STFT algorithm is: centre the
signal by reflect-padding n_fft//2 samples each side, cut frames of n_fft
samples every hop, multiply by the window, real FFT.  Number of frames is
1 + L // hop.  Parity with librosa itself is "unpinned"; the restatement is
cross-checked against torch.stft(center=True, pad_mode='reflect') in
tests/test_oracle_misc.py.
"""
import numpy as np

FRAME = 320
HOP = 160
NBINS = FRAME // 2 + 1


def hann_symmetric(n=FRAME, dtype=np.float64):
    """torch.hann_window(n, periodic=False): 0.5 - 0.5 cos(2 pi k / (n-1))."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))).astype(dtype)


def num_frames(num_samples, hop=HOP):
    return 1 + num_samples // hop


def stft_magnitude(x, frame=FRAME, hop=HOP, dtype=np.float64):
    """|STFT| with centre reflect padding; returns (T_in, frame//2+1)."""
    x = np.asarray(x, dtype=dtype)
    assert x.ndim == 1 and x.shape[0] > frame // 2, "reflect pad needs L > n_fft/2"
    xp = np.pad(x, frame // 2, mode='reflect')
    n = num_frames(x.shape[0], hop)
    idx = np.arange(frame)[None, :] + hop * np.arange(n)[:, None]
    frames = xp[idx] * hann_symmetric(frame, dtype)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    return np.abs(spec).astype(dtype)


def log_spectrogram(x, normalize=True, eps=1e-9, dtype=np.float32):
    """Full frontend for one utterance: (L,) float -> (T_in, 161) float32.

    Statistics are taken in float64 over all T_in*161 elements; the std is the
    unbiased one (torch.Tensor.std default), as at codes/transforms.py:117.
    """
    s = np.log1p(stft_magnitude(x, dtype=np.float64))
    if normalize:
        mean = s.mean()
        std = s.std(ddof=1)
        s = (s - mean) / (std + eps)
    return s.astype(dtype)


def batch_log_spectrogram(wavs):
    """Frontend + collate for a list of 1-D clips -> (inputs, input_percentages).

    Mirrors what the reference produces by running ToSpectrogram per utterance in
    AudioDataset.__getitem__ (codes/data.py:61-62) and zero-padding in
    AudioDataLoader._collate_fn (codes/data.py:132-152).
    """
    specs = [log_spectrogram(w) for w in wavs]
    tmax = max(s.shape[0] for s in specs)
    out = np.zeros((len(specs), tmax, NBINS), dtype=np.float32)
    pct = np.zeros((len(specs),), dtype=np.float32)
    for i, s in enumerate(specs):
        out[i, :s.shape[0]] = s
        pct[i] = np.float32(s.shape[0] / float(tmax))
    return out, pct
