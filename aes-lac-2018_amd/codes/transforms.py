"""Log-spectrogram frontend on the GPU (reference ``codes/transforms.py:26-127``, ``ToSpectrogram``).

The reference runs ``librosa.stft`` per utterance on CPU DataLoader workers (``codes/data.py:61-62``,
``codes/transforms.py:94-119``).  Here the same arithmetic -- centre reflect-pad 160, 320-sample frames
every 160, symmetric Hann (``librosa_compat=True`` forces ``periodic=False``, ``:52-53``), |rFFT|, log1p,
per-utterance (S - mean) / (std_unbiased + eps) -- is one HIP kernel pair (csrc/spectrogram.hip) that
processes a whole padded minibatch after collate: ``BatchSpectrogram``.  ``ToSpectrogram`` keeps the
reference's constructor and per-utterance ``__call__`` for drop-in use.
"""
import torch

from ds2hip import ops

FRAME, HOP, NBINS = 320, 160, 161


class ToSpectrogram(object):
    def __init__(self, frame_length=320, hop=160, fft_size=None, pad_end=0, normalize=True,
                 window=torch.hann_window, window_params=None, librosa_compat=False, eps=1e-9, device='cuda'):
        fft_size = fft_size or frame_length
        hop = hop if hop is not None else frame_length // 2
        window_params = dict(window_params or {})
        if librosa_compat:
            window_params.setdefault('periodic', False)
        ok = (frame_length == FRAME and hop == HOP and fft_size == FRAME and pad_end == 0 and librosa_compat and
              window is torch.hann_window and window_params == {'periodic': False})
        if not ok:
            raise NotImplementedError('the HIP frontend implements the configuration the reference trains with: '
                                      'frame 320, hop 160, symmetric Hann, librosa_compat=True '
                                      '(codes/utils/training_utils.py:19-23)')
        self.frame_length, self.hop, self.fft_size = frame_length, hop, fft_size
        self.normalize, self.pad_end, self.eps = normalize, pad_end, eps
        self.window_params, self.librosa_compat = window_params, librosa_compat
        self.device = device

    def __call__(self, x):
        """x: 1-D float tensor of samples -> (T_in, 161) on x's device (a CPU input makes a GPU round trip)."""
        assert x.dim() == 1 and isinstance(x, torch.Tensor)
        src = x.device
        wav = x.to(self.device, torch.float32).contiguous()
        assert wav.numel() > FRAME // 2, 'reflect padding needs more than 160 samples'
        offs = torch.tensor([0, wav.numel()], dtype=torch.int64, device=wav.device)
        out = ops.spectrogram(wav, offs, 1 + wav.numel() // HOP, self.normalize, self.eps)[0]
        return out.to(src)

    def __repr__(self):
        return ('{}(frame_length={}, hop={}, fft_size={}, pad_end={}, normalize={},librosa_compat={})').format(
            self.__class__.__name__, self.frame_length, self.hop, self.fft_size, self.pad_end, self.normalize,
            self.librosa_compat)


class BatchSpectrogram(object):
    """Frontend + collate for a minibatch of raw clips, entirely on the device.

    ``__call__(wavs)`` with ``wavs`` a list of 1-D tensors (or (flat, offsets)) returns
    ``inputs (B,T_max,161)`` and ``input_percentages (B)`` float32 exactly as
    ``AudioDataLoader._collate_fn`` would have (``codes/data.py:132-152``): zero padding past each
    clip's frames, percentage = T_i / float(T_max) stored as float32.
    """

    def __init__(self, normalize=True, eps=1e-9, device='cuda'):
        self.normalize, self.eps, self.device = normalize, eps, device

    def __call__(self, wavs, offsets=None):
        if offsets is None:
            lens = [int(w.numel()) for w in wavs]
            flat = torch.cat([w.reshape(-1).to(self.device, torch.float32) for w in wavs])
        else:
            flat = wavs.to(self.device, torch.float32).contiguous()
            lens = [int(offsets[i + 1] - offsets[i]) for i in range(len(offsets) - 1)]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        frames = [1 + n // HOP for n in lens]
        t_max = max(frames)
        offs_d = ops.upload_small(torch.tensor(offs, dtype=torch.int64), flat.device)
        inputs = ops.spectrogram(flat, offs_d, t_max, self.normalize, self.eps)
        pct = torch.tensor([f / float(t_max) for f in frames], dtype=torch.float32)
        return inputs, pct


# ---------------------------------------------------------------------------------------------------------
# Host-side neighbours of the hot path: waveform loading and transcript -> labels.
# ---------------------------------------------------------------------------------------------------------
class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def wsola_tempo(x, tempo, sample_rate=16000, segment_ms=82.0, search_ms=14.68, overlap_ms=12.0):
    """Change the tempo of a 1-D float array without changing its pitch (WSOLA: waveform-similarity overlap-add).

    Stands in for ``sox ... tempo <factor>`` (``codes/transforms.py:185-218``), whose defaults are a 82 ms segment,
    a 14.68 ms search window and a 12 ms overlap.  Output length ~ len(x) / tempo.  The segment that continues the
    output is looked for around the ideal input position, where it correlates best with the natural continuation of
    the previous segment, and cross-faded in over the overlap.  Not bit-compatible with sox (nor claimed to be)."""
    import numpy as np
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    if abs(tempo - 1.0) < 1e-6 or x.size == 0:
        return x.copy()
    seg = max(int(sample_rate * segment_ms / 1000.0), 4)
    ovl = max(min(int(sample_rate * overlap_ms / 1000.0), seg // 2), 1)
    half = max(int(sample_rate * search_ms / 1000.0) // 2, 1)
    hop_out = seg - ovl
    hop_in = tempo * hop_out
    if x.size <= seg + half:
        return x.copy()
    fade_in = np.linspace(0.0, 1.0, ovl, endpoint=False, dtype=np.float32)
    out = np.zeros(int(x.size / tempo) + 2 * seg, dtype=np.float32)
    out[:seg] = x[:seg]
    out_pos, prev, ideal = hop_out, 0, 0.0
    while True:
        ideal += hop_in
        base = int(round(ideal))
        lo, hi = max(base - half, 0), min(base + half, x.size - seg)
        if hi < lo:
            break
        want = x[prev + hop_out: prev + hop_out + ovl]                # how the previous segment would have gone on
        if want.size < ovl:
            break
        window = x[lo: hi + ovl]
        corr = np.correlate(window, want, mode='valid')                # corr[d] = <x[lo + d : lo + d + ovl], want>
        start = lo + int(np.argmax(corr))
        out[out_pos: out_pos + ovl] = out[out_pos: out_pos + ovl] * (1.0 - fade_in) + x[start: start + ovl] * fade_in
        out[out_pos + ovl: out_pos + seg] = x[start + ovl: start + seg]
        prev, out_pos = start, out_pos + hop_out
    return out[: out_pos + ovl]


class ToTensor(object):
    """16-bit PCM mono WAV -> 1-D float tensor (reference ``codes/transforms.py:130-224``).

    The reference decodes with torchaudio/sox (absent here); this loader reads PCM16 with the standard library and
    scales to [-1, 1).  ``augment=True`` draws a tempo and a gain uniformly from the ranges exactly as the reference
    does (``np.random.uniform``) and applies them on the host: tempo with ``wsola_tempo`` above, gain in dB, then
    the clip and 16-bit rounding sox's ``-b 16`` output implies.  sox itself is not available, so the augmented
    waveform is the same kind of signal, not the same samples."""

    def __init__(self, sample_rate=16000, augment=False, tempo_range=(0.85, 1.15), gain_range=(-6, 8)):
        self.sample_rate, self.augment = sample_rate, augment
        self.tempo_range, self.gain_range = tempo_range, gain_range

    def _load(self, path):
        import wave

        import numpy as np
        with wave.open(path, 'rb') as w:
            assert w.getframerate() == self.sample_rate, 'sample rate mismatch'
            assert w.getsampwidth() == 2 and w.getnchannels() == 1, 'expected 16-bit mono PCM'
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
        return pcm.astype('float32') / 32768.0

    def __call__(self, path):
        import numpy as np
        y = self._load(path)
        if self.augment:
            tempo = np.random.uniform(low=self.tempo_range[0], high=self.tempo_range[1])
            gain = np.random.uniform(low=self.gain_range[0], high=self.gain_range[1])
            y = wsola_tempo(y, float('{:.3f}'.format(tempo)), self.sample_rate)      # sox got three decimals
            y = y * np.float32(10.0 ** (float('{:.3f}'.format(gain)) / 20.0))
            y = np.clip(np.round(y * 32768.0), -32768, 32767).astype('float32') / 32768.0
        return torch.from_numpy(np.ascontiguousarray(y))

    def __repr__(self):
        return '{}(sample_rate={}, augment={}, tempo_range={}, gain_range={})'.format(
            self.__class__.__name__, self.sample_rate, self.augment, self.tempo_range, self.gain_range)


_ACCENT_FOLD = {'À': 'A', 'Á': 'A', 'Â': 'A', 'Ã': 'A', 'Ä': 'A', 'Ç': 'C', 'È': 'E', 'É': 'E', 'Ê': 'E', 'Ë': 'E',
                'Ì': 'I', 'Í': 'I', 'Î': 'I', 'Ï': 'I', 'Ñ': 'N', 'Ò': 'O', 'Ó': 'O', 'Ô': 'O', 'Õ': 'O', 'Ö': 'O',
                'Ù': 'U', 'Ú': 'U', 'Û': 'U', 'Ü': 'U'}


class ToLabel(object):
    """Transcript (string or path) -> (L,1) int array of label ids (reference ``codes/transforms.py:295-392``).

    Upper-cases, optionally folds accents (a fixed Latin-1 table stands in for ``unidecode``), drops every
    character that is not in the label set.  Number-to-words conversion (``num2words``, absent here) is NOT
    applied.  In the reference the conversion runs AFTER upper-casing (``codes/transforms.py:370-376``) and
    num2words writes lower-case words, so with the upper-case alphabets of ``data/labels.*.json`` every letter
    of the spelled-out number is filtered again and only the spaces between its words survive: "I HAVE 2 DOGS"
    becomes "I HAVE  DOGS" there and here alike; a multi-word number ("1,234") leaves a few more spaces in the
    reference than here -- a documented deviation (LibriSpeech transcripts contain no digits)."""

    def __init__(self, labels='labels.en.json', to_upper=True, one_hot=False, convert_number_to_words=True, lang=None,
                 remove_accents=True, dtype=None):
        import os

        import numpy as np

        from .preprocessing import OrderedLabelEncoder
        from .utils.io_utils import read_labels
        if one_hot:
            raise NotImplementedError('one-hot targets are not used by the CTC path')
        if isinstance(labels, str) and os.path.isfile(labels):
            labels_list = read_labels(labels)
            lang = lang or labels.split('.')[-2]
        else:
            labels_list = list(labels)
        self._labels, self._lang, self._to_upper = labels, lang, to_upper
        self._remove_accents = remove_accents
        self._dtype = dtype or np.int64
        self.label_encoder = OrderedLabelEncoder().fit(labels_list)
        self._known = set(self.label_encoder.classes_.tolist())

    def __call__(self, x):
        import os

        import numpy as np
        if isinstance(x, bytes):
            x = x.decode('utf8')
        if os.path.isfile(x):
            with open(x, 'r', encoding='utf8') as f:
                x = f.readline().strip()
        if self._to_upper:
            x = x.upper()
        if self._remove_accents:
            x = ''.join(_ACCENT_FOLD.get(c, c) for c in x)
        chars = [c for c in x if c in self._known]
        ids = np.asarray(self.label_encoder.transform(chars), dtype=self._dtype).reshape(-1)
        return ids[:, np.newaxis]
