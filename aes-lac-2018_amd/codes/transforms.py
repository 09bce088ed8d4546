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

    def __init__(self, normalize=True, eps=1e-9, device='cuda', scale=None):
        self.normalize, self.eps, self.device = normalize, eps, device
        self.scale = ops.amplitude_scale(scale)      # int16 clips (RawAudioBatch) come out as q * scale: see ToTensor

    def __call__(self, wavs, offsets=None):
        if isinstance(wavs, RawAudioBatch):                  # int16 clips (+ drawn augmentation): decode on the device
            if wavs.ready is not None:               # uploaded ahead of time on the prefetcher's copy stream
                torch.cuda.current_stream().wait_event(wavs.ready)
                wavs.pcm.record_stream(torch.cuda.current_stream())
            pcm = wavs.pcm if wavs.pcm.is_cuda else wavs.pcm.to(self.device, non_blocking=True)
            flat, offs = ops.decode_augment(pcm, wavs.offsets, wavs.tempos, wavs.gains_db, scale=self.scale)
            lens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
        elif offsets is None:
            lens = [int(w.numel()) for w in wavs]
            flat = torch.cat([w.reshape(-1).to(self.device, torch.float32) for w in wavs])
        else:
            flat = wavs if (wavs.is_cuda and wavs.dtype == torch.float32 and wavs.is_contiguous()) \
                else wavs.to(self.device, torch.float32).contiguous()
            offsets = offsets.tolist() if hasattr(offsets, 'tolist') else offsets     # (Python ints: numpy scalars are 10x slower)
            lens = [offsets[i + 1] - offsets[i] for i in range(len(offsets) - 1)]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        frames = [1 + n // HOP for n in lens]
        t_max = max(frames)
        # (the offsets stay on the host, in page-locked memory the two kernels read in place: ops.spectrogram)
        inputs = ops.spectrogram(flat, torch.tensor(offs, dtype=torch.int64), t_max, self.normalize, self.eps)
        pct = torch.tensor([f / float(t_max) for f in frames], dtype=torch.float32)
        return inputs, pct


# ---------------------------------------------------------------------------------------------------------
# Neighbours of the hot path: waveform loading (+ augmentation, on the device) and
# transcript -> labels (host).
# ---------------------------------------------------------------------------------------------------------
class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class PCMClip(object):
    """What a loader worker hands on for one utterance: the int16 samples as read from the file plus the augmentation
    DRAWN for it (tempo factor, gain in dB; None = no augmentation).  The arithmetic -- int16 -> float, WSOLA tempo,
    gain, 16-bit requantisation -- happens on the GPU after collate (``ds2hip.ops.decode_augment``)."""
    __slots__ = ('pcm', 'tempo', 'gain_db')

    def __init__(self, pcm, tempo=None, gain_db=None):
        self.pcm, self.tempo, self.gain_db = pcm, tempo, gain_db

    def numel(self):
        return int(self.pcm.numel())


class RawAudioBatch(object):
    """A collated minibatch of ``PCMClip``s: ONE int16 buffer (page-locked when the DataLoader pins) + clip offsets +
    the per-clip augmentation parameters.  2 bytes per sample cross PCIe; everything else happens on the device."""

    def __init__(self, pcm, offsets, tempos=None, gains_db=None):
        self.pcm, self.offsets, self.tempos, self.gains_db = pcm, list(offsets), tempos, gains_db
        self.ready = None                            # event recorded behind an asynchronous upload (DevicePrefetcher)

    @classmethod
    def from_clips(cls, clips):
        offs = [0]
        for c in clips:
            offs.append(offs[-1] + c.numel())
        pcm = torch.cat([c.pcm.reshape(-1) for c in clips]) if clips else torch.zeros(0, dtype=torch.int16)
        aug = any(c.tempo is not None or c.gain_db is not None for c in clips)
        tempos = [1.0 if c.tempo is None else float(c.tempo) for c in clips] if aug else None
        gains = [0.0 if c.gain_db is None else float(c.gain_db) for c in clips] if aug else None
        return cls(pcm, offs, tempos, gains)

    def __len__(self):
        return len(self.offsets) - 1

    def pin_memory(self):                            # torch DataLoader(pin_memory=True) calls this on custom batch types
        self.pcm = self.pcm.pin_memory()
        return self

    def to(self, device, non_blocking=False):
        out = RawAudioBatch(self.pcm.to(device, non_blocking=non_blocking), self.offsets, self.tempos, self.gains_db)
        return out


class ToTensor(object):
    """16-bit PCM mono WAV -> waveform (reference ``codes/transforms.py:130-224``).

    The reference decodes with torchaudio and, with ``augment=True``, pipes every training clip through
    ``sox ... tempo T gain G`` (T, G drawn uniformly from the ranges, printed with three decimals).  Here a loader worker
    only READS the file's int16 samples and DRAWS (tempo, gain) exactly as the reference does (``np.random.uniform``,
    tempo first): with ``defer=True`` (what the training loader uses) it returns a ``PCMClip`` and the decode + WSOLA
    tempo + gain + 16-bit requantisation run on the GPU for the whole minibatch after collate; with ``defer=False``
    (the reference's per-clip contract) the same kernels run at once and a 1-D float tensor comes back.  There is no
    host implementation in the product; ``oracle/audio.py`` specifies the arithmetic (sox itself is absent from the
    reference tree, so the tempo change is the published WSOLA algorithm with sox's defaults, not sox's samples).

    ``scale`` is the amplitude contract of ``torchaudio.load`` (reference ``codes/transforms.py:156-161``), which changed
    between torchaudio versions and which the log1p of the spectrogram is NOT invariant to: ``'unit'`` (default) gives
    samples in [-1, 1) (int16 / 32768); ``'int32'`` gives int16 * 65536, the un-normalised floats of the mid-2018
    torchaudio master the released checkpoints were most likely trained with (unpinned: that torchaudio build is not in the
    reference tree).  With ``defer=True`` the scale is applied by the ``BatchSpectrogram`` that decodes the minibatch.

    The non-deferred path runs device kernels, so it cannot run inside a forked DataLoader worker (the child would have
    to re-initialise the GPU): it raises there -- use ``defer=True`` (what ``get_default_transforms`` builds) or
    ``num_workers=0``."""

    def __init__(self, sample_rate=16000, augment=False, tempo_range=(0.85, 1.15), gain_range=(-6, 8), defer=False,
                 device='cuda', scale=None):
        self.sample_rate, self.augment = sample_rate, augment
        self.tempo_range, self.gain_range = tempo_range, gain_range
        self.defer, self.device = defer, device
        self.scale = ops.amplitude_scale(scale)

    def _load(self, path):
        import wave

        import numpy as np
        with wave.open(path, 'rb') as w:
            assert w.getframerate() == self.sample_rate, 'sample rate mismatch'
            assert w.getsampwidth() == 2 and w.getnchannels() == 1, 'expected 16-bit mono PCM'
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
        return torch.from_numpy(pcm.astype(np.int16, copy=True))

    def __call__(self, path):
        import numpy as np
        clip = PCMClip(self._load(path))
        if self.augment:
            clip.tempo = float(np.random.uniform(low=self.tempo_range[0], high=self.tempo_range[1]))
            clip.gain_db = float(np.random.uniform(low=self.gain_range[0], high=self.gain_range[1]))
        if self.defer:
            return clip
        if torch.utils.data.get_worker_info() is not None:
            raise RuntimeError('ToTensor(defer=False) decodes on the GPU and cannot run in a DataLoader worker process; '
                               'build the transform with defer=True (the minibatch is then decoded on the device after '
                               'collate) or use num_workers=0')
        batch = RawAudioBatch.from_clips([clip]).to(self.device)
        wav, _ = ops.decode_augment(batch.pcm, batch.offsets, batch.tempos, batch.gains_db, self.sample_rate,
                                    scale=self.scale)
        return wav.cpu()

    def __repr__(self):
        return '{}(sample_rate={}, augment={}, tempo_range={}, gain_range={})'.format(
            self.__class__.__name__, self.sample_rate, self.augment, self.tempo_range, self.gain_range)


def waveform_scale(transform):
    """The amplitude scale of the ``ToTensor`` stage of ``transform`` (1/32768 when there is none): what the
    ``BatchSpectrogram`` that decodes its deferred clips must apply."""
    for t in getattr(transform, 'transforms', [transform]):
        if isinstance(t, ToTensor):
            return t.scale
    return ops.UNIT_SCALE


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
