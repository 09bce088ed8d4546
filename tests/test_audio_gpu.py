"""The device-resident data path: int16 clips -> float -> WSOLA tempo -> gain + 16-bit
requantisation -> log-spectrogram, all HIP kernels behind the C ABI, against oracle/audio.py BIT FOR BIT (integer /
sample work), and the loader plumbing that feeds it (page-locked int16 batches uploaded one minibatch ahead)."""
import os
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import audio as oa  # noqa: E402
from oracle import spectrogram as ospec  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pcm(rng, n, amp=0.1):
    return (np.clip(amp * rng.standard_normal(n), -1, 1) * 32767).astype(np.int16)


def _run(clips, tempos=None, gains=None, scale=None):
    from ds2hip import ops
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).tolist()
    pcm = torch.from_numpy(np.concatenate(clips)).to('cuda')
    kw = {} if scale is None else {'scale': ops.amplitude_scale(scale)}
    wav, out_offs = ops.decode_augment(pcm, offs, tempos, gains, **kw)
    wav = wav.cpu().numpy()
    return [wav[out_offs[i]:out_offs[i + 1]] for i in range(len(clips))]


def test_decode_is_bit_exact():
    rng = np.random.default_rng(0)
    clips = [_pcm(rng, n, 0.5) for n in (1, 7, 16000, 33333)] + [np.asarray([-32768, 32767, 0, -1, 1], np.int16)]
    for got, c in zip(_run(clips), clips):
        assert np.array_equal(got, oa.pcm16_to_float(c))


@pytest.mark.parametrize('scale,value', [('unit', oa.UNIT_SCALE), ('int32', oa.INT32_SCALE), (0.37, 0.37)])
def test_amplitude_scale_is_bit_exact(scale, value):
    """The waveform amplitude contract of torchaudio.load (reference codes/transforms.py:156-161): [-1, 1) samples, the
    un-normalised int32-range floats of mid-2018 torchaudio, or any number -- plain decode, the augmented path (WSOLA on
    [-1, 1), requantised integer * scale) and a tempo-only batch (never requantised: the 16-bit output format belongs to the
    gain step; the float WSOLA output times scale / unit, the same samples at every scale)."""
    rng = np.random.default_rng(11)
    clips = [_pcm(rng, 20000, 0.4), np.asarray([-32768, 32767, 0, -1, 1] * 400, np.int16), _pcm(rng, 5000)]
    for got, c in zip(_run(clips, scale=scale), clips):
        assert np.array_equal(got, oa.pcm16_to_float(c, value))
    tempos, gains = [0.9, 1.1, 1.0], [2.5, 8.0, -3.0]
    for got, c, t, db in zip(_run(clips, tempos, gains, scale=scale), clips, tempos, gains):
        assert np.array_equal(got, oa.augment(c, t, db, scale=value))
    for got, c, t in zip(_run(clips, tempos, None, scale=scale), clips, tempos):
        want = oa.wsola_tempo(oa.pcm16_to_float(c), t)
        if scale != 'unit':
            want = want * np.float32(value / oa.UNIT_SCALE)
        assert np.array_equal(got, want)
    if scale == 'int32':                                   # what sox hands on: the 16-bit sample shifted into 32 bits
        assert float(_run([np.asarray([-32768, 12345], np.int16)], scale=scale)[0][1]) == 12345.0 * 65536.0


@pytest.mark.parametrize('n,tempo', [(16000, 0.9), (50001, 1.137), (240000, 0.85), (240000, 1.15), (1400, 1.1),
                                     (20000, 1.0), (123457, 0.9996), (31999, 1.0004)])
def test_wsola_tempo_is_bit_exact(n, tempo):
    rng = np.random.default_rng(n)
    clip = _pcm(rng, n)
    got = _run([clip], [tempo], None)[0]
    want = oa.wsola_tempo(oa.pcm16_to_float(clip), tempo)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_wsola_ties_and_a_ragged_batch_with_gain():
    """Silence makes every candidate tie (the first must win); a batch mixes augmented and plain clips; +8 dB on a loud
    clip exercises the clip of the 16-bit requantisation."""
    rng = np.random.default_rng(5)
    tone = (0.6 * 32767 * np.sin(2 * np.pi * 300.0 * np.arange(40000) / 16000.0)).astype(np.int16)
    clips = [np.zeros(30000, np.int16), _pcm(rng, 47000, 0.3), tone, _pcm(rng, 900), _pcm(rng, 16000)]
    tempos = [1.1, 0.87, 1.149, 1.05, 1.0]
    gains = [3.0, 8.0, 7.999, -6.0, 0.0]
    got = _run(clips, tempos, gains)
    for g, c, t, db in zip(got, clips, tempos, gains):
        want = oa.augment(c, t, db)
        assert g.shape == want.shape and np.array_equal(g, want)
    assert float(np.abs(got[1]).max()) == 32767 / 32768.0 or float(got[1].min()) == -1.0     # it did clip


def _wav_corpus(tmp_path, lens, seed=0):
    rng = np.random.default_rng(seed)
    rows = []
    for i, n in enumerate(lens):
        with wave.open(str(tmp_path / ('u%d.wav' % i)), 'wb') as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(_pcm(rng, n).astype('<i2').tobytes())
        (tmp_path / ('u%d.txt' % i)).write_text('hello world\n')
        rows.append('u%d.wav,u%d.txt,%.3f' % (i, i, n / 16000.0))
    (tmp_path / 'm.csv').write_text('\n'.join(rows) + '\n')


def test_loader_to_spectrogram_pipeline_matches_oracle(tmp_path):
    """wav files -> AudioDataset(ToTensor(augment, defer)) -> pinned RawAudioBatch -> DevicePrefetcher (upload one bin
    ahead on a copy stream) -> BatchSpectrogram == the oracle run on the same files with the same draws."""
    from codes.data import AudioDataLoader, AudioDataset, DevicePrefetcher
    from codes.sampler import BucketingSampler
    from codes.transforms import BatchSpectrogram, Compose, RawAudioBatch, ToLabel, ToTensor
    lens = [16000, 20000, 26000, 33000, 41000, 50000, 60000]
    _wav_corpus(tmp_path, lens)
    ds = AudioDataset(str(tmp_path), str(tmp_path / 'm.csv'), Compose([ToTensor(augment=True, defer=True)]),
                      ToLabel(os.path.join(ROOT, 'data', 'labels.en.json')))
    sampler = BucketingSampler(ds, batch_size=3)
    loader = DevicePrefetcher(AudioDataLoader(ds, batch_sampler=sampler, raw_audio=True, num_workers=0,
                                              pin_memory=True))
    front = BatchSpectrogram()
    np.random.seed(11)
    got = []
    for wavs, targets, _, sizes in loader:
        assert isinstance(wavs, RawAudioBatch) and wavs.pcm.is_cuda and wavs.ready is not None
        inputs, pct = front(wavs)
        got.append((inputs.cpu().numpy(), pct.numpy(), list(wavs.tempos), list(wavs.gains_db)))
    assert [g[0].shape[0] for g in got] == [3, 3, 1]
    # the oracle on the same files with the draws each batch recorded (which draw follows which is pinned by the CPU test)
    for (inputs, pct, tempos, gains), ids in zip(got, sampler.bins):
        wavs = []
        for row, (t, g) in enumerate(zip(tempos, gains)):
            n_out = int(round(pct[row] * inputs.shape[1]))             # frames of this row: identifies the file
            cand = [i for i in ids if 1 + oa.wsola_out_len(lens[i], t) // 160 == n_out]
            assert len(cand) == 1, (cand, n_out)
            with wave.open(str(tmp_path / ('u%d.wav' % cand[0])), 'rb') as w:
                pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
            wavs.append(oa.augment(pcm, t, g))
        ref_in, ref_pct = ospec.batch_log_spectrogram(wavs)
        np.testing.assert_allclose(inputs, ref_in, atol=2e-4)
        assert np.array_equal(pct, ref_pct)


def test_totensor_reference_contract_runs_the_device_kernels(tmp_path):
    """``ToTensor(...)(path)`` with the reference's per-clip contract: a 1-D float tensor, produced by the same kernels."""
    from codes.transforms import ToTensor
    _wav_corpus(tmp_path, [20000])
    path = str(tmp_path / 'u0.wav')
    with wave.open(path, 'rb') as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2')
    plain = ToTensor()(path)
    assert plain.device.type == 'cpu' and plain.dtype == torch.float32 and plain.ndim == 1
    assert np.array_equal(plain.numpy(), oa.pcm16_to_float(pcm))
    np.random.seed(3)
    t, g = np.random.uniform(0.85, 1.15), np.random.uniform(-6, 8)
    np.random.seed(3)
    aug = ToTensor(augment=True)(path)
    assert np.array_equal(aug.numpy(), oa.augment(pcm, t, g))


def test_amplitude_scale_through_transforms_config_and_spectrogram(tmp_path):
    """``training.audio_scale`` of the config reaches both decode paths -- the per-clip ``ToTensor`` and the minibatch
    ``BatchSpectrogram`` -- and changes the spectrogram (log1p is not scale invariant), exactly as the oracle says."""
    from codes.transforms import BatchSpectrogram, RawAudioBatch, waveform_scale
    from codes.utils import training_utils as tu
    from codes.utils.model_utils import AttrDict
    _wav_corpus(tmp_path, [20000, 26000])
    paths = [str(tmp_path / 'u0.wav'), str(tmp_path / 'u1.wav')]
    pcms = []
    for p in paths:
        with wave.open(p, 'rb') as w:
            pcms.append(np.frombuffer(w.readframes(w.getnframes()), dtype='<i2'))
    specs = {}
    for name, value in (('unit', oa.UNIT_SCALE), ('int32', oa.INT32_SCALE)):
        cfg = AttrDict({'model': AttrDict({'langs': ['en']}), 'training': AttrDict({'audio_scale': name})})
        train_t, val_t, _ = tu.get_default_transforms(os.path.join(ROOT, 'data'), cfg)
        assert waveform_scale(val_t) == value
        clips = [val_t(p) for p in paths]                                   # deferred: int16 clips
        inputs, pct = BatchSpectrogram(scale=waveform_scale(val_t))(RawAudioBatch.from_clips(clips))
        ref_in, ref_pct = ospec.batch_log_spectrogram([oa.pcm16_to_float(p, value) for p in pcms])
        np.testing.assert_allclose(inputs.cpu().numpy(), ref_in, atol=2e-4)
        assert np.array_equal(pct.numpy(), ref_pct)
        _, val_cpu, _ = tu.get_default_transforms(os.path.join(ROOT, 'data'), cfg, gpu_frontend=False)
        per_clip = val_cpu(paths[0])                                        # the reference's per-utterance contract
        np.testing.assert_allclose(per_clip.numpy(), ospec.log_spectrogram(oa.pcm16_to_float(pcms[0], value)), atol=2e-4)
        specs[name] = inputs.cpu().numpy()
    assert float(np.abs(specs['unit'] - specs['int32']).max()) > 0.1       # the two contracts are different inputs
