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


def _run(clips, tempos=None, gains=None):
    from ds2hip import ops
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).tolist()
    pcm = torch.from_numpy(np.concatenate(clips)).to('cuda')
    wav, out_offs = ops.decode_augment(pcm, offs, tempos, gains)
    wav = wav.cpu().numpy()
    return [wav[out_offs[i]:out_offs[i + 1]] for i in range(len(clips))]


def test_decode_is_bit_exact():
    rng = np.random.default_rng(0)
    clips = [_pcm(rng, n, 0.5) for n in (1, 7, 16000, 33333)] + [np.asarray([-32768, 32767, 0, -1, 1], np.int16)]
    for got, c in zip(_run(clips), clips):
        assert np.array_equal(got, oa.pcm16_to_float(c))


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
