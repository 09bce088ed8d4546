"""Oracle pins for the parts whose reference lives in absent third-party code or un-importable files."""
import numpy as np
import torch

from oracle import ctc as octc
from oracle import host
from oracle import spectrogram as ospec


# ---------------------------------------------------------------- spectrogram
def test_stft_matches_torch_stft():
    rng = np.random.default_rng(5)
    x = (0.1 * rng.standard_normal(16000 * 3 + 77)).astype(np.float32)
    mag = ospec.stft_magnitude(x, dtype=np.float64)
    assert mag.shape == (1 + x.shape[0] // 160, 161)
    win = torch.hann_window(320, periodic=False, dtype=torch.float64)
    ref = torch.stft(torch.from_numpy(x).double(), 320, 160, 320, win, center=True, pad_mode='reflect',
                     return_complex=True).abs().T.numpy()
    np.testing.assert_allclose(mag, ref, atol=1e-9)


def test_log_spectrogram_statistics():
    rng = np.random.default_rng(6)
    x = (0.1 * rng.standard_normal(16000)).astype(np.float32)
    s = ospec.log_spectrogram(x)
    assert s.shape == (101, 161) and s.dtype == np.float32
    assert abs(s.mean()) < 1e-5
    assert abs(s.astype(np.float64).std(ddof=1) - 1.0) < 1e-4


def test_hann_window_is_torch_symmetric():
    np.testing.assert_allclose(ospec.hann_symmetric(320), torch.hann_window(320, periodic=False,
                                                                           dtype=torch.float64).numpy(), atol=1e-14)


def test_batch_collate_of_spectrograms():
    rng = np.random.default_rng(7)
    wavs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in (16000, 24123, 19999)]
    x, pct = ospec.batch_log_spectrogram(wavs)
    assert x.shape == (3, 1 + 24123 // 160, 161)
    assert pct[1] == 1.0 and np.all(x[0, 101:] == 0)


# ------------------------------------------------------------------------ CTC
def test_ctc_known_answers_brute_force():
    rng = np.random.default_rng(11)
    for t_len, label in ((3, [1]), (4, [1, 2]), (4, [2, 2]), (4, [1, 1]), (3, [1, 1]), (2, [])):
        acts = rng.standard_normal((t_len, 1, 4))
        costs, _ = octc.ctc_loss_and_grad(acts, label, [t_len], [len(label)])
        bf = octc.ctc_brute_force(acts[:, 0], label)
        if np.isinf(bf):
            assert np.isinf(costs[0])
        else:
            assert abs(costs[0] - bf) < 1e-10, (t_len, label)


def test_ctc_matches_torch_with_ragged_lengths():
    rng = np.random.default_rng(12)
    for nalpha in (29, 43):
        t_max, bsz = 30, 4
        acts = rng.standard_normal((t_max, bsz, nalpha)).astype(np.float32)
        label_lens = [5, 0, 9, 3]
        act_lens = [30, 12, 25, 7]
        labels = rng.integers(1, nalpha, size=sum(label_lens))
        labels[0] = labels[1]                      # a repeated label
        costs, grad = octc.ctc_loss_and_grad(acts, labels, act_lens, label_lens)
        loss_t, grad_t = octc.ctc_torch(acts, labels, act_lens, label_lens)
        assert abs(costs.sum() - loss_t) < 1e-4 * abs(loss_t)
        np.testing.assert_allclose(grad, grad_t, atol=5e-5)   # torch side is float32
        assert np.all(grad[12:, 1] == 0)           # frames past act_len get zero gradient


def test_ctc_gradient_finite_difference():
    rng = np.random.default_rng(13)
    acts = rng.standard_normal((6, 1, 5))
    label = [2, 2, 4]
    _, grad = octc.ctc_loss_and_grad(acts, label, [6], [3])
    eps = 1e-6
    for (t, k) in ((0, 2), (3, 0), (5, 4)):
        a2 = acts.copy(); a2[t, 0, k] += eps
        a1 = acts.copy(); a1[t, 0, k] -= eps
        fd = (octc.ctc_loss_and_grad(a2, label, [6], [3])[0][0] - octc.ctc_loss_and_grad(a1, label, [6], [3])[0][0]) / (2 * eps)
        assert abs(fd - grad[t, 0, k]) < 1e-6


# ----------------------------------------------------------------------- host
def test_out_sizes_float32_truncation():
    # (T_i / T_max) stored as float32, multiplied by T in float32, truncated (codes/engine.py:16)
    for t_i, t_max in ((101, 1501), (1501, 1501), (747, 1501), (233, 301), (1000, 1500)):
        t_out = (t_max + 9) // 2 - 9
        pct = torch.tensor([t_i / float(t_max)], dtype=torch.float32)
        want = (pct * t_out).int().numpy()
        got = host.out_sizes(np.asarray([t_i / float(t_max)], dtype=np.float32), t_out)
        assert got.dtype == np.int32 and got[0] == want[0]


def test_collate_layout():
    a = np.ones((3, 4), np.float32); b = 2 * np.ones((5, 4), np.float32)
    x, tg, pct, ts = host.collate([(a, [1, 2]), (b, [3])])
    assert x.shape == (2, 5, 4) and np.all(x[0, 3:] == 0) and np.all(x[1] == 2)
    assert tg.tolist() == [1, 2, 3] and tg.dtype == np.int32
    assert ts.tolist() == [2, 1] and pct.dtype == np.float32 and pct[0] == np.float32(0.6)


def test_greedy_decode_known_answers():
    labels = ['_', ' ', 'A', 'B']
    seq = [2, 2, 0, 2, 3, 3, 0, 1, 2]          # A A _ A B B _ ' ' A
    probs = np.zeros((1, len(seq), 4), np.float32)
    probs[0, np.arange(len(seq)), seq] = 1.0
    s, o = host.greedy_decode(probs, [len(seq)], labels)
    assert s == ['AAB A'] and o[0].tolist() == [0, 3, 4, 7, 8]
    s, _ = host.greedy_decode(probs, [3], labels)
    assert s == ['A']
    probs[0, 0] = 0.25                          # tie -> first index (blank)
    s, _ = host.greedy_decode(probs, [2], labels)
    assert s == ['A']                           # frame 0 became blank, so frame 1's 'A' is kept


def test_greedy_tie_then_symbol():
    labels = ['_', 'A']
    probs = np.asarray([[[0.5, 0.5], [0.1, 0.9]]], np.float32)
    s, o = host.greedy_decode(probs, [2], labels)
    assert s == ['A'] and o[0].tolist() == [1]


def test_edit_distances_and_cer():
    assert host.levenshtein('kitten', 'sitting') == 3
    assert host.cer_distance('a b', 'ab') == 0
    assert host.wer_distance('the cat sat', 'the cat sat down') == 1
    cer, wer = host.corpus_cer_wer(['ab', 'abcd'], ['ab', 'abce'])
    assert abs(cer - 100.0 * 1 / 6) < 1e-12 and wer == 50.0
    assert abs(host.mean_utterance_cer(['ab', 'abcd'], ['ab', 'abce']) - 12.5) < 1e-12


def test_clip_and_nesterov_match_torch():
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(50))
    opt = torch.optim.SGD([p], lr=3e-4, momentum=0.9, nesterov=True)
    params = [p.detach().numpy().copy()]
    bufs = [np.zeros(50, np.float32)]
    for step in range(3):
        g = torch.randn(50) * 100
        p.grad = g.clone()
        total = torch.nn.utils.clip_grad_norm_([p], 400.0)
        tot2, gs = host.clip_grad_norm([g.numpy().copy()], 400.0)
        assert abs(tot2 - total.item()) < 1e-3
        opt.step()
        params, bufs = host.sgd_nesterov(params, gs, bufs, 3e-4, 0.9, step == 0)
        np.testing.assert_allclose(params[0], p.detach().numpy(), atol=1e-6)


def test_ddp_bin_partition():
    assert host.ddp_bins(10, 2, 2, 0) == [[0, 1], [4, 5], [8, 9]]
    assert host.ddp_bins(10, 2, 2, 1) == [[2, 3], [6, 7], [0, 1]]     # wraps to even out


def test_golden_fixtures_hold_exactly_the_keys_the_generator_writes():
    """A fixture that was not regenerated after tests/golden/make_golden.py changed shows up here (ref_tiny
    and ref_full lacked ``tstride``).  The key set is derived from the generator's case table, not from the reference."""
    import os

    import numpy as np

    from tests.golden.make_golden import CASES, HERE, expected_keys
    on_disk = sorted(f for f in os.listdir(HERE) if f.endswith('.npz'))
    assert on_disk == sorted(CASES), 'fixtures and generator cases differ'
    for name in CASES:
        with np.load(os.path.join(HERE, name)) as f:
            assert set(f.files) == expected_keys(name), (name, sorted(set(f.files) ^ expected_keys(name)))
