"""End-to-end parity of the HIP model / trainer against the reference's golden outputs and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import host  # noqa: E402
from oracle.model import OracleDeepSpeech, conv_out_time, seeded_state_dict  # noqa: E402
from tests.golden.make_golden import seeded_inputs, seeded_labels  # noqa: E402

LABELS = ['_', ' ', "'"] + [chr(65 + i) for i in range(26)]


def _build(kwargs):
    from codes.model import DeepSpeech
    shapes = OracleDeepSpeech(**kwargs)
    model = DeepSpeech(**kwargs)
    assert list(model.state_dict().keys()) == list(shapes.state_dict().keys())
    model.load_state_dict(seeded_state_dict(shapes, 1234))
    return model.to('cuda')


def _case(golden_dir, name, kwargs, bsz, t_in, lengths, label_lens, full):
    from codes.ctc import CTCLoss
    from codes.decoder import GreedyDecoder
    g = np.load(os.path.join(golden_dir, name))
    model = _build(kwargs)
    x = torch.from_numpy(seeded_inputs(77, bsz, t_in, lengths=lengths)).to('cuda')
    labels = torch.from_numpy(seeded_labels(78, label_lens, 29))
    model.train()
    logits = model(x)                                            # autograd path
    assert tuple(logits.shape) == (bsz, conv_out_time(t_in), 29)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g['logits'], rtol=0, atol=1e-3)
    pct = torch.from_numpy(g['pct'])
    out_sizes = (pct * logits.shape[1]).int()
    assert np.array_equal(out_sizes.numpy(), g['out_sizes'])
    loss = CTCLoss()(logits.transpose(0, 1), labels, out_sizes, torch.tensor(label_lens, dtype=torch.int32))
    assert tuple(loss.shape) == (1,)
    ref_loss = float(g['loss_sum'])
    assert abs(float(loss.item()) - ref_loss) <= 1e-4 * abs(ref_loss)
    (loss / bsz).sum().backward()
    for k, p in model.named_parameters():
        gn = float(np.sqrt((p.grad.cpu().numpy().astype(np.float64) ** 2).sum()))
        if k in ('conv.0.bias', 'conv.3.bias'):
            # a bias in front of a BatchNorm has an exactly-zero gradient; both sides hold only round-off
            assert gn < 1e-3 and float(g['gnorm_' + k]) < 1e-3, k
            continue
        assert abs(gn - float(g['gnorm_' + k])) <= 2e-3 * float(g['gnorm_' + k]) + 1e-6, k
        if full:
            ref = g['grad_' + k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max() + 1e-7,
                                       err_msg=k)
        else:
            flat = p.grad.cpu().numpy().reshape(-1)
            stride = max(1, flat.shape[0] // 1024)
            ref = g['gsample_' + k]
            np.testing.assert_allclose(flat[::stride][:1024], ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max() + 1e-7,
                                       err_msg=k)
    for k, v in model.state_dict().items():
        if 'running' in k:
            np.testing.assert_allclose(v.cpu().numpy(), g['buf_' + k], rtol=1e-4, atol=1e-5, err_msg=k)
    model.eval()
    with torch.no_grad():
        probs = model(x)
    np.testing.assert_allclose(probs.cpu().numpy(), g['probs'], rtol=0, atol=1e-3)
    # greedy strings: bit-identical to the oracle's decode of the REFERENCE probabilities
    dec = GreedyDecoder(LABELS)
    strings, offsets = dec.decode(probs, out_sizes)
    want, want_off = host.greedy_decode(g['probs'], g['out_sizes'], LABELS)
    assert [s[0] for s in strings] == want
    for o, w in zip(offsets, want_off):
        assert np.array_equal(o[0].numpy(), w)
    return model


def test_tiny_model_against_reference_golden(golden_dir):
    _case(golden_dir, 'ref_tiny.npz', dict(rnn_hidden_size=32, num_rnn_layers=2), 3, 121, [121, 97, 64], [9, 6, 4],
          full=True)


def test_full_model_against_reference_golden(golden_dir):
    _case(golden_dir, 'ref_full.npz', dict(), 2, 301, [301, 233], [30, 21], full=False)


def test_tiny_intermediates_localise_errors(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ref_tiny.npz'))
    model = _build(dict(rnn_hidden_size=32, num_rnn_layers=2))
    x = torch.from_numpy(seeded_inputs(77, 3, 121, lengths=[121, 97, 64])).to('cuda')
    model.train()
    model._ensure_flat()
    acts, sv = model._forward_impl(x, training=True, need_grad=True)
    np.testing.assert_allclose(sv['a1'].cpu().numpy(), g['inter_conv1'], atol=1e-4)
    for li in (0, 1):
        h = sv['layers'][li]['hout']
        np.testing.assert_allclose((h[0] + h[1]).cpu().numpy(), g['inter_rnn%d' % li], atol=1e-4)
    x0 = sv['layers'][0]['xin']                                   # (T,B,672) = conv2 output re-laid-out
    ref = torch.from_numpy(g['inter_conv2'])
    b, c, d, t = ref.shape
    np.testing.assert_allclose(x0.cpu().numpy(), ref.reshape(b, c * d, t).permute(2, 0, 1).numpy(), atol=1e-4)


def test_trainer_steps_match_oracle_sgd():
    from codes.engine import Trainer
    kwargs = dict(rnn_hidden_size=64, num_rnn_layers=3, num_classes=29)
    oracle = OracleDeepSpeech(**kwargs)
    oracle.load_state_dict(seeded_state_dict(oracle, 99))
    model = _build(kwargs)
    model.load_state_dict(seeded_state_dict(oracle, 99))
    model.to('cuda')
    opt_o = torch.optim.SGD(oracle.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=5.0)           # small max_norm: the clip engages
    rng = np.random.default_rng(3)
    for step in range(3):
        t_in = 90 + 10 * step
        lengths = [t_in, t_in - 17, t_in - 40, 51]
        x = torch.from_numpy(seeded_inputs(100 + step, 4, t_in, lengths=lengths))
        label_lens = [6, 4, 3, 2]
        labels = torch.from_numpy(rng.integers(1, 29, size=sum(label_lens)).astype(np.int32))
        pct = torch.tensor([n / float(t_in) for n in lengths], dtype=torch.float32)
        sizes = torch.tensor(label_lens, dtype=torch.int32)
        oracle.train()
        logits = oracle(x)
        out_sizes = (pct * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), labels.long(), out_sizes.long(), sizes.long(),
                          blank=0, reduction='sum') / 4
        opt_o.zero_grad()
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
        opt_o.step()
        got = trainer.update((x, labels, pct, sizes))
        assert abs(got - float(loss.item())) <= 2e-4 * abs(float(loss.item()))
        assert abs(trainer.last_grad_norm - float(total)) <= 5e-3 * float(total)
        for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), atol=5e-5, err_msg='%s step %d' % (k, step))
    # the torch optimizer's state is a view of the trainer's flat momentum buffer
    p0 = next(model.parameters())
    assert opt.state[p0]['momentum_buffer'].data_ptr() == trainer._buf.data_ptr()


def test_frontend_to_strings_pipeline():
    """raw audio -> GPU spectrogram -> model(eval) -> greedy strings == the same through the oracle."""
    from codes.decoder import GreedyDecoder
    from codes.transforms import BatchSpectrogram, ToSpectrogram
    from oracle import spectrogram as ospec
    rng = np.random.default_rng(8)
    wavs = [np.clip(0.1 * rng.standard_normal(n), -1, 1).astype(np.float32) for n in (16000, 20480, 17777)]
    inputs, pct = BatchSpectrogram()([torch.from_numpy(w) for w in wavs])
    ref_in, ref_pct = ospec.batch_log_spectrogram(wavs)
    np.testing.assert_allclose(inputs.cpu().numpy(), ref_in, atol=2e-4)
    assert np.array_equal(pct.numpy(), ref_pct)
    one = ToSpectrogram(librosa_compat=True)(torch.from_numpy(wavs[0]))
    assert one.device.type == 'cpu'
    np.testing.assert_allclose(one.numpy(), ospec.log_spectrogram(wavs[0]), atol=2e-4)
    kwargs = dict(rnn_hidden_size=64, num_rnn_layers=2)
    oracle = OracleDeepSpeech(**kwargs)
    oracle.load_state_dict(seeded_state_dict(oracle, 5, scale=3.0))
    model = _build(kwargs)
    model.load_state_dict(seeded_state_dict(oracle, 5, scale=3.0))
    model.eval()
    oracle.eval()
    with torch.no_grad():
        probs = model(inputs)
        ref_probs = oracle(torch.from_numpy(ref_in))
    np.testing.assert_allclose(probs.cpu().numpy(), ref_probs.numpy(), atol=1e-3)
    sizes = host.out_sizes(ref_pct, probs.shape[1])
    strings, _ = GreedyDecoder(LABELS).decode(probs, torch.from_numpy(sizes))
    want, _ = host.greedy_decode(ref_probs.numpy(), sizes, LABELS)
    assert [s[0] for s in strings] == want


def test_full_size_step_properties():
    """B=10 x 15 s (T_in=1501 -> T=746) on the real 5xBiGRU-800: shapes, finiteness, loss decreases."""
    from codes.engine import Trainer
    from codes.model import DeepSpeech
    torch.manual_seed(0)
    model = DeepSpeech().to('cuda')
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=400)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((10, 1501, 161)).astype(np.float32))
    label_lens = [int(v) for v in rng.integers(100, 210, size=10)]
    labels = torch.from_numpy(rng.integers(1, 29, size=sum(label_lens)).astype(np.int32))
    pct = torch.ones(10)
    sizes = torch.tensor(label_lens, dtype=torch.int32)
    losses = [trainer.update((x, labels, pct, sizes)) for _ in range(4)]
    assert all(np.isfinite(v) and v > 0 for v in losses)
    assert losses[-1] < losses[0]
    assert np.isfinite(trainer.last_grad_norm)
    for p in model.parameters():
        assert torch.isfinite(p).all()
    model.eval()
    with torch.no_grad():
        probs = model(x.to('cuda'))
    assert tuple(probs.shape) == (10, 746, 29)
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize('bsz,t_in', [(3, 181), (10, 400), (2, 1001)])
def test_forward_pass_is_bitwise_reproducible(bsz, t_in):
    """No float atomics on the forward path: eval outputs and the training-mode logits are bit-identical from call to call
    (the greedy transcript of an utterance must not depend on the run).  Short inputs matter: the conv2 gather kernel and
    the GEMMs split K over workgroups only in the backward pass."""
    from codes.model import DeepSpeech
    torch.manual_seed(11)
    model = DeepSpeech().to('cuda')
    x = torch.randn(bsz, t_in, 161, device='cuda')
    model.eval()
    with torch.no_grad():
        a = model(x)
        b = model(x)
    assert torch.equal(a, b)
    model.train()
    acts0, _ = model._forward_impl(x.contiguous().float(), training=True, need_grad=False)
    acts1, _ = model._forward_impl(x.contiguous().float(), training=True, need_grad=False)
    assert torch.equal(acts0, acts1)


@pytest.mark.parametrize('frozen_conv', [False, True])
def test_deferred_readback_gives_the_same_steps(frozen_conv):
    """``Trainer.update(defer=True)`` (the host reads step i's loss after enqueuing step i + 1) against the synchronous
    form: same losses, gradient norm and parameters; ``run()`` reports every step exactly once.

    ``frozen_conv``: the variant WITHOUT the discontinuity that forces the loose bounds below.  The only non-smooth
    function of the model is the conv block's hard clip, and its gradient mask only reaches the conv filters: with the
    conv layers frozen (fine-tuning's ``freeze_layers``, training_utils.py:52-54) a flipped mask changes nothing that is
    applied, what is left is the last-bit noise of the float atomics, and the round-2 tolerances hold -- tight enough to
    tell an ordering or race error between the deferred and the synchronous path from that noise."""
    from codes.engine import PendingLoss, Trainer
    from codes.model import DeepSpeech
    kw = dict(rnn_hidden_size=96, num_rnn_layers=2, num_classes=29)
    rng = np.random.default_rng(5)
    batches = []
    for t_in in (181, 240, 121, 240):
        x = torch.from_numpy(rng.standard_normal((3, t_in, 161)).astype(np.float32))
        lens = [int(v) for v in rng.integers(3, 12, size=3)]
        labels = torch.from_numpy(rng.integers(1, 29, size=sum(lens)).astype(np.int32))
        batches.append((x, labels, torch.ones(3), torch.tensor(lens, dtype=torch.int32)))
    runs = []
    for mode in ('sync', 'defer', 'run'):
        torch.manual_seed(3)
        model = DeepSpeech(**kw).to('cuda')
        if frozen_conv:
            for p in model.conv.parameters():
                p.requires_grad_(False)
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, nesterov=True)
        tr = Trainer(model, opt, device='cuda', max_norm=50)
        assert tr._fused
        if mode == 'sync':
            losses = [tr.update(b) for b in batches]
        elif mode == 'defer':
            handles = [tr.update(b, defer=True) for b in batches]
            assert all(isinstance(h, PendingLoss) for h in handles)
            assert all(h._value is not None for h in handles[:-1])       # resolved when the next step was enqueued
            assert tr.flush() == handles[-1].result()
            losses = [h.result() for h in handles]
        else:
            seen = []
            tr.run(batches, num_epochs=1, on_iteration=lambda t, e, i, loss: seen.append((i, loss)))
            assert [i for i, _ in seen] == list(range(len(batches)))
            losses = [v for _, v in seen]
        runs.append((losses, tr.last_grad_norm, [p.detach().clone() for p in model.parameters()], tr.iteration))
    # Not bit-for-bit: the split-K weight-gradient GEMMs add their partial products with float atomics, in an order that
    # changes from run to run -- two synchronous runs differ in the last bits too.  And the last bits decide on which side of
    # the conv block's hard clip an activation within 1e-7 of the boundary falls (tools/attic/clip_boundary_probe.py): one such flip
    # moves one output channel's conv filter gradient by a few per cent.  So: the first step tightly (same weights), later
    # steps and the parameters with room for a flip -- a step reported twice, skipped or applied out of order would be off by
    # orders of magnitude more.
    for losses, norm, params, it in runs[1:]:
        np.testing.assert_allclose(losses[0], runs[0][0][0], rtol=1e-6)
        np.testing.assert_allclose(losses, runs[0][0], rtol=1e-5 if frozen_conv else 1e-4)
        np.testing.assert_allclose(norm, runs[0][1], rtol=1e-5 if frozen_conv else 1e-3)
        assert it == len(batches)
        for a, b in zip(params, runs[0][2]):
            d = (a - b).abs()
            if frozen_conv:
                assert float(d.max()) <= 2e-6
            else:
                assert float(d.max()) <= 1e-4
                assert float((d > 2e-6).float().mean()) <= 0.01


def test_statistics_wait_reports_a_drained_stream_and_honours_its_timeout(monkeypatch):
    """The host's poll of the page-locked statistics slot (ops.wait_step_stats): a slot nobody writes is reported as soon as
    the stream has drained (not after the time-out); the time-out comes from DS2_STATS_TIMEOUT_S, is 120 s single-GPU and
    unlimited inside a process group (the statistics queue behind the gradient all-reduce: a slow peer is not an error)."""
    import time
    from ds2hip import ops
    slot = torch.empty(4, dtype=torch.float64).pin_memory()
    ops.arm_step_stats(slot)
    torch.cuda.synchronize()
    t0 = time.time()
    with pytest.raises(RuntimeError, match='never written'):
        ops.wait_step_stats(slot)
    assert time.time() - t0 < 5.0
    monkeypatch.delenv('DS2_STATS_TIMEOUT_S', raising=False)
    assert ops._stats_timeout() == 120.0
    monkeypatch.setenv('DS2_STATS_TIMEOUT_S', '7.5')
    assert ops._stats_timeout() == 7.5
    monkeypatch.setenv('DS2_STATS_TIMEOUT_S', '0')
    assert ops._stats_timeout() is None
    monkeypatch.delenv('DS2_STATS_TIMEOUT_S')
    monkeypatch.setattr(torch.distributed, 'is_initialized', lambda: True)
    assert ops._stats_timeout() is None


def test_step_statistics_readback_paths_agree(monkeypatch):
    """The step's one readback two ways: the statistics kernel writing the trainer's page-locked slot itself (the default;
    the host polls the memory) and the round-3 path (device buffer, non-blocking copy, polled event; DS2_STATS_DIRECT=0).
    Same loss and gradient norm, synchronous and deferred."""
    from codes import engine
    from codes.model import DeepSpeech
    kw = dict(rnn_hidden_size=64, num_rnn_layers=2, num_classes=29)
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((3, 150, 161)).astype(np.float32))
    lens = [5, 3, 7]
    labels = torch.from_numpy(rng.integers(1, 29, size=sum(lens)).astype(np.int32))
    batch = (x, labels, torch.ones(3), torch.tensor(lens, dtype=torch.int32))
    got = {}
    for direct in (True, False):
        monkeypatch.setattr(engine, '_STATS_DIRECT', direct)
        torch.manual_seed(5)
        model = DeepSpeech(**kw).to('cuda')
        tr = engine.Trainer(model, torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True), device='cuda',
                            max_norm=50)
        first = tr.update(batch)
        pend = tr.update(batch, defer=True)
        assert (pend.done is None) == direct
        got[direct] = (first, tr.last_grad_norm, pend.result(), tr.last_grad_norm)
    np.testing.assert_allclose(got[True], got[False], rtol=1e-5)


def test_beam_decoder_agrees_with_greedy_on_confident_outputs():
    """A width-1..16 prefix beam search returns the greedy transcript when every frame is confident."""
    from codes.decoder import BeamCTCDecoder, GreedyDecoder
    labels = "_'ABCDEFGHIJKLMNOPQRSTUVWXYZ "
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 29, size=(3, 60))
    logits = torch.full((3, 60, 29), -6.0)
    logits.scatter_(2, torch.from_numpy(ids)[..., None], 6.0)
    probs = torch.softmax(logits, -1).to('cuda')
    sizes = torch.tensor([60, 41, 7], dtype=torch.int32)
    want, _ = GreedyDecoder(labels).decode(probs, sizes)
    for width in (1, 16):
        got, offs = BeamCTCDecoder(labels, beam_width=width).decode(probs, sizes)
        assert got == want
        assert all(len(o[0]) == len(g[0]) for o, g in zip(offs, got))


@pytest.mark.parametrize('bsz,t_in,label_lens', [(4, 11, [1, 0, 1, 0]), (1, 13, [1]), (3, 29, [2, 0, 1]), (16, 31, [1] * 16)])
def test_edge_shapes_against_oracle(bsz, t_in, label_lens):
    """Shortest inputs the conv stack accepts (T_in = 11 -> one output step), a single utterance, empty transcripts,
    a batch of exactly one 16-row tile: one training step through the fused path against the oracle."""
    from codes.engine import Trainer
    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
    oracle = OracleDeepSpeech(**kwargs)
    oracle.load_state_dict(seeded_state_dict(oracle, 21))
    model = _build(kwargs)
    model.load_state_dict(seeded_state_dict(oracle, 21))
    model.to('cuda')
    opt_o = torch.optim.SGD(oracle.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=400)
    x = torch.from_numpy(seeded_inputs(7, bsz, t_in))
    rng = np.random.default_rng(bsz)
    labels = torch.from_numpy(rng.integers(1, 29, size=max(sum(label_lens), 0)).astype(np.int32))
    pct = torch.ones(bsz, dtype=torch.float32)
    sizes = torch.tensor(label_lens, dtype=torch.int32)
    oracle.train()
    logits = oracle(x)
    t_out = logits.shape[1]
    assert t_out == conv_out_time(t_in)
    out_sizes = (pct * t_out).int()
    loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), labels.long(), out_sizes.long(), sizes.long(), blank=0,
                      reduction='sum') / bsz
    opt_o.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(oracle.parameters(), 400)
    opt_o.step()
    got = trainer.update((x, labels, pct, sizes))
    assert abs(got - float(loss.item())) <= 2e-4 * max(abs(float(loss.item())), 1.0)
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), atol=5e-5, err_msg=k)


def test_fused_step_with_param_groups_and_frozen_conv():
    """Fine-tuning shape of the reference (pt_BR-finetune-freeze.json + per-layer learning rates): conv block frozen
    (no gradient, no update), rnns and fc in separate groups with their own learning rates -- still on the fused clip +
    SGD pass, and equal to torch on the oracle.

    The frozen block's BatchNorm: ``_freeze_layers`` puts it in eval mode (training_utils.py:52-54,73) and the reference's
    update step calls ``model.train()`` before every forward pass (codes/engine.py:51), which puts it back -- so under the
    trainer it normalises with BATCH statistics and keeps updating its running estimates.  The oracle side below does
    literally that (eval at set-up, ``train()`` per step)."""
    from codes.engine import Trainer
    from codes.utils.training_utils import _freeze_layers, get_per_params_lr
    from codes.utils.io_utils import AttrDict
    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
    oracle = OracleDeepSpeech(**kwargs)
    oracle.load_state_dict(seeded_state_dict(oracle, 31))
    model = _build(kwargs)
    model.load_state_dict(seeded_state_dict(oracle, 31))
    model.to('cuda')
    _freeze_layers(model, ['conv'])
    assert not model.conv[1].training and not model.conv[4].training and model.rnns[1].batch_norm.module.training
    for p in oracle.conv.parameters():
        p.requires_grad_(False)
    for m in oracle.conv.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()                                             # batch_norm_eval_mode, training_utils.py:52-54
    conf = AttrDict({'per_layer_lr': [['rnns', 2e-2], ['fc', 5e-3], ['base']]})
    opt = torch.optim.SGD(get_per_params_lr(model, conf), lr=1e-2, momentum=0.9, nesterov=True)
    opt_o = torch.optim.SGD([{'params': list(oracle.rnns.parameters()), 'lr': 2e-2},
                             {'params': list(oracle.fc.parameters()), 'lr': 5e-3},
                             {'params': list(oracle.conv.parameters())}], lr=1e-2, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=3.0)
    assert trainer._fused and len(trainer._spans) == 2
    rng = np.random.default_rng(1)
    for step in range(2):
        x = torch.from_numpy(seeded_inputs(40 + step, 3, 70))
        label_lens = [4, 3, 2]
        labels = torch.from_numpy(rng.integers(1, 29, size=sum(label_lens)).astype(np.int32))
        pct = torch.ones(3, dtype=torch.float32)
        sizes = torch.tensor(label_lens, dtype=torch.int32)
        oracle.train()                                       # codes/engine.py:51 -- the frozen BatchNorm is training again
        logits = oracle(x)
        out_sizes = (pct * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), labels.long(), out_sizes.long(), sizes.long(),
                          blank=0, reduction='sum') / 3
        opt_o.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in oracle.parameters() if p.requires_grad], 3.0)
        opt_o.step()
        got = trainer.update((x, labels, pct, sizes))
        assert abs(got - float(loss.item())) <= 2e-4 * abs(float(loss.item()))
        for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), atol=5e-5, err_msg='%s step %d' % (k, step))
    assert model.conv[1].training and int(model.conv[1].num_batches_tracked) == 2
    for k in ('conv.1.running_mean', 'conv.1.running_var', 'conv.4.running_mean', 'conv.4.running_var'):
        np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), oracle.state_dict()[k].numpy(), rtol=2e-5, atol=1e-6,
                                   err_msg=k)


def test_conv_batchnorm_held_in_inference_mode_inside_a_training_pass():
    """A caller that drives the model itself (not through the trainer, whose ``model.train()`` per step undoes it) can hold
    the frozen conv block's BatchNorm in inference mode: ``model.conv[i].eval()`` under ``model.train()``, torch's own
    semantics.  Logits and the gradients of the trainable layers against the oracle in the same state."""
    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
    oracle = OracleDeepSpeech(**kwargs)
    sd = seeded_state_dict(oracle, 33)
    rng = np.random.default_rng(33)
    for k in ('conv.1.running_mean', 'conv.4.running_mean'):
        sd[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, size=32).astype(np.float32))
    for k in ('conv.1.running_var', 'conv.4.running_var'):
        sd[k] = torch.from_numpy(rng.uniform(0.5, 2.0, size=32).astype(np.float32))
    oracle.load_state_dict(sd)
    model = _build(kwargs)
    model.load_state_dict(sd)
    model.to('cuda')
    for net in (model, oracle):
        net.train()
        for p in net.conv.parameters():
            p.requires_grad_(False)
        net.conv[1].eval()
        net.conv[4].eval()
    x = torch.from_numpy(seeded_inputs(44, 3, 90))
    dl = torch.from_numpy(rng.standard_normal((3, 40, 29)).astype(np.float32))
    want = oracle(x)
    (want * dl).sum().backward()
    got = model(x.to('cuda'))
    (got * dl.to('cuda')).sum().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=2e-4)
    assert int(model.conv[1].num_batches_tracked) == 0
    np.testing.assert_array_equal(model.conv[1].running_mean.cpu().numpy(), sd['conv.1.running_mean'].numpy())
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.requires_grad:
            scale = max(float(q.grad.abs().max()), 1e-6)
            np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), atol=2e-4 * scale, err_msg=k)
    model.rnns[1].batch_norm.module.eval()
    with pytest.raises(NotImplementedError):
        model(x.to('cuda'))


@pytest.mark.parametrize('optimizer', ['sgd_fused', 'sgd_autograd'])
def test_infinite_batch_loss_is_a_zero_gradient_step_on_both_paths(optimizer, caplog):
    """``_sanitize_loss`` (codes/engine.py:24-30) turns an infinite batch loss into ``0 * loss``: the reported value is 0,
    no utterance contributes a gradient, and ``optimizer.step()`` still runs (momentum moves the weights).  The fused
    step and the autograd step must both do exactly that -- equal to torch SGD stepping on all-zero gradients."""
    from codes.ctc import CTCLoss
    from codes.engine import Trainer
    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
    oracle = OracleDeepSpeech(**kwargs)
    oracle.load_state_dict(seeded_state_dict(oracle, 77))
    model = _build(kwargs)
    model.load_state_dict(seeded_state_dict(oracle, 77))
    model.to('cuda')
    opt_o = torch.optim.SGD(oracle.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    if optimizer == 'sgd_fused':
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    else:            # a (negligible) weight decay is not the fused kernel's form -> the reference-shaped autograd step
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=1e-30)
    trainer = Trainer(model, opt, CTCLoss(), device='cuda', max_norm=5.0)    # (clipped steps keep fp32 noise x lr small)
    assert trainer._fused == (optimizer == 'sgd_fused')
    x = torch.from_numpy(seeded_inputs(5, 3, 80))
    pct = torch.ones(3)
    ok_labels = (torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.int32), torch.tensor([3, 2, 1], dtype=torch.int32))
    # utterance 1 asks for 40 labels out of 35 output frames: infeasible, cost +inf
    bad_labels = (torch.tensor([1, 2] + [3] * 40 + [4], dtype=torch.int32), torch.tensor([2, 40, 1], dtype=torch.int32))
    for step, (labels, sizes) in enumerate((ok_labels, bad_labels, ok_labels)):
        oracle.train()
        logits = oracle(x)
        out_sizes = (pct * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), labels.long(), out_sizes.long(), sizes.long(), blank=0,
                          reduction='sum') / 3
        opt_o.zero_grad()
        if torch.isinf(loss):
            for p in oracle.parameters():
                p.grad = torch.zeros_like(p)                 # 0 * loss: zero gradients, the step still runs
            want = 0.0
        else:
            loss.backward()
            want = float(loss.item())
        torch.nn.utils.clip_grad_norm_(oracle.parameters(), 5.0)
        opt_o.step()
        got = trainer.update((x, labels, pct, sizes))
        assert abs(got - want) <= 2e-4 * max(abs(want), 1.0), (step, got, want)
        if step == 1:
            assert got == 0.0 and 'received an inf loss' in caplog.text
        for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().numpy(), atol=5e-5, err_msg='%s step %d' % (k, step))
