"""The oracle's model restatement against the REFERENCE's own outputs (golden fixtures)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle.model import OracleDeepSpeech, conv_out_time, gru_direction_explicit, seeded_state_dict
from tests.golden.make_golden import seeded_inputs, seeded_labels


def _run(golden_dir, name, kwargs, bsz, t_in, lengths, label_lens):
    g = np.load(os.path.join(golden_dir, name))
    model = OracleDeepSpeech(**kwargs)
    model.load_state_dict(seeded_state_dict(model, 1234))
    x = torch.from_numpy(seeded_inputs(77, bsz, t_in, lengths=lengths))
    model.train()
    logits, inter = model(x, return_intermediates=True)
    assert logits.shape[1] == conv_out_time(t_in)
    np.testing.assert_allclose(logits.detach().numpy(), g['logits'], rtol=0, atol=2e-5)
    labels = seeded_labels(78, label_lens, 29)
    loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(),
                      torch.from_numpy(g['out_sizes']).long(), torch.tensor(label_lens), blank=0, reduction='sum')
    assert abs(loss.item() - float(g['loss_sum'])) <= 1e-4 * abs(float(g['loss_sum']))
    (loss / bsz).backward()
    for k, p in model.named_parameters():
        gn = np.sqrt((p.grad.numpy().astype(np.float64) ** 2).sum())
        assert abs(gn - float(g['gnorm_' + k])) <= 1e-4 * float(g['gnorm_' + k]) + 1e-7, k
        if 'grad_' + k in g:
            np.testing.assert_allclose(p.grad.numpy(), g['grad_' + k], rtol=1e-3, atol=1e-5)
    for k, v in model.state_dict().items():
        if 'running' in k:
            np.testing.assert_allclose(v.numpy(), g['buf_' + k], rtol=1e-5, atol=1e-6)
    model.eval()
    with torch.no_grad():
        probs = model(x)
    np.testing.assert_allclose(probs.numpy(), g['probs'], rtol=0, atol=1e-5)
    return g, inter


def test_tiny_model_matches_reference(golden_dir):
    g, inter = _run(golden_dir, 'ref_tiny.npz', dict(rnn_hidden_size=32, num_rnn_layers=2), 3, 121,
                    [121, 97, 64], [9, 6, 4])
    for k in ('conv1', 'conv2', 'rnn0', 'rnn1'):
        np.testing.assert_allclose(inter[k].detach().numpy(), g['inter_' + k], rtol=0, atol=2e-5)


def test_full_model_matches_reference(golden_dir):
    _run(golden_dir, 'ref_full.npz', dict(), 2, 301, [301, 233], [30, 21])


def test_state_dict_keys_are_the_reference_names(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ref_full.npz'))
    ref_params = sorted(k[len('gnorm_'):] for k in g.files if k.startswith('gnorm_'))
    model = OracleDeepSpeech()
    assert sorted(k for k, _ in model.named_parameters()) == ref_params
    assert sum(p.numel() for p in model.parameters()) == 38067968


def test_explicit_gru_equals_nn_gru():
    torch.manual_seed(3)
    t, b, n_in, h = 7, 3, 5, 4
    gru = torch.nn.GRU(n_in, h, bidirectional=True, bias=False)
    x = torch.randn(t, b, n_in)
    y, _ = gru(x)
    f = gru_direction_explicit(x, gru.weight_ih_l0, gru.weight_hh_l0)
    r = gru_direction_explicit(x, gru.weight_ih_l0_reverse, gru.weight_hh_l0_reverse, reverse=True)
    np.testing.assert_allclose(f['h'].detach().numpy(), y[:, :, :h].detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(r['h'].detach().numpy(), y[:, :, h:].detach().numpy(), atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------
# full-size fixtures at the batch sizes of BASELINE configs[1..4] (tests/golden_cases.py)
def _oracle_case(golden_dir, name):
    import pytest  # noqa: F401
    from oracle.model import swap_fc_en_to_pt_br
    from tests.golden_cases import FT43_SEED, case_inputs, check_against_golden
    g = np.load(os.path.join(golden_dir, name))
    kw, x, labels, nalpha = case_inputs(name)
    model = OracleDeepSpeech()
    model.load_state_dict(seeded_state_dict(model, 1234, scale=kw.get('weight_scale')))
    if kw.get('finetune43'):
        head = model.fc[0].module
        new = torch.nn.Linear(800, 43, bias=False)
        with torch.no_grad():
            new.weight.copy_(torch.from_numpy(swap_fc_en_to_pt_br(head[1].weight.detach().numpy(), FT43_SEED)))
        head[1] = new
    model.train()
    logits = model(torch.from_numpy(x))
    loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(),
                      torch.from_numpy(g['out_sizes']).long(), torch.tensor(kw['label_lens']), blank=0, reduction='sum')
    (loss / kw['bsz']).backward()
    grads = {k: p.grad.numpy() for k, p in model.named_parameters()}
    bufs = {k: v.numpy().copy() for k, v in model.state_dict().items() if 'running' in k}
    model.eval()
    with torch.no_grad():
        probs = model(torch.from_numpy(x))
    check_against_golden(g, logits.detach().numpy(), float(loss.item()), grads, bufs, probs.numpy(), logit_tol=2e-5,
                         prob_tol=1e-5, gnorm_rtol=1e-4, gsample_rtol=1e-3, buf_rtol=1e-5, buf_atol=1e-6)


def test_full_model_b8_matches_reference(golden_dir):
    _oracle_case(golden_dir, 'ref_full_b8.npz')


def test_finetuned_pt_br_head_matches_reference(golden_dir):
    _oracle_case(golden_dir, 'ref_ft43_b16.npz')


def test_sharp_weight_model_matches_reference(golden_dir):
    """3x wider weights (confident outputs, a much less contractive recurrence), T_in = 801: the oracle is the reference here
    too -- bench.py's parity block and smoke() compare the HIP path with THIS model."""
    _oracle_case(golden_dir, 'ref_sharp_b10.npz')


def test_frozen_conv_trajectory_matches_reference(golden_dir):
    """Five optimisation steps with the conv block frozen (``ref_traj_frozen_b10.npz``: the reference model under
    ``torch.optim.SGD`` + ``clip_grad_norm_``, codes/engine.py:45-94 after ``_freeze_layers(model, ['conv'])``), on the
    oracle's stock torch modules: eval() on the frozen block's BatchNorm at set-up, ``model.train()`` at the top of every
    step -- the reference's literal sequence, under which that BatchNorm trains (tests/test_host2_cpu.py)."""
    from tests.golden.make_golden import CASES, TRAJ_MAX_NORM, TRAJ_OPT, traj_batches
    g = np.load(os.path.join(golden_dir, 'ref_traj_frozen_b10.npz'))
    kw = CASES['ref_traj_frozen_b10.npz']
    model = OracleDeepSpeech()
    model.load_state_dict(seeded_state_dict(model, 1234))
    for m in model.conv.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    for p in model.conv.parameters():
        p.requires_grad = False
    opt = torch.optim.SGD(model.parameters(), **TRAJ_OPT)
    batches = traj_batches(kw['bsz'], kw['t_in'], kw['lengths'], kw['label_lens'])
    pct = torch.from_numpy(g['pct'])
    losses, gnorms = [], []
    for i in range(kw['trajectory']):
        x, labels = batches[i % 2]
        model.train()
        logits = model(torch.from_numpy(x))
        out_sizes = (pct * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                          torch.tensor(kw['label_lens'], dtype=torch.long), blank=0, reduction='sum') / kw['bsz']
        opt.zero_grad()
        loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), TRAJ_MAX_NORM)))
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-5)
    np.testing.assert_allclose(gnorms, g['gnorms'], rtol=1e-4)
    for k, p in model.named_parameters():
        flat = p.detach().numpy().reshape(-1)
        stride = max(1, flat.shape[0] // 1024)
        np.testing.assert_allclose(flat[::stride][:1024], g['wsample_' + k], rtol=1e-4, atol=1e-7, err_msg=k)
    assert int(model.conv[1].num_batches_tracked) == 5
    for k, v in model.state_dict().items():
        if 'running' in k:
            np.testing.assert_allclose(v.numpy(), g['buf_' + k], rtol=1e-4, atol=1e-6, err_msg=k)
