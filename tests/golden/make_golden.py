#!/usr/bin/env python
"""Generate golden fixtures by running the REFERENCE's own model in this container.

Run once in the build container (``python tests/golden/make_golden.py``); the
outputs (``*.npz`` next to this file) are committed, the reference is not.  It
imports ``/root/reference/codes/model.py`` (which only needs torch) by file
path, loads the portable weights-from-seed recipe (oracle/model.py
``seeded_state_dict``) into it and records what the reference computes:

  ref_tiny.npz   hidden 32, 2 layers: logits, eval probs, every intermediate,
                 CTC loss (torch F.ctc_loss on the reference's logits), full
                 gradients of every parameter, BN running stats after the step.
  ref_full.npz   the default 5xBiGRU-800 model, B=2, T_in=301: train logits,
                 eval probs, loss, per-parameter gradient norms + strided samples.
  ref_full_b8 / _b10 / _b32.npz
                 the same model at the batch sizes of BASELINE configs[3] (8 per GPU),
                 configs[1] (10) and configs[2] (32), ragged lengths.  To keep the
                 fixtures small, logits / probs keep every ``tstride``-th output frame;
                 the eval-mode argmax of EVERY frame is kept (uint8) so greedy strings
                 are checked over the whole output.
  ref_ft43_b16.npz
                 the fine-tuned pt_BR head (A=43): the reference model's last Linear
                 swapped as ``codes/utils/training_utils.py:87-122`` does -- rows of the
                 en layer copied through data/map_en-pt_BR.json, the other rows from the
                 seeded recipe -- then one forward/backward of the reference model.

  ref_full_b10_15s / ref_full_b32_10s.npz   (round 4)
                 the lengths the headline metric is quoted on: B=10 with the longest clip 15 s (T_in = 1501 -> T = 746
                 output steps) and B=32 at 10 s (T_in = 1001 -> T = 496; the two-part recurrence forms with bf16 state
                 planes over a long chain), ragged.  ``lnoise`` / ``pnoise`` = how far the reference's own fp32 logits /
                 probabilities are from its float64 twin's.
  ref_sharp_b10.npz
                 the same model with every weight matrix drawn 3x wider (``seeded_state_dict(scale=3.0)``): confident
                 outputs, a recurrence that is much less contractive, T_in = 801.
  ref_traj_b10.npz
  ref_traj_frozen_b10.npz
                 FIVE optimisation steps of the reference model as ``codes/engine.py:45-94`` runs them (loss / B,
                 ``clip_grad_norm_(400)``, SGD lr 3e-4 momentum 0.9 Nesterov -- scripts/librispeech-from_scratch.json) on
                 two alternating B=10 minibatches: per-step loss and gradient norm, the final BatchNorm buffers, strided
                 samples of the final weights and of the momentum buffers.

The GPU box regenerates weights and inputs from the same seeds, so only outputs
are stored.  Nothing here is read at test time except the .npz files.
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.model import seeded_state_dict, swap_fc_en_to_pt_br  # noqa: E402


def load_reference_model_module():
    path = '/root/reference/codes/model.py'
    spec = importlib.util.spec_from_file_location('reference_codes_model', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seeded_inputs(seed, bsz, t_in, nfreq=161, lengths=None):
    """Spectrogram-like inputs: N(0,1) values, frames past each length zeroed (collate padding)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((bsz, t_in, nfreq)).astype(np.float32)
    if lengths is not None:
        for b, n in enumerate(lengths):
            x[b, n:] = 0.0
    return x


def seeded_labels(seed, label_lens, nalpha):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(1, nalpha, size=int(sum(label_lens))).astype(np.int32)


def ragged_lengths(seed, bsz, t_in, lo=0.35):
    """Deterministic ragged frame counts: the first utterance full length (collate pads to the longest)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = sorted((int(v) for v in rng.integers(int(lo * t_in), t_in, size=bsz - 1)), reverse=True)
    return [t_in] + lens


def label_lengths_for(lengths, per_frame=0.09):
    """Transcript lengths that stay feasible for CTC after the conv stack halves the frame rate."""
    return [max(1, int(per_frame * n)) for n in lengths]


def run_case(ref, name, model_kwargs, bsz, t_in, lengths, label_lens, full_grads, tstride=1, finetune43=False,
             f64_truth=False, weight_scale=None, trajectory=0, lnoise=False, frozen_conv=False):
    if trajectory:
        return run_trajectory(ref, name, model_kwargs, bsz, t_in, lengths, label_lens, trajectory, frozen_conv)
    torch.manual_seed(0)
    model = ref.DeepSpeech(**model_kwargs)
    sd = seeded_state_dict(model, seed=1234, scale=weight_scale)
    model.load_state_dict(sd)
    nalpha = model_kwargs.get('num_classes', 29)
    if finetune43:
        old = model.fc[0].module[1]
        new = torch.nn.Linear(old.in_features, 43, bias=False)       # training_utils.py:100-104
        with torch.no_grad():
            new.weight.copy_(torch.from_numpy(swap_fc_en_to_pt_br(old.weight.detach().numpy(), seed=4343)))
        model.fc[0].module[1] = new
        nalpha = 43
    x = torch.from_numpy(seeded_inputs(77, bsz, t_in, lengths=lengths))
    labels = seeded_labels(78, label_lens, nalpha)
    out = {}

    # hooks for intermediates (reference module outputs)
    inter = {}
    hooks = [model.conv[2].register_forward_hook(lambda m, i, o: inter.__setitem__('conv1', o.detach().clone())),
             model.conv[5].register_forward_hook(lambda m, i, o: inter.__setitem__('conv2', o.detach().clone()))]
    for nm, layer in model.rnns.named_children():
        hooks.append(layer.register_forward_hook(
            lambda m, i, o, nm=nm: inter.__setitem__('rnn' + nm, o.detach().clone())))

    model.train()
    logits = model(x)                                   # (B,T,A) train-mode logits
    for h in hooks:
        h.remove()
    t_out = logits.shape[1]
    pct = torch.tensor([n / float(t_in) for n in lengths], dtype=torch.float32)
    out_sizes = (pct * t_out).int()
    acts = logits.transpose(0, 1)                       # (T,B,A) as codes/engine.py:13-15
    loss = F.ctc_loss(acts.log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                      torch.tensor(label_lens, dtype=torch.long), blank=0, reduction='sum')
    total = loss / bsz                                   # codes/engine.py:23,80
    model.zero_grad()
    total.backward()

    out['logits'] = logits.detach().numpy()[:, ::tstride].copy()
    out['tstride'] = np.int32(tstride)
    out['loss_sum'] = np.float32(loss.item())
    out['out_sizes'] = out_sizes.numpy().astype(np.int32)
    out['pct'] = pct.numpy()
    for k, v in inter.items():
        if full_grads:
            out['inter_' + k] = v.numpy()
    for k, p in model.named_parameters():
        g = p.grad.detach().numpy()
        out['gnorm_' + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        if full_grads:
            out['grad_' + k] = g
        else:
            flat = g.reshape(-1)
            stride = max(1, flat.shape[0] // 1024)
            out['gsample_' + k] = flat[::stride][:1024].copy()
    for k, v in model.state_dict().items():
        if 'running' in k:
            out['buf_' + k] = v.numpy().copy()
    if f64_truth:
        # The same step of the reference model in float64: how far the reference's OWN fp32 gradients are from the exact
        # ones (conv2's filter gradient sums ~10^5 products per element: 2.4e-3 of the largest element at B = 32).  The
        # parity tests allow the HIP path that much distance from the fp32 reference, no more.
        model64 = ref.DeepSpeech(**model_kwargs)
        model64.load_state_dict(sd)
        if finetune43:
            model64.fc[0].module[1] = torch.nn.Linear(model.fc[0].module[1].in_features, 43, bias=False)
            model64.fc[0].module[1].load_state_dict(model.fc[0].module[1].state_dict())
        model64 = model64.double().train()
        logits64 = model64(x.double())
        loss64 = F.ctc_loss(logits64.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                            torch.tensor(label_lens, dtype=torch.long), blank=0, reduction='sum')
        (loss64 / bsz).backward()
        for k, p in model64.named_parameters():
            flat = p.grad.numpy().reshape(-1)
            stride = max(1, flat.shape[0] // 1024)
            out['gnoise_' + k] = np.float32(np.abs(flat[::stride][:1024] - out['gsample_' + k]).max())
        if lnoise:
            out['lnoise'] = np.float32(np.abs(logits64.detach().numpy() - logits.detach().numpy()).max())
            out['loss64'] = np.float64(loss64.item())
            model64.eval()
            with torch.no_grad():
                probs64 = model64(x.double()).numpy()

    model.eval()
    with torch.no_grad():
        probs = model(x)
    out['probs'] = probs.numpy()[:, ::tstride].copy()
    if f64_truth and lnoise:
        out['pnoise'] = np.float32(np.abs(probs64 - probs.numpy()).max())
    if tstride > 1:
        # every frame: best and runner-up class plus whether they are closer than 1e-4 (with near-uniform outputs of
        # random weights some frames are ties at fp32 round-off; there the runner-up is an equally valid argmax)
        order = np.argsort(-probs.numpy(), axis=-1, kind='stable')
        top = np.take_along_axis(probs.numpy(), order[..., :2], -1)
        out['argmax'] = order[..., 0].astype(np.uint8)
        out['argmax2'] = order[..., 1].astype(np.uint8)
        out['near_tie'] = ((top[..., 0] - top[..., 1]) < 1e-4)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'logits', out['logits'].shape, 'loss', out['loss_sum'],
          'size %.1f KB' % (os.path.getsize(os.path.join(HERE, name)) / 1024.0))


def traj_batches(bsz, t_in, lengths, label_lens, nalpha=29):
    """The two minibatches the trajectory alternates between (the second: other values, the lengths reversed in time order
    of the draw so that the BatchNorm statistics differ)."""
    a = (seeded_inputs(77, bsz, t_in, lengths=lengths), seeded_labels(78, label_lens, nalpha))
    b = (seeded_inputs(79, bsz, t_in, lengths=lengths), seeded_labels(80, label_lens, nalpha))
    return [a, b]


TRAJ_OPT = dict(lr=3e-4, momentum=0.9, nesterov=True)      # scripts/librispeech-from_scratch.json
TRAJ_MAX_NORM = 400.0


def run_trajectory(ref, name, model_kwargs, bsz, t_in, lengths, label_lens, steps, frozen_conv=False):
    """``steps`` updates of the reference model exactly as codes/engine.py:45-94 does them (warp-ctc's place taken by
    F.ctc_loss on log_softmax, reduction sum -- SURVEY 8c).  ``frozen_conv``: after ``_freeze_layers(model, ['conv'])``
    (codes/utils/training_utils.py:57-84, restated: its module cannot be imported here): the conv block's BatchNorm modules
    put in eval mode once, its parameters ``requires_grad = False`` -- and ``model.train()`` at the top of every step
    (codes/engine.py:51) as below, which returns that BatchNorm to training mode."""
    torch.manual_seed(0)
    model = ref.DeepSpeech(**model_kwargs)
    model.load_state_dict(seeded_state_dict(model, seed=1234))
    if frozen_conv:
        for m in model.conv.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.eval()
        for p in model.conv.parameters():
            p.requires_grad = False
    opt = torch.optim.SGD(model.parameters(), **TRAJ_OPT)
    batches = traj_batches(bsz, t_in, lengths, label_lens)
    pct = torch.tensor([n / float(t_in) for n in lengths], dtype=torch.float32)
    out = {'pct': pct.numpy(), 'losses': [], 'gnorms': []}
    for i in range(steps):
        x, labels = batches[i % 2]
        model.train()
        logits = model(torch.from_numpy(x))
        out_sizes = (pct * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                          torch.tensor(label_lens, dtype=torch.long), blank=0, reduction='sum') / bsz
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), TRAJ_MAX_NORM)
        opt.step()
        out['losses'].append(loss.item())
        out['gnorms'].append(float(gn))
        print('  step', i, 'loss', loss.item(), 'gnorm', float(gn))
    out['losses'] = np.asarray(out['losses'], np.float64)
    out['gnorms'] = np.asarray(out['gnorms'], np.float64)
    out['out_sizes'] = out_sizes.numpy().astype(np.int32)
    for k, p in model.named_parameters():
        flat = p.detach().numpy().reshape(-1)
        stride = max(1, flat.shape[0] // 1024)
        out['wsample_' + k] = flat[::stride][:1024].copy()
        out['wnorm_' + k] = np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum()))
        mom = opt.state[p]['momentum_buffer'].numpy().reshape(-1) if 'momentum_buffer' in opt.state[p] else np.zeros_like(flat)
        out['msample_' + k] = mom[::stride][:1024].copy()                 # (a frozen parameter has no buffer: zeros)
    for k, v in model.state_dict().items():
        if 'running' in k:
            out['buf_' + k] = v.numpy().copy()
    model.eval()
    with torch.no_grad():
        probs = model(torch.from_numpy(batches[0][0])).numpy()
    out['probs'] = probs[:, ::2].copy()
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'losses', out['losses'], 'size %.1f KB' % (os.path.getsize(os.path.join(HERE, name)) / 1024.0))


def cases():
    yield 'ref_tiny.npz', dict(model_kwargs=dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29), bsz=3, t_in=121,
                               lengths=[121, 97, 64], label_lens=[9, 6, 4], full_grads=True)
    yield 'ref_full.npz', dict(model_kwargs=dict(), bsz=2, t_in=301, lengths=[301, 233], label_lens=[30, 21],
                               full_grads=False)
    for bsz, t_in, tstride in ((8, 301, 2), (10, 301, 2), (32, 301, 6)):
        lens = ragged_lengths(500 + bsz, bsz, t_in)
        yield 'ref_full_b%d.npz' % bsz, dict(model_kwargs=dict(), bsz=bsz, t_in=t_in, lengths=lens,
                                             label_lens=label_lengths_for(lens), full_grads=False, tstride=tstride,
                                             f64_truth=True)
    lens = ragged_lengths(516, 16, 261)
    yield 'ref_ft43_b16.npz', dict(model_kwargs=dict(), bsz=16, t_in=261, lengths=lens,
                                   label_lens=label_lengths_for(lens), full_grads=False, tstride=4, finetune43=True,
                                   f64_truth=True)


    # round 4: the lengths the metric is quoted on, a sharp-weight model, and a 5-step trajectory
    lens = ragged_lengths(1510, 10, 1501)
    yield 'ref_full_b10_15s.npz', dict(model_kwargs=dict(), bsz=10, t_in=1501, lengths=lens,
                                       label_lens=label_lengths_for(lens), full_grads=False, tstride=8, f64_truth=True,
                                       lnoise=True)
    lens = ragged_lengths(1032, 32, 1001)
    yield 'ref_full_b32_10s.npz', dict(model_kwargs=dict(), bsz=32, t_in=1001, lengths=lens,
                                       label_lens=label_lengths_for(lens), full_grads=False, tstride=16, f64_truth=True,
                                       lnoise=True)
    lens = ragged_lengths(810, 10, 801)
    yield 'ref_sharp_b10.npz', dict(model_kwargs=dict(), bsz=10, t_in=801, lengths=lens,
                                    label_lens=label_lengths_for(lens), full_grads=False, tstride=4, f64_truth=True,
                                    weight_scale=3.0, lnoise=True)
    lens = ragged_lengths(510, 10, 301)
    yield 'ref_traj_b10.npz', dict(model_kwargs=dict(), bsz=10, t_in=301, lengths=lens,
                                   label_lens=label_lengths_for(lens), full_grads=False, trajectory=5)
    # round 5: the same five steps with the conv block frozen (scripts/pt_BR-finetune-freeze.json): the conv block's hard clip
    # is the model's only non-smooth function and its gradient mask only reaches the conv filters, so with those frozen the
    # trajectories of two fp32 implementations stay within round-off of each other -- the tight form of the trajectory test
    yield 'ref_traj_frozen_b10.npz', dict(model_kwargs=dict(), bsz=10, t_in=301, lengths=lens,
                                          label_lens=label_lengths_for(lens), full_grads=False, trajectory=5, frozen_conv=True)


CASES = dict(cases())


def expected_keys(name):
    """The keys ``run_case`` writes for fixture ``name``, derived without running the reference (the oracle model has the
    reference's parameter and buffer names: tests/test_oracle_model.py).  tests/test_oracle_misc.py checks every committed
    fixture against this, so a fixture that was not regenerated after the generator changed cannot go unnoticed."""
    from oracle.model import OracleDeepSpeech
    kw = CASES[name]
    model = OracleDeepSpeech(**kw['model_kwargs'])
    params = [k for k, _ in model.named_parameters()]
    if kw.get('trajectory'):
        keys = {'pct', 'losses', 'gnorms', 'out_sizes', 'probs'}
        keys |= {pre + k for k in params for pre in ('wsample_', 'wnorm_', 'msample_')}
        return keys | {'buf_' + k for k in model.state_dict() if 'running' in k}
    keys = {'logits', 'tstride', 'loss_sum', 'out_sizes', 'pct', 'probs'}
    keys |= {'gnorm_' + k for k in params}
    keys |= {('grad_' if kw['full_grads'] else 'gsample_') + k for k in params}
    keys |= {'buf_' + k for k in model.state_dict() if 'running' in k}
    if kw['full_grads']:
        keys |= {'inter_conv1', 'inter_conv2'} | {'inter_rnn%d' % i for i in range(kw['model_kwargs'].get('num_rnn_layers', 5))}
    if kw.get('f64_truth'):
        keys |= {'gnoise_' + k for k in params}
        if kw.get('lnoise'):
            keys |= {'lnoise', 'pnoise', 'loss64'}
    if kw.get('tstride', 1) > 1:
        keys |= {'argmax', 'argmax2', 'near_tie'}
    return keys


def main():
    """``make_golden.py`` regenerates everything; ``make_golden.py ref_full_b8.npz ...`` only the named fixtures."""
    ref = load_reference_model_module()
    for name, kw in CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        kw = dict(kw)
        run_case(ref, name, kw.pop('model_kwargs'), **kw)


if __name__ == '__main__':
    main()
