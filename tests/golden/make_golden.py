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

from oracle.model import seeded_state_dict  # noqa: E402


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


def run_case(ref, name, model_kwargs, bsz, t_in, lengths, label_lens, full_grads):
    torch.manual_seed(0)
    model = ref.DeepSpeech(**model_kwargs)
    sd = seeded_state_dict(model, seed=1234)
    model.load_state_dict(sd)
    nalpha = model_kwargs.get('num_classes', 29)
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

    out['logits'] = logits.detach().numpy()
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

    model.eval()
    with torch.no_grad():
        probs = model(x)
    out['probs'] = probs.numpy()
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, 'logits', out['logits'].shape, 'loss', out['loss_sum'],
          'size %.1f KB' % (os.path.getsize(os.path.join(HERE, name)) / 1024.0))


def main():
    ref = load_reference_model_module()
    run_case(ref, 'ref_tiny.npz', dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29),
             bsz=3, t_in=121, lengths=[121, 97, 64], label_lens=[9, 6, 4], full_grads=True)
    run_case(ref, 'ref_full.npz', dict(), bsz=2, t_in=301, lengths=[301, 233], label_lens=[30, 21],
             full_grads=False)


if __name__ == '__main__':
    main()
