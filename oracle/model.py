"""Oracle: DeepSpeech2 conv -> BiGRU -> FC on stock PyTorch CPU ops.

TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Restates ``DeepSpeech`` of the reference (``codes/model.py:116-207``) with the
same state-dict key names so weights interchange by name
(``codes/utils/model_utils.py:70``).  What is restated, by line:

  conv stack        codes/model.py:142-145  Conv2d(1,32,(41,11),(2,2),(0,10)) BN Hardtanh(0,20)
                                            Conv2d(32,32,(21,11),(2,1)) BN Hardtanh(0,20)
  rnn input size    codes/model.py:148-151
  BatchRNN          codes/model.py:43-69    [BN1d over T*B rows] -> GRU(bias=False, bidir) -> sum dirs
  SequenceWise      codes/model.py:27-34    (T,B,F) -> (T*B,F) -> module -> (T,B,.)
  fc                codes/model.py:177-180  BN1d -> Linear(bias=False)
  forward           codes/model.py:182-207  no sequence packing; eval -> softmax

This file is pinned against the reference itself: tests/golden/make_golden.py
imports /root/reference/codes/model.py in the build container, loads identical
weights into both and commits the reference's outputs as fixtures.

Besides the nn.Module there is ``gru_direction_explicit`` -- the GRU recurrence
written out gate by gate (torch.nn.GRU documentation formulae) -- used to check
the tensors the HIP kernels save for their backward pass.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def conv_out_time(t_in):
    """Output frames for T_in spectrogram frames (conv1 pad 10/k 11/s 2, conv2 k 11)."""
    t1 = (t_in + 2 * 10 - 11) // 2 + 1
    return t1 - 10


def rnn_input_features(window_size=320):
    f = window_size // 2 + 1
    f = (f - 41) // 2 + 1
    f = (f - 21) // 2 + 1
    return 32 * f


class _RowWise(nn.Module):
    """(T,B,F) -> apply ``module`` on (T*B,F) rows -> (T,B,.)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        t, b = x.shape[0], x.shape[1]
        return self.module(x.reshape(t * b, -1)).reshape(t, b, -1)


class _BiGRULayer(nn.Module):
    def __init__(self, n_in, n_hidden, with_bn):
        super().__init__()
        self.batch_norm = _RowWise(nn.BatchNorm1d(n_in)) if with_bn else None
        self.rnn = nn.GRU(input_size=n_in, hidden_size=n_hidden, bidirectional=True, bias=False)

    def forward(self, x):
        if self.batch_norm is not None:
            x = self.batch_norm(x)
        y, _ = self.rnn(x)
        h = y.shape[2] // 2
        return y[:, :, :h] + y[:, :, h:]


class OracleDeepSpeech(nn.Module):
    """Bidirectional-GRU DeepSpeech2 with the reference's parameter names."""

    def __init__(self, num_classes=29, rnn_hidden_size=800, num_rnn_layers=5, window_size=320):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=(41, 11), stride=(2, 2), padding=(0, 10)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20),
            nn.Conv2d(32, 32, kernel_size=(21, 11), stride=(2, 1)),
            nn.BatchNorm2d(32),
            nn.Hardtanh(0, 20))
        n_in = rnn_input_features(window_size)
        layers = []
        for i in range(num_rnn_layers):
            layers.append((str(i), _BiGRULayer(n_in if i == 0 else rnn_hidden_size, rnn_hidden_size, i > 0)))
        self.rnns = nn.Sequential(OrderedDict(layers))
        self.fc = nn.Sequential(_RowWise(nn.Sequential(
            nn.BatchNorm1d(rnn_hidden_size), nn.Linear(rnn_hidden_size, num_classes, bias=False))))

    def forward(self, x, return_intermediates=False):
        inter = OrderedDict()
        y = x.unsqueeze(1).transpose(2, 3).contiguous()          # (B,1,F,T_in)
        y = self.conv[2](self.conv[1](self.conv[0](y)))
        inter['conv1'] = y
        y = self.conv[5](self.conv[4](self.conv[3](y)))
        inter['conv2'] = y
        b, c, d, t = y.shape
        y = y.reshape(b, c * d, t).permute(2, 0, 1).contiguous()  # (T,B,C*D)
        for name, layer in self.rnns.named_children():
            y = layer(y)
            inter['rnn' + name] = y
        y = self.fc(y).transpose(0, 1)                            # (B,T,A)
        inter['logits'] = y
        out = y if self.training else F.softmax(y, dim=-1)
        return (out, inter) if return_intermediates else out


def seeded_state_dict(model_or_shapes, seed, scale=None):
    """Portable weights-from-seed recipe.

    Fills every floating tensor of the state dict, in state-dict order, from
    numpy PCG64(seed): parameters uniform(-k, k) with k = 1/sqrt(fan) (fan = the
    product of all but the first dim, min 1), BN weights uniform(0.5, 1.5),
    running_var uniform(0.5, 1.5), running_mean uniform(-0.5, 0.5).  The same
    function runs on the GPU box, so no weight file has to be committed.
    """
    sd = model_or_shapes.state_dict() if hasattr(model_or_shapes, 'state_dict') else model_or_shapes
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for key, val in sd.items():
        shape = tuple(val.shape)
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros(shape, dtype=torch.long)
            continue
        if key.endswith('running_var'):
            a = rng.uniform(0.5, 1.5, size=shape)
        elif key.endswith('running_mean'):
            a = rng.uniform(-0.5, 0.5, size=shape)
        elif len(shape) == 1 and key.endswith('weight'):
            a = rng.uniform(0.5, 1.5, size=shape)           # BN gamma
        elif len(shape) == 1:
            a = rng.uniform(-0.1, 0.1, size=shape)          # biases / BN beta
        else:
            fan = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            k = (scale if scale is not None else 1.0) / math.sqrt(max(fan, 1))
            a = rng.uniform(-k, k, size=shape)
        out[key] = torch.from_numpy(a.astype(np.float32))
    return out


def swap_fc_en_to_pt_br(linear_weight_en, seed, num_classes=43):
    """The FC surgery of ``finetune_model`` (reference codes/utils/training_utils.py:96-120) as a pure function of the
    old weight: rows listed in data/map_en-pt_BR.json are copied, the others come from N(0, 0.01) drawn with the
    portable generator (the reference draws them with torch's global RNG)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = json.load(open(os.path.join(root, 'data', 'map_en-pt_BR.json')))
    old_idx, new_idx = zip(*pairs)
    rng = np.random.Generator(np.random.PCG64(seed))
    w = (0.01 * rng.standard_normal((num_classes, linear_weight_en.shape[1]))).astype(np.float32)
    w[list(new_idx)] = linear_weight_en[list(old_idx)]
    return w


def gru_direction_explicit(x, w_ih, w_hh, reverse=False):
    """One GRU direction written out (bias-free), returning every saved tensor.

    x (T,B,In); w_ih (3H,In); w_hh (3H,H); gate order r,z,n as in torch.nn.GRU:
      r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z)
      n = tanh(gi_n + r * gh_n); h' = (1 - z) * n + z * h
    Returns dict(h (T,B,H), r, z, n, ghn).
    """
    t_len, b, _ = x.shape
    hdim = w_hh.shape[1]
    gi = x.reshape(t_len * b, -1) @ w_ih.t()
    gi = gi.reshape(t_len, b, 3 * hdim)
    h = torch.zeros(b, hdim, dtype=x.dtype)
    keys = ('h', 'r', 'z', 'n', 'ghn')
    saved = {k: torch.zeros(t_len, b, hdim, dtype=x.dtype) for k in keys}
    order = range(t_len - 1, -1, -1) if reverse else range(t_len)
    for t in order:
        gh = h @ w_hh.t()
        r = torch.sigmoid(gi[t, :, :hdim] + gh[:, :hdim])
        z = torch.sigmoid(gi[t, :, hdim:2 * hdim] + gh[:, hdim:2 * hdim])
        ghn = gh[:, 2 * hdim:]
        n = torch.tanh(gi[t, :, 2 * hdim:] + r * ghn)
        h = (1.0 - z) * n + z * h
        for k, v in zip(keys, (h, r, z, n, ghn)):
            saved[k][t] = v
    return saved
