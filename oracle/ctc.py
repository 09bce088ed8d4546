"""Oracle: CTC loss and gradient (TEST INFRASTRUCTURE ONLY).

The reference calls ``warpctc_pytorch.CTCLoss()(acts, labels, act_lens,
label_lens)`` (``train.py:12,179``; ``codes/engine.py:22``;
``codes/metrics.py:43,51``).  warp-ctc is an external, un-pinned dependency
(SeanNaren fork cloned at HEAD, ``docker/Dockerfile:54-66``) that is absent from
the reference tree, so this file restates the *published* algorithm (Graves et
al. 2006, as implemented by warp-ctc):

  * acts are UN-normalised (T,B,A) activations; a softmax over A is applied
    inside the loss (reference README.md:168);
  * blank index 0 (``codes/utils/io_utils.py:3,17-25``);
  * labels are a flat 1-D int array of sum(label_lens) entries;
  * the result is the SUM over the batch of -log p(labels | acts) and the
    gradient with respect to the un-normalised acts:
        grad[t,b,k] = softmax[t,b,k] - (1/p_b) * sum_{s: l'_s = k} alpha_t(s) beta_t(s) / y_t(k)
    with zero gradient for frames t >= act_lens[b];
  * an infeasible alignment (label_len + repeats > act_len) gives p = 0: this
    restatement returns cost +inf there like torch's ctc_loss; the reference
    guards the training loop against it at ``codes/engine.py:27-30``.

Pins: ``ctc_brute_force`` enumerates every alignment in float64 for tiny cases;
``ctc_torch`` wraps torch.nn.functional.ctc_loss.  Parity with warp-ctc's own
binary is "unpinned" (it cannot be built here).
"""
import itertools

import numpy as np


def _log_softmax(a):
    m = a.max(axis=-1, keepdims=True)
    e = a - m
    return e - np.log(np.exp(e).sum(axis=-1, keepdims=True))


def ctc_loss_and_grad(acts, labels, act_lens, label_lens, blank=0):
    """float64 alpha/beta CTC.  Returns (costs (B,), grad (T,B,A))."""
    acts = np.asarray(acts, dtype=np.float64)
    t_max, bsz, nalpha = acts.shape
    logp = _log_softmax(acts)
    grad = np.zeros_like(acts)
    costs = np.zeros(bsz, dtype=np.float64)
    off = 0
    neg_inf = -np.inf
    for b in range(bsz):
        tl = int(act_lens[b])
        ll = int(label_lens[b])
        lab = np.asarray(labels[off:off + ll], dtype=np.int64)
        off += ll
        s_len = 2 * ll + 1
        ext = np.full(s_len, blank, dtype=np.int64)
        ext[1::2] = lab
        lp = logp[:tl, b, :]
        alpha = np.full((tl, s_len), neg_inf)
        beta = np.full((tl, s_len), neg_inf)
        if tl == 0:
            costs[b] = 0.0 if ll == 0 else np.inf
            continue
        alpha[0, 0] = lp[0, blank]
        if s_len > 1:
            alpha[0, 1] = lp[0, ext[1]]
        for t in range(1, tl):
            for s in range(s_len):
                terms = [alpha[t - 1, s]]
                if s >= 1:
                    terms.append(alpha[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    terms.append(alpha[t - 1, s - 2])
                m = max(terms)
                if m > neg_inf:
                    alpha[t, s] = m + np.log(sum(np.exp(x - m) for x in terms)) + lp[t, ext[s]]
        beta[tl - 1, s_len - 1] = lp[tl - 1, blank]
        if s_len > 1:
            beta[tl - 1, s_len - 2] = lp[tl - 1, ext[s_len - 2]]
        for t in range(tl - 2, -1, -1):
            for s in range(s_len):
                terms = [beta[t + 1, s]]
                if s + 1 < s_len:
                    terms.append(beta[t + 1, s + 1])
                if s + 2 < s_len and ext[s] != blank and ext[s] != ext[s + 2]:
                    terms.append(beta[t + 1, s + 2])
                m = max(terms)
                if m > neg_inf:
                    beta[t, s] = m + np.log(sum(np.exp(x - m) for x in terms)) + lp[t, ext[s]]
        tail = [alpha[tl - 1, s_len - 1]]
        if s_len > 1:
            tail.append(alpha[tl - 1, s_len - 2])
        m = max(tail)
        ll_total = m + np.log(sum(np.exp(x - m) for x in tail)) if m > neg_inf else neg_inf
        costs[b] = -ll_total
        if ll_total == neg_inf:
            continue  # p = 0: gradient left at zero (loss is inf; the caller zeroes it)
        y = np.exp(lp)
        occ = np.zeros((tl, nalpha))
        ab = alpha + beta
        for s in range(s_len):
            finite = ab[:, s] > neg_inf
            occ[finite, ext[s]] += np.exp(ab[finite, s] - ll_total - lp[finite, ext[s]])
        grad[:tl, b, :] = y - occ
    return costs, grad


def ctc_brute_force(acts_tb, label, blank=0):
    """-log p(label | acts) by enumerating every length-T path (tiny T, A only)."""
    acts_tb = np.asarray(acts_tb, dtype=np.float64)
    t_len, nalpha = acts_tb.shape
    p = np.exp(_log_softmax(acts_tb))
    total = 0.0
    label = list(label)
    for path in itertools.product(range(nalpha), repeat=t_len):
        collapsed = []
        prev = None
        for c in path:
            if c != prev and c != blank:
                collapsed.append(c)
            prev = c
        if collapsed == label:
            total += np.prod([p[t, c] for t, c in enumerate(path)])
    return -np.log(total) if total > 0 else np.inf


def ctc_torch(acts, labels, act_lens, label_lens, blank=0):
    """Sum-reduced CTC on un-normalised acts via torch CPU; returns (loss, grad)."""
    import torch
    import torch.nn.functional as F
    a = torch.as_tensor(np.asarray(acts), dtype=torch.float32).clone().requires_grad_(True)
    loss = F.ctc_loss(a.log_softmax(-1), torch.as_tensor(np.asarray(labels), dtype=torch.long),
                      torch.as_tensor(np.asarray(act_lens), dtype=torch.long),
                      torch.as_tensor(np.asarray(label_lens), dtype=torch.long),
                      blank=blank, reduction='sum', zero_infinity=False)
    loss.backward()
    return float(loss.item()), a.grad.numpy()
