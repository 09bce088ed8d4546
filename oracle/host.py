"""Oracle: the Python rows of the hot path (TEST INFRASTRUCTURE ONLY).

Restates, for checking the product's host-side code:

  collate            codes/data.py:107-164   zero-pad to longest, input_percentages, flat labels
  out_sizes          codes/engine.py:12-16   (input_percentages * T).int()  -- float32 multiply, truncation
  sanitize_loss      codes/engine.py:19-32   loss / B, sum, +-inf -> 0
  clip + SGD         codes/engine.py:87-90   clip_grad_norm_(400) then SGD(momentum, nesterov)
  greedy decode      codes/decoder.py:123-160 argmax; for t < size: drop blank; drop c[t]==c[t-1]
  CER / WER          codes/decoder.py:49-78, codes/metrics.py:114-132,143-162, test.py:98-104
  DDP bin partition  codes/sampler.py:113-125

These reference files cannot be imported here (ignite, Levenshtein, removed
sklearn module), so the pins are hand-computed known answers in
tests/test_oracle_host.py.
"""
import math

import numpy as np


def collate(samples):
    """samples: list of (spect (T_i,F) float32, labels list[int]).

    Returns inputs (B,T_max,F) f32, targets (sum L) i32, input_percentages (B) f32,
    target_sizes (B) i32 -- codes/data.py:132-158.
    """
    bsz = len(samples)
    tmax = max(s[0].shape[0] for s in samples)
    nfreq = samples[0][0].shape[1]
    inputs = np.zeros((bsz, tmax, nfreq), dtype=np.float32)
    pct = np.zeros(bsz, dtype=np.float32)
    sizes = np.zeros(bsz, dtype=np.int32)
    flat = []
    for i, (spect, lab) in enumerate(samples):
        n = spect.shape[0]
        inputs[i, :n] = spect
        pct[i] = np.float32(n / float(tmax))       # python double division, stored as float32
        sizes[i] = len(lab)
        flat.extend(int(v) for v in lab)
    return inputs, np.asarray(flat, dtype=np.int32), pct, sizes


def out_sizes(input_percentages, seq_length):
    """(pct * T).int(): float32 product truncated toward zero -- codes/engine.py:16."""
    pct = np.asarray(input_percentages, dtype=np.float32)
    return (pct * np.float32(seq_length)).astype(np.int32)


def sanitize_loss(cost_sum, batch_size):
    """loss / B, +-inf -> 0 -- codes/engine.py:23-30."""
    v = np.float32(cost_sum) / np.float32(batch_size)
    if np.isinf(v):
        return np.float32(0.0), True
    return np.float32(v), False


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): returns (total_norm, scaled grads)."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    if coef < 1.0:
        grads = [(g * np.float32(coef)).astype(np.float32) for g in grads]
    return total, grads


def sgd_nesterov(params, grads, bufs, lr, momentum, first_step):
    """torch.optim.SGD(momentum, nesterov=True, dampening=0, weight_decay=0).

    buf = g on the first step, else momentum*buf + g; p -= lr * (g + momentum*buf).
    """
    new_p, new_b = [], []
    for p, g, b in zip(params, grads, bufs):
        b = g.copy() if first_step else (np.float32(momentum) * b + g)
        d = g + np.float32(momentum) * b
        new_p.append((p - np.float32(lr) * d).astype(np.float32))
        new_b.append(b.astype(np.float32))
    return new_p, new_b


def greedy_decode(probs, sizes, labels, blank=0):
    """probs (B,T,A) -> (strings, offsets) -- codes/decoder.py:123-160.

    argmax ties resolve to the first index (torch.max).  A symbol is dropped when
    it equals the previous FRAME's symbol, so 'a _ a' keeps both a's and 'a a'
    keeps one.
    """
    probs = np.asarray(probs)
    best = probs.argmax(axis=2)
    strings, offsets = [], []
    for b in range(probs.shape[0]):
        n = int(sizes[b]) if sizes is not None else probs.shape[1]
        chars, offs = [], []
        for t in range(n):
            c = int(best[b, t])
            if c == blank:
                continue
            if t != 0 and c == int(best[b, t - 1]):
                continue
            chars.append(labels[c])
            offs.append(t)
        strings.append(''.join(chars))
        offsets.append(np.asarray(offs, dtype=np.int32))
    return strings, offsets


def labels_to_string(ids, labels):
    """convert_to_strings without repetition removal -- codes/decoder.py:99-121 (targets)."""
    return ''.join(labels[int(i)] for i in ids if int(i) != 0)


def levenshtein(a, b):
    """Unit-cost edit distance (python-Levenshtein's ``distance``)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cer_distance(hyp, ref):
    """Decoder.cer: spaces removed, edit distance -- codes/decoder.py:69-78."""
    return levenshtein(hyp.replace(' ', ''), ref.replace(' ', ''))


def wer_distance(hyp, ref):
    """Decoder.wer: word-level edit distance -- codes/decoder.py:49-67."""
    return levenshtein(hyp.split(), ref.split())


def corpus_cer_wer(hyps, refs):
    """test.py:81-104: total edits / total reference chars (incl. spaces) and words, x100."""
    te = tw = nc = nw = 0
    for h, r in zip(hyps, refs):
        te += cer_distance(h, r)
        tw += wer_distance(h, r)
        nc += len(r)
        nw += len(r.split())
    return 100.0 * te / nc, 100.0 * tw / nw


def mean_utterance_cer(hyps, refs):
    """codes/metrics.py:114-132,154-162: mean over utterances of Lev/len(ref), x100."""
    tot = 0.0
    for h, r in zip(hyps, refs):
        d = cer_distance(h, r)
        tot += d / len(r) if len(r) else d
    return 100.0 * tot / len(hyps)


def ddp_bins(num_items, batch_size, world, rank):
    """DistributedBucketingSampler.__iter__ -- codes/sampler.py:113-125."""
    ids = list(range(num_items))
    bins = [ids[i:i + batch_size] for i in range(0, num_items, batch_size)]
    per = int(math.ceil(len(bins) / float(world)))
    total = per * world
    bins = bins + bins[:total - len(bins)]
    return bins[rank::world]
