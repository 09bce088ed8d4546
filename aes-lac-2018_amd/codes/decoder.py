"""Greedy CTC decoding and edit-distance metrics (reference ``codes/decoder.py``).

``GreedyDecoder.decode(probs (B,T,A), sizes) -> (strings, offsets)`` keeps the reference's return shape
(``[[str]]`` and ``[[IntTensor]]``, ``codes/decoder.py:99-160``).  The argmax over the alphabet and the
collapse (drop blank; drop a symbol equal to the previous FRAME's symbol) run on the GPU
(``ds2_argmax_rows`` + ``ds2_greedy_collapse``); only the compacted ids cross to the host to become strings.
Edit distance replaces python-Levenshtein (absent here) with ``ds2_edit_distance`` (host C++ in libds2hip.so).
``BeamCTCDecoder`` (CTC prefix beam search, ``ds2_ctc_beam_search``) is an addition: the reference's ``test.py:21``
offers greedy / none only.
"""
import numpy as np
import torch

from ds2hip import lib, ops

from .preprocessing import OrderedLabelEncoder


def _levenshtein(a, b):
    """Edit distance of two sequences of hashable items (``ds2_edit_distance``, host C++)."""
    ids = {}
    ia = np.asarray([ids.setdefault(x, len(ids)) for x in a], dtype=np.int32)
    ib = np.asarray([ids.setdefault(x, len(ids)) for x in b], dtype=np.int32)
    return lib.host_call('ds2_edit_distance', ia, len(ia), ib, len(ib))


class Decoder(object):
    def __init__(self, label_encoder, blank_index=0):
        if isinstance(label_encoder, str):
            label_encoder = list(label_encoder)
        if isinstance(label_encoder, (set, list)):
            label_encoder = OrderedLabelEncoder().fit(label_encoder)
        self.label_encoder = label_encoder
        self.blank_index = blank_index

    def wer(self, s1, s2):
        """Word-level edit distance (codes/decoder.py:49-67)."""
        return _levenshtein(s1.split(), s2.split())

    def cer(self, s1, s2):
        """Character edit distance with spaces removed (codes/decoder.py:69-78)."""
        return _levenshtein(s1.replace(' ', ''), s2.replace(' ', ''))

    def decode(self, probs, sizes=None):
        raise NotImplementedError


class GreedyDecoder(Decoder):
    def _to_string(self, ids):
        if len(ids) == 0:
            return ''
        return ''.join(self.label_encoder.inverse_transform(ids))

    def convert_to_strings(self, sequences, sizes=None, remove_repetitions=False, return_offsets=False):
        """Host path for already-integer sequences (targets), codes/decoder.py:99-140."""
        strings, offsets = [], []
        for i in range(len(sequences)):
            seq = np.asarray(torch.as_tensor(sequences[i]).cpu()).reshape(-1)
            n = int(sizes[i]) if sizes is not None else len(seq)
            keep, offs = [], []
            for t in range(n):
                c = int(seq[t])
                if c == self.blank_index:
                    continue
                if remove_repetitions and t != 0 and c == int(seq[t - 1]):
                    continue
                keep.append(c)
                offs.append(t)
            strings.append([self._to_string(keep)])
            offsets.append([torch.IntTensor(offs)])
        return (strings, offsets) if return_offsets else strings

    def decode(self, probs, sizes=None):
        """probs (B,T,A) on the GPU -> ([[str]], [[IntTensor offsets]]); device argmax + collapse."""
        if not probs.is_cuda:
            raise RuntimeError('GreedyDecoder.decode runs on device tensors only')
        bsz, t, a = probs.shape
        flat = probs.contiguous().float().view(bsz * t, a)
        best = ops.argmax_rows(flat, bsz * t, a).view(bsz, t)
        if sizes is None:
            sizes_d = torch.full((bsz,), t, dtype=torch.int32, device=probs.device)
        else:
            sizes_d = torch.as_tensor(sizes).to(device=probs.device, dtype=torch.int32)
        ids, offs, lens = ops.greedy_collapse(best, sizes_d, self.blank_index)
        ids, offs, lens = ids.cpu().numpy(), offs.cpu().numpy(), lens.cpu().numpy()
        strings = [[self._to_string(ids[b, :lens[b]])] for b in range(bsz)]
        offsets = [[torch.IntTensor(offs[b, :lens[b]].copy())] for b in range(bsz)]
        return strings, offsets


class BeamCTCDecoder(GreedyDecoder):
    """CTC prefix beam search without a language model (not in the reference).

    ``decode(probs (B,T,A), sizes)`` returns the same ``([[str]], [[IntTensor offsets]])`` shape as the greedy
    decoder; the probabilities cross to the host once and the search runs in ``ds2_ctc_beam_search``.
    ``log_input=True`` if ``probs`` are log-probabilities."""

    def __init__(self, label_encoder, blank_index=0, beam_width=16, log_input=False):
        super().__init__(label_encoder, blank_index)
        if beam_width < 1:
            raise ValueError('beam_width must be >= 1')
        self.beam_width, self.log_input = int(beam_width), bool(log_input)
        self.last_log_probs = None

    def decode(self, probs, sizes=None):
        import ctypes
        host = np.ascontiguousarray(torch.as_tensor(probs).detach().float().cpu().numpy())
        bsz, t, a = host.shape
        strings, offsets, scores = [], [], []
        for b in range(bsz):
            n = int(sizes[b]) if sizes is not None else t
            ids, offs = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32)
            length, logp = ctypes.c_int(0), ctypes.c_float(0.0)
            lib.host_call('ds2_ctc_beam_search', host[b, :n], n, a, self.blank_index, self.beam_width,
                          int(self.log_input), ids, offs, len(ids), length, logp)
            strings.append([self._to_string(ids[:length.value])])
            offsets.append([torch.IntTensor(offs[:length.value].copy())])
            scores.append(logp.value)
        self.last_log_probs = scores
        return strings, offsets
