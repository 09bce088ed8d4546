"""Tensor-level wrappers over the C ABI: allocate outputs with torch, launch HIP kernels.

Every function takes/returns ROCm device tensors (fp32 unless noted).  PyTorch is used only as
the allocator and stream owner; all arithmetic happens inside libds2hip.so.
"""
import os

import torch

from . import lib

F_BINS = 161
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _bytes_ws(nbytes, like):
    return torch.empty((int(nbytes) + 15) // 16 * 2, dtype=torch.float64, device=like.device)


def conv_out_frames(t_in):
    t1 = (t_in + 20 - 11) // 2 + 1
    return t1, t1 - 10


# ----------------------------------------------------------------------------- frontend
def spectrogram(wav, wav_offsets, t_max, normalize=True, eps=1e-9):
    """wav: concatenated clips (sum L,), wav_offsets (B+1,) int64 -> (B, t_max, 161).  ``wav_offsets`` on the device, or a
    HOST tensor: it is then staged in page-locked memory the kernels read in place (no copy; ``_PinnedRing.stage``)."""
    bsz = wav_offsets.numel() - 1
    out = _empty((bsz, t_max, F_BINS), wav)
    ws = _bytes_ws(lib.query('ds2_spectrogram_ws_bytes', bsz, t_max), wav)
    if wav_offsets.is_cuda:
        lib.call('ds2_spectrogram_fwd', wav, wav_offsets, bsz, t_max, int(normalize), float(eps), out, ws)
    else:
        slot, staged = _staged.stage(wav_offsets)
        try:
            lib.call('ds2_spectrogram_fwd', wav, staged.data_ptr(), bsz, t_max, int(normalize), float(eps), out, ws)
        finally:
            _staged.staged_done(wav_offsets.dtype, slot)
    return out



# ----------------------------------------------------------------------------- waveform decode + augmentation
WSOLA_MS = (82.0, 14.68, 12.0)          # sox `tempo` defaults: segment, search, overlap


def wsola_params(sample_rate=16000):
    seg = max(int(sample_rate * WSOLA_MS[0] / 1000.0), 4)
    ovl = max(min(int(sample_rate * WSOLA_MS[2] / 1000.0), seg // 2), 1)
    half = max(int(sample_rate * WSOLA_MS[1] / 1000.0) // 2, 1)
    return seg, ovl, half


def wsola_schedule(n, tempo, sample_rate=16000):
    """Segment schedule of one clip (data independent): int32 array of rounded ideal input positions + output length.
    ``tempo`` is used with three decimals, as on the reference's sox command line (codes/transforms.py:197-199)."""
    import numpy as np
    seg, ovl, half = wsola_params(sample_rate)
    tempo = float('{:.3f}'.format(tempo))
    if abs(tempo - 1.0) < 1e-6 or n <= seg + half:
        return np.zeros(0, np.int32), n
    hop_out = seg - ovl
    kmax = int(n / (tempo * hop_out)) + 4
    base = np.rint(np.cumsum(np.full(kmax, tempo * hop_out, dtype=np.float64))).astype(np.int64)   # running sum, half-even
    dead = np.minimum(base + half, n - seg) < np.maximum(base - half, 0)
    nseg = int(np.argmax(dead)) if dead.any() else kmax
    return base[:nseg].astype(np.int32), (nseg + 1) * hop_out + ovl


UNIT_SCALE = 1.0 / 32768.0


def amplitude_scale(spec):
    """The waveform amplitude contract (what ``torchaudio.load`` returned, reference ``codes/transforms.py:156-161``):
    ``'unit'`` / None -> 1/32768 (samples in [-1, 1)); ``'int32'`` -> 65536 (the un-normalised int32-range floats of the
    mid-2018 torchaudio master: sox hands a 16-bit sample on as a 32-bit one); or a positive number."""
    if spec is None or spec == 'unit':
        return UNIT_SCALE
    if spec == 'int32':
        return 65536.0
    scale = float(spec)
    if not scale > 0:
        raise ValueError('amplitude scale must be positive, got %r' % (spec,))
    return scale


def decode_augment(pcm, offsets, tempos=None, gains_db=None, sample_rate=16000, scale=UNIT_SCALE):
    """pcm: concatenated int16 clips on the device; offsets: python list (B+1).  Returns (float clips concatenated,
    new offsets list).  ``tempos`` / ``gains_db`` (per clip, or None): the training-set augmentation -- WSOLA tempo,
    then gain in dB and 16-bit requantisation -- entirely on the device.  ``scale``: a sample of value q (int16, or the
    requantised integer after augmentation) comes out as q * scale (``amplitude_scale``).

    The 16-bit requantisation belongs to the GAIN step (sox writes a 16-bit file at the end of ``tempo T gain G``,
    codes/transforms.py:185-218, and the reference never draws one without the other): it happens exactly when ``gains_db`` is
    given, at every amplitude scale.  A tempo-only call (a diagnostic path: the WSOLA kernel against its specification) returns
    the float WSOLA output times ``scale / UNIT_SCALE`` -- the same samples at every scale, up to that factor."""
    import numpy as np
    wav = _empty((pcm.numel(),), pcm)
    if tempos is None and gains_db is None:
        lib.call('ds2_pcm16_to_float', pcm, pcm.numel(), float(scale), wav)
        return wav, list(offsets)
    lib.call('ds2_pcm16_to_float', pcm, pcm.numel(), UNIT_SCALE, wav)      # the augmentation works on [-1, 1)
    bsz = len(offsets) - 1
    lens = [offsets[i + 1] - offsets[i] for i in range(bsz)]
    if tempos is not None:
        sched = [wsola_schedule(n, t, sample_rate) for n, t in zip(lens, tempos)]
        out_offs = [0]
        for _, m in sched:
            out_offs.append(out_offs[-1] + m)
        boffs = np.concatenate([[0], np.cumsum([len(b) for b, _ in sched])]).astype(np.int32)
        bases = np.concatenate([b for b, _ in sched] + [np.zeros(1, np.int32)]).astype(np.int32)
        # one upload: int64 offsets (in, out) then the int32 schedule viewed as int64 pairs
        meta = np.concatenate([np.asarray(offsets, np.int64), np.asarray(out_offs, np.int64)])
        meta_d = upload_small(torch.from_numpy(meta), pcm.device)
        sched_d = upload_small(torch.from_numpy(np.concatenate([boffs, bases])), pcm.device)
        seg, ovl, half = wsola_params(sample_rate)
        out = _empty((out_offs[-1],), pcm)
        lib.call('ds2_wsola_tempo', wav, meta_d[:bsz + 1], meta_d[bsz + 1:], sched_d[bsz + 1:], sched_d[:bsz + 1], bsz,
                 seg, ovl, half, out)
        wav, offsets = out, out_offs
    if gains_db is not None:
        g = torch.tensor([10.0 ** (float('{:.3f}'.format(v)) / 20.0) for v in gains_db], dtype=torch.float32)
        offs_d = upload_small(torch.tensor(list(offsets), dtype=torch.int64), pcm.device)
        lib.call('ds2_gain_requantize', wav, offs_d, upload_small(g, pcm.device), bsz, float(scale), wav)
    elif scale != UNIT_SCALE:
        wav.mul_(float(scale) / UNIT_SCALE)
    return wav, list(offsets)


# ----------------------------------------------------------------------------- small host <-> device transfers
class _PinnedRing(object):
    """A few reusable page-locked staging buffers per dtype: ``tensor.pin_memory()`` allocates page-locked memory
    on every call (~0.1-0.3 ms of host time each, with the GPU idle at the start of a step).  A slot is reused only
    after the copy that last read it has completed (event per slot); a slot handed out by ``stage`` is BUSY until
    ``staged_done`` and is skipped meanwhile (the ring grows if every slot is busy).  One lock per ring: frontends run on the
    prefetch thread of a loader as well as on the main thread."""

    def __init__(self, slots=4):
        import threading
        self.slots, self.ring, self.next, self.lock = slots, {}, {}, threading.Lock()

    def _slot(self, dtype, n, busy=False):
        """(under the lock) index and buffer of the next slot that is not busy; marks it busy if asked."""
        ring = self.ring.setdefault(dtype, [None] * self.slots)
        start = self.next.get(dtype, 0)
        i = None
        for d in range(len(ring)):
            j = (start + d) % len(ring)
            if ring[j] is None or not ring[j][2]:
                i = j
                break
        if i is None:                             # every slot is between stage() and staged_done(): one more
            ring.append(None)
            i = len(ring) - 1
        self.next[dtype] = (i + 1) % len(ring)
        buf, ev = (ring[i][0], ring[i][1]) if ring[i] is not None else (None, None)
        if ev is not None:
            ev.synchronize()
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(64, 2 * n), dtype=dtype).pin_memory()
        ring[i] = (buf, None, busy)
        return i, buf

    def upload(self, host_tensor, device):
        """1-D CPU tensor -> device tensor through a pinned slot, asynchronously on the current stream."""
        n = host_tensor.numel()
        with self.lock:
            i, buf = self._slot(host_tensor.dtype, n)
            buf[:n].copy_(host_tensor.reshape(-1))
            out = torch.empty(n, dtype=host_tensor.dtype, device=device)
            out.copy_(buf[:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.ring[host_tensor.dtype][i] = (buf, ev, False)
        return out

    def stage(self, host_tensor):
        """1-D CPU tensor -> a page-locked copy the DEVICE reads in place (its address is valid there): for a few dozen bytes a
        kernel reads once, the host-to-device copy costs more than the reads over the bus -- and, issued into an idle queue at
        the top of a step, ~50-80 us before it even starts (kernel trace).  Call ``staged_done(i)`` behind the last launch that
        reads it: until then the slot is busy, afterwards it is reused only once that launch has completed."""
        n = host_tensor.numel()
        with self.lock:
            i, buf = self._slot(host_tensor.dtype, n, busy=True)
            view = buf[:n]
            view.copy_(host_tensor.reshape(-1))
        return i, view

    def staged_done(self, dtype, i):
        ev = torch.cuda.Event()
        ev.record()
        with self.lock:
            self.ring[dtype][i] = (self.ring[dtype][i][0], ev, False)

    def download(self, dev_tensor):
        """1-D device tensor -> pinned host view; valid once the returned event has completed."""
        n = dev_tensor.numel()
        with self.lock:
            i, buf = self._slot(dev_tensor.dtype, n)
            buf[:n].copy_(dev_tensor.reshape(-1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.ring[dev_tensor.dtype][i] = (buf, ev, False)
        return buf[:n], ev


_pinned = _PinnedRing()
_staged = _PinnedRing()          # device-read-in-place staging has its own ring: uploads never wait behind a staged slot


def upload_small(host_tensor, device):
    return _pinned.upload(host_tensor, device)


def download_small(dev_tensor):
    return _pinned.download(dev_tensor)


def spin_wait(event):
    """Poll an event instead of a blocking ``synchronize()`` (which sleeps on an interrupt and wakes up ~0.1 ms late).
    ``sleep(0)`` hands the GIL to any other thread that wants it -- the DataLoader's pin-memory thread, a prefetcher --
    and returns at once when none does: a bare ``while not query(): pass`` starves them for the whole step."""
    import time
    while not event.query():
        time.sleep(0)


def low_priority_stream(device):
    """A torch stream object on a LOW-priority HIP stream (torch.cuda.Stream cannot create one: it clamps to [-1, 0])."""
    import ctypes
    with torch.cuda.device(device):
        out = ctypes.c_void_p()
        rc = lib.load().ds2_stream_create(1, ctypes.byref(out))
        if rc != 0:
            raise RuntimeError('ds2_stream_create failed (%d): %s' % (rc, lib.load().ds2_last_error().decode()))
        return torch.cuda.ExternalStream(out.value, device=device)

# ----------------------------------------------------------------------------- GEMM
def gemm(a, b, trans_a=False, trans_b=False, out=None, beta=0.0, m=None, n=None, k=None, lda=None, ldb=None,
         ldc=None, split_k=1):
    """C = op(a) @ op(b) (+ beta*C).  a, b 2-D row-major unless explicit dims/ld are given."""
    if m is None:
        m = a.shape[1] if trans_a else a.shape[0]
    if k is None:
        k = a.shape[0] if trans_a else a.shape[1]
    if n is None:
        n = b.shape[0] if trans_b else b.shape[1]
    if lda is None:
        lda = a.shape[-1]
    if ldb is None:
        ldb = b.shape[-1]
    if out is None:
        out = _empty((m, n), a)
    if ldc is None:
        ldc = out.shape[-1]
    lib.call('ds2_gemm_f32', int(trans_a), int(trans_b), m, n, k, a, lda, b, ldb, out, ldc, float(beta), split_k)
    return out


def gemm_split_mode(mode=-1):
    """Select the GEMM kernel family (0: f32-input MFMA, 6 / 9: bf16 split-operand kernels, include/ds2hip.h); returns the
    one in effect.  -1 only queries."""
    return int(lib.load().ds2_gemm_split_mode(int(mode)))


def gemm_raw(trans_a, trans_b, m, n, k, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, beta=0.0, split_k=1):
    """Pointer-level GEMM for sub-matrix views (pointers are ints = data_ptr() + byte offsets)."""
    lib.call('ds2_gemm_f32', int(trans_a), int(trans_b), m, n, k, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, float(beta),
             split_k)


def gemm_tn_group(problems, n, k, accumulate=False):
    """Up to four TN problems sharing N and K in one launch: ``problems`` = [(a_ptr, lda, m, b_ptr, ldb, c_ptr, ldc), ...]
    with device pointers as ints.  C_p[m_p, n] = A_p^T B_p (A_p stored k x m_p); ``accumulate``: C_p += instead (the caller
    has zeroed C, e.g. the whole flat gradient in one fill)."""
    import ctypes
    cnt = len(problems)
    assert 1 <= cnt <= 4
    vp, ip = ctypes.c_void_p * cnt, ctypes.c_int * cnt
    a, lda, m, b, ldb, c, ldc = zip(*problems)
    args = (vp(*a), ip(*lda), ip(*m), vp(*b), ip(*ldb), vp(*c), ip(*ldc))
    fn = lib.load().ds2_gemm_f32_tn_group
    rc = fn(cnt, *[ctypes.cast(x, ctypes.c_void_p) for x in args], int(n), int(k), int(bool(accumulate)),
            lib.stream_ptr())
    if rc != 0:
        raise lib.Ds2Error(rc, 'ds2_gemm_f32_tn_group failed (%d): %s' % (rc, lib.load().ds2_last_error().decode()))


def transpose2d(x, rows, cols, out=None):
    if out is None:
        out = _empty((cols, rows), x)
    lib.call('ds2_transpose2d', x, rows, cols, out)
    return out


def transpose2d_group(tensors, batch, rows, cols, out):
    """``tensors``: up to 8 device tensors of ``batch`` row-major (rows, cols) matrices each -> out (len, batch, cols, rows), ONE
    launch (ds2_transpose2d_group)."""
    import ctypes
    cnt = len(tensors)
    assert 1 <= cnt <= 8 and all(t.is_cuda and t.is_contiguous() and t.numel() == batch * rows * cols for t in tensors)
    assert out.is_cuda and out.is_contiguous() and out.numel() == cnt * batch * rows * cols
    ptrs = (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in tensors])
    rc = lib.load().ds2_transpose2d_group(cnt, ctypes.cast(ptrs, ctypes.c_void_p), int(batch), int(rows), int(cols),
                                          out.data_ptr(), lib.stream_ptr())
    if rc != 0:
        raise lib.Ds2Error(rc, 'ds2_transpose2d_group failed (%d): %s' % (rc, lib.load().ds2_last_error().decode()))
    return out


def add2(a, b):
    out = torch.empty_like(a)
    lib.call('ds2_add2', a, b, a.numel(), out)
    return out


# ----------------------------------------------------------------------------- conv stack
def transpose_btf(x):
    bsz, t, f = x.shape
    out = _empty((bsz, f, t), x)
    lib.call('ds2_transpose_btf_to_bft', x, bsz, t, f, out)
    return out


def conv_fwd(which, inp, weight, bias, t_in):
    bsz = inp.shape[0]
    t1, t2 = conv_out_frames(t_in) if which == 1 else (None, t_in - 10)
    out = _empty((bsz, 32, 61, t1), inp) if which == 1 else _empty((bsz, 32, 21, t2), inp)
    ws = _empty((lib.query('ds2_conv_wt_ws_floats', which),), inp)
    lib.call('ds2_conv_fwd', which, inp, weight, bias, bsz, t_in, out, ws)
    return out


def conv2_dgrad(d_out, weight, t1):
    bsz = d_out.shape[0]
    d_in = _empty((bsz, 32, 61, t1), d_out)
    ws = _empty((lib.query('ds2_conv2_dgrad_ws_floats', bsz, t1),), d_out)
    lib.call('ds2_conv2_dgrad', d_out, weight, bsz, t1, d_in, ws, ws.numel())
    return d_in


def conv_wgrad(which, inp, d_out, t_in, d_weight, d_bias):
    lib.call('ds2_conv_wgrad', which, inp, d_out, inp.shape[0], t_in, d_weight, d_bias)


# ----------------------------------------------------------------------------- batch norm
def _bn_ws(c, like):
    return _bytes_ws(lib.query('ds2_bn_ws_bytes', c), like)


def bn2d_stats(x, running_mean, running_var, training):
    bsz, c = x.shape[0], x.shape[1]
    inner = x.numel() // (bsz * c)
    mi = _empty((2 * c,), x)
    lib.call('ds2_bn2d_stats', x, bsz, c, inner, BN_EPS, BN_MOMENTUM, int(not training), running_mean, running_var, mi,
             _bn_ws(c, x))
    return mi


def bn2d_apply_htanh(x, mi, gamma, beta, layout_tbf):
    bsz, c, d, t = x.shape
    y = _empty((t, bsz, c * d), x) if layout_tbf else torch.empty_like(x)
    lib.call('ds2_bn2d_apply_htanh', x, mi, gamma, beta, bsz, c, d, t, int(layout_tbf), y)
    return y


def bn2d_htanh_bwd(x, dy, mi, gamma, beta, dgamma, dbeta):
    bsz, c, d, t = x.shape
    dx = torch.empty_like(x)
    lib.call('ds2_bn2d_htanh_bwd', x, dy, mi, gamma, beta, bsz, c, d, t, dx, dgamma, dbeta, _bn_ws(c, x))
    return dx


def bn1d_stats(xa, xb, rows, feat, running_mean, running_var, training):
    mi = _empty((2 * feat,), xa)
    lib.call('ds2_bn1d_stats', xa, xb, rows, feat, BN_EPS, BN_MOMENTUM, int(not training), running_mean, running_var,
             mi, _bn_ws(feat, xa))
    return mi


def bn1d_apply(xa, xb, mi, gamma, beta, rows, feat):
    y = _empty((rows, feat), xa)
    lib.call('ds2_bn1d_apply', xa, xb, mi, gamma, beta, rows, feat, y)
    return y


def bn1d_bwd(xa, xb, dy, mi, gamma, rows, feat, dgamma, dbeta):
    dx = _empty((rows, feat), xa)
    lib.call('ds2_bn1d_bwd', xa, xb, dy, mi, gamma, rows, feat, dx, dgamma, dbeta, _bn_ws(feat, xa))
    return dx


# ----------------------------------------------------------------------------- GRU recurrence
GRU_MODE = os.environ.get('DS2_GRU_MODE', 'auto')     # 'auto' | 'persistent' | 'step'
_sync_ws = {}
fallback_count = 0                                    # persistent -> per-step fall-backs in this process (bench.py reports it)
_persistent_off = {}                                  # device key -> reason: that device runs the per-step kernels from now on


def _dev_key(dev):
    return (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())


def _gru_sync_ws(dev, bsz, hid):
    """Per-device hand-off workspace of the persistent recurrence (caller-owned, grown on demand, reused).  It is
    zeroed HERE, once: a launch that completes leaves its counters zero for the next one (include/ds2hip.h)."""
    key = _dev_key(dev)
    n = (lib.query('ds2_gru_sync_ws_bytes', bsz, hid) + 3) // 4
    if key not in _sync_ws or _sync_ws[key].numel() < n:
        old = _sync_ws.get(key)
        if old is not None:
            # a larger (B, H) needs a larger workspace: the old one's STICKY timeout flag must not be dropped unread (a
            # launch that timed out earlier in this step, or before a deferred check, left invalid results behind)
            word = lib.query('ds2_gru_sync_error_offset') // 4
            if int(old[word].item()) != 0:
                raise_async_error()
            _err_ptr_tables.clear()                   # cached pointer tables name the old workspace's flag
        _sync_ws[key] = torch.zeros(n, dtype=torch.int32, device=dev)
    return _sync_ws[key]


def _use_persistent(dev, bsz, hid):
    if GRU_MODE == 'step':
        return False
    if _dev_key(dev) in _persistent_off:
        if GRU_MODE == 'persistent':
            raise RuntimeError('persistent GRU kernels are disabled on %s: %s' % (dev, _persistent_off[_dev_key(dev)]))
        return False
    ok = bool(lib.query('ds2_gru_persistent_supported', bsz, hid))
    if GRU_MODE == 'persistent' and not ok:
        raise RuntimeError('persistent GRU kernel does not support B=%d H=%d on this device' % (bsz, hid))
    return ok


def _disable_persistent(dev, reason):
    """Same process, later launches: this device uses the launch-per-step kernels from now on (3x slower at B = 10).
    ``DS2_GRU_STRICT=1`` (set by bench.py: a benchmark must never report the fall-back's rate as the product's) makes
    the fall-back an error instead."""
    import logging
    global fallback_count
    fallback_count += 1
    if os.environ.get('DS2_GRU_STRICT', '0') == '1':
        raise RuntimeError('ds2hip: the persistent GRU kernels would fall back to the launch-per-step kernels on %s (%s) '
                           'and DS2_GRU_STRICT=1 forbids it' % (dev, reason))
    _persistent_off[_dev_key(dev)] = reason
    logging.getLogger('aes-lac-2018').warning('ds2hip: persistent GRU kernels disabled on %s (%s); using the '
                                              'launch-per-step kernels', dev, reason)


def async_error_words():
    """Device views of the STICKY timeout flags of the persistent recurrence (one int32 per device workspace)."""
    word = lib.query('ds2_gru_sync_error_offset') // 4
    return [ws[word:word + 1] for ws in _sync_ws.values()]


def raise_async_error():
    """A bounded spin timed out somewhere since the last check: results of that step are invalid.  Reset the
    workspaces (counters, flag), switch the affected devices to the per-step kernels for later launches, raise."""
    for key, ws in _sync_ws.items():
        word = lib.query('ds2_gru_sync_error_offset') // 4
        bad = int(ws[word].item()) != 0
        ws.zero_()
        if bad:
            _disable_persistent(ws.device, 'a hand-off spin timed out (were all workgroups resident?)')
    raise RuntimeError('ds2hip: persistent GRU kernel timed out waiting for a workgroup hand-off; the results of this '
                       'step are invalid')


def check_async_errors():
    """Raise if a persistent kernel's bounded spin timed out since the last check (call after a device synchronize)."""
    for w in async_error_words():
        if int(w.item()) != 0:
            raise_async_error()


_dh_supported = {}


def gru_bwd_dh_wanted(dev, bsz, hid):
    """Will ``gru_bidir_bwd`` run the d(h)-hand-off backward recurrence for this shape (so that the training forward pass
    should ask for the coefficient planes)?  Default: where the library has the form AND the forward kernel writes the planes
    itself (B = 9 .. 12 at H = 800: measured stand-alone at B = 10, us per step, d(gh) -> d(h) hand-off: 2.55 -> 2.36 on 240
    workgroups, 2.81 -> 2.67 on 174); for B = 5 .. 8 the planes would cost an extra elementwise pass that eats the
    0.03 us per step the form gains there.  ``DS2_GRU_BWD_DH`` = 0 / 1 forces it off / on wherever the form exists."""
    if not _use_persistent(dev, bsz, hid):
        return False
    key = (bsz, hid)
    if key not in _dh_supported:                      # (a shape's answer never changes: one library query per shape)
        _dh_supported[key] = bool(lib.query('ds2_gru_bwd_dh_supported', bsz, hid))
    if not _dh_supported[key]:
        return False
    e = os.environ.get('DS2_GRU_BWD_DH')
    if e in ('0', '1'):
        return e == '1'
    return bsz >= 9


def gru_bidir_fwd(gates, w_hh, t, bsz, hid, want_coef=False):
    """gates (T,B,2,3H) holds gi on entry, (r,z,n) on exit.  Returns (ghn (T,B,2,H), hout (2,T,B,H)), and with ``want_coef``
    also the (T,B,2,3H) coefficient planes of the d(h)-hand-off backward recurrence (None where that form will not run)."""
    ghn = _empty((t, bsz, 2, hid), gates)
    hout = _empty((2, t, bsz, hid), gates)
    coef = _empty((t, bsz, 2, 3 * hid), gates) if want_coef and gru_bwd_dh_wanted(gates.device, bsz, hid) else None
    if _use_persistent(gates.device, bsz, hid):
        try:
            lib.call('ds2_gru_bidir_fwd_persistent_ex', gates, ghn, hout, w_hh, coef, _gru_sync_ws(gates.device, bsz, hid), t,
                     bsz, hid)
            return (ghn, hout, coef) if want_coef else (ghn, hout)
        except lib.Ds2Error as e:               # nothing was launched: the chosen kernel's grid is not co-resident here
            if e.code != lib.ERR_UNSUPPORTED or GRU_MODE == 'persistent':
                raise
            _disable_persistent(gates.device, str(e))
    lib.call('ds2_gru_bidir_fwd', gates, ghn, hout, w_hh, t, bsz, hid)
    return (ghn, hout, None) if want_coef else (ghn, hout)     # (the launch-per-step backward kernels take no coefficients)


def gru_bwd_coef(gates, ghn, hout, t, bsz, hid):
    """(T,B,2,3H) coefficient planes of the d(h) hand-off backward recurrence from the forward pass's saved tensors."""
    coef = _empty((t, bsz, 2, 3 * hid), gates)
    lib.call('ds2_gru_bwd_coef', gates, ghn, hout, coef, t, bsz, hid)
    return coef


def gru_bidir_bwd(gates, ghn, hout, d_out, w_hh_t, t, bsz, hid, spare_cus=-1, coef=None):
    """``spare_cus``: compute units to leave free beside the launch for work queued on other streams (-1: the library's
    default; see ds2_gru_bidir_bwd_persistent_ex in include/ds2hip.h).  ``coef``: the forward pass's coefficient planes
    (``gru_bidir_fwd(.., want_coef=True)``) -> the d(h)-hand-off form; None -> the d(gh)-hand-off form."""
    if _use_persistent(gates.device, bsz, hid):
        try:
            if coef is not None:
                lib.call('ds2_gru_bidir_bwd_persistent_dh', gates, ghn, hout, d_out, w_hh_t, coef,
                         _gru_sync_ws(gates.device, bsz, hid), t, bsz, hid, int(spare_cus))
                return
            lib.call('ds2_gru_bidir_bwd_persistent_ex', gates, ghn, hout, d_out, w_hh_t,
                     _gru_sync_ws(gates.device, bsz, hid), t, bsz, hid, int(spare_cus))
            return
        except lib.Ds2Error as e:
            if e.code != lib.ERR_UNSUPPORTED or GRU_MODE == 'persistent':
                raise
            _disable_persistent(gates.device, str(e))
    ws = torch.zeros((2 * 2 * bsz * hid,), dtype=torch.float32, device=gates.device)
    lib.call('ds2_gru_bidir_bwd', gates, ghn, hout, d_out, w_hh_t, ws, t, bsz, hid)


# ----------------------------------------------------------------------------- head / decode
def softmax_rows(x, rows, a):
    y = torch.empty_like(x)
    lib.call('ds2_softmax_rows', x, rows, a, y)
    return y


def argmax_rows(x, rows, a):
    idx = _empty((rows,), x, torch.int32)
    lib.call('ds2_argmax_rows', x, rows, a, idx)
    return idx


def greedy_collapse(best, sizes, blank=0):
    bsz, t = best.shape
    ids = torch.zeros((bsz, t), dtype=torch.int32, device=best.device)
    offs = torch.zeros((bsz, t), dtype=torch.int32, device=best.device)
    lens = torch.zeros((bsz,), dtype=torch.int32, device=best.device)
    lib.call('ds2_greedy_collapse', best, sizes, bsz, t, blank, ids, offs, lens)
    return ids, offs, lens


# ----------------------------------------------------------------------------- CTC
def ctc_loss_grad(acts, labels, label_offsets, label_lens, act_lens, max_label_len, grad_scale=1.0,
                  zero_batch_if_inf=False):
    """acts (T,B,A) -> costs (B,), grad (T,B,A); int32 device tensors for the rest.  ``zero_batch_if_inf``: the training
    step's rule -- an infinite batch loss zeroes the whole batch's gradient (codes/engine.py:24-30)."""
    t, bsz, a = acts.shape
    costs = _empty((bsz,), acts)
    grad = torch.empty_like(acts)
    ws = _bytes_ws(lib.query('ds2_ctc_ws_bytes', t, bsz, a, max_label_len), acts)
    lib.call('ds2_ctc_loss_grad', acts, labels, label_offsets, label_lens, act_lens, t, bsz, a, max_label_len,
             float(grad_scale), int(zero_batch_if_inf), costs, grad, ws)
    return costs, grad


# ----------------------------------------------------------------------------- optimiser
def sumsq(x, out=None):
    if out is None:
        out = torch.empty((1,), dtype=torch.float64, device=x.device)
    ws = _bytes_ws(lib.query('ds2_sumsq_ws_bytes', x.numel()), x)
    lib.call('ds2_sumsq', x, x.numel(), out, ws)
    return out


_err_ptr_tables = {}


def step_stats(costs, sumsq_t, out=None):
    """One launch gathering what the host reads after a step: [sum of costs, squared grad norm, any timeout flag,
    number of infinite costs] as float64 (4,)."""
    dev = costs.device
    words = async_error_words()
    key = (_dev_key(dev), tuple(w.data_ptr() for w in words))
    table = _err_ptr_tables.get(key)
    if table is None:
        table = _err_ptr_tables[key] = torch.tensor([w.data_ptr() for w in words] or [0], dtype=torch.int64).to(dev)
    if out is None:
        out = torch.empty((4,), dtype=torch.float64, device=dev)
    # ``out`` may be PAGE-LOCKED HOST memory (its address is valid on the device): the kernel then writes the four values
    # straight to the host, which polls them (wait_step_stats) -- no copy, no event (tools/attic/readback_probe.py: 14.5 against
    # 32-34 us from launch to the host having the values)
    lib.call('ds2_step_stats', costs, costs.numel(), sumsq_t, table, len(words), out if out.is_cuda else out.data_ptr())
    return out


STATS_SENTINEL = 0x7FF8DEADBEEF0001          # a NaN payload no arithmetic produces: "not written yet"


def arm_step_stats(slot):
    """Mark a page-locked float64 (4,) slot as not yet written (before the launch that will write it)."""
    slot.view(torch.int64).fill_(STATS_SENTINEL)


def _stats_timeout():
    """Seconds the host waits for a step's statistics before giving up: ``DS2_STATS_TIMEOUT_S`` if set (``0`` / ``inf`` = no
    limit); otherwise no limit inside a process group -- the statistics are queued behind the gradient all-reduce there, a
    slow peer (rank 0 saving a checkpoint, uneven evaluation shards, a cold loader) is not an error, and the group's own
    watchdog bounds the wait -- and 120 s for a single-GPU run."""
    v = os.environ.get('DS2_STATS_TIMEOUT_S')
    if v is not None:
        t = float(v)
        return None if t <= 0 or t == float('inf') else t
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return None
    return 120.0


def wait_step_stats(slot, timeout_s='default', stream=None):
    """Poll a slot armed by ``arm_step_stats`` until the device has written all four values.

    While waiting, the stream the step was queued on is queried from time to time: a device fault raises as the HIP error it
    is (not as a time-out minutes later), and a stream that has DRAINED without the slot being written -- the statistics
    kernel never ran -- raises at once.  ``timeout_s``: None = wait without limit, 'default' = ``_stats_timeout()``."""
    import time
    words = slot.view(torch.int64).numpy()
    t0, next_query, every = None, 0.0, 0.02
    while (words == STATS_SENTINEL).any():
        time.sleep(0)                             # (hands the GIL to a loader / pin-memory thread that wants it)
        now = time.time()
        if t0 is None:
            t0, next_query = now, now + every
            if timeout_s == 'default':
                timeout_s = _stats_timeout()
            continue
        if now < next_query:
            continue
        every = min(every * 2, 1.0)
        next_query = now + every
        st = stream if stream is not None else torch.cuda.current_stream()
        if st.query():                            # (raises the pending HIP error of a faulted device)
            time.sleep(0.001)
            if (words == STATS_SENTINEL).any():
                raise RuntimeError('the stream has run everything queued on it and the step\'s statistics were never '
                                   'written to host memory')
        if timeout_s is not None and now - t0 > timeout_s:
            raise RuntimeError('the step\'s statistics never arrived in host memory (%.0f s; DS2_STATS_TIMEOUT_S)' % timeout_s)
    return slot.tolist()


def clip_sgd_nesterov(p, g, buf, sumsq_t, grad_scale, max_norm, lr, momentum, first_step):
    lib.call('ds2_clip_sgd_nesterov', p, g, buf, p.numel(), sumsq_t, float(grad_scale), float(max_norm), float(lr),
             float(momentum), int(first_step))
