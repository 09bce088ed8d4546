#!/usr/bin/env python
"""Train-throughput benchmark of the MI355X DeepSpeech2 hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full training pass over one minibatch of synthetic 16 kHz clips already resident in HBM:
GPU log-spectrogram frontend -> conv/BiGRU/FC forward -> CTC -> backward -> (RCCL gradient all-reduce)
-> clip + Nesterov SGD -> synchronize (the reference's codes/engine.py:45-94).  Workload at N=1:
BASELINE configs[1] (scripts/librispeech-from_scratch.json): the default 5xBiGRU-800 model, A=29, batch
10, clip durations uniform on [1 s, 15 s] (numpy default_rng(42)), length-sorted bins of 10 like
BucketingSampler, labels 14 chars/s.  N>1: each rank takes every N-th bin (DistributedBucketingSampler
rule, codes/sampler.py:119-125), per-GPU batch stays 10 -> weak scaling; value = frames of all ranks /
max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'aes-lac-2018_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32-input MFMA dense peak
TRAIN_FLOP_PER_OUT_STEP = 261.94e6  # 3 x 87.31 MFLOP per output step per utterance
HOP = 160
NUM_BINS = 24


def bin_plan(batch_size, num_bins, seed=42, world=1):
    """SURVEY.md 8(d) synthetic corpus, as a PLAN: a list of (bin_seed, durations[batch_size]).

    Durations are uniform on [1 s, 15 s], sorted, cut into bins of batch_size (BucketingSampler).  Groups of `world`
    ADJACENT bins are the units of the order, so that the bins the `world` ranks consume in the same step under the
    every-world-th rule (codes/sampler.py:119-125) hold clips of similar length (minimises straggling).
    The order INTERLEAVES short and long groups (shortest, longest, second shortest, second longest, ...): any window
    of consecutive steps then has the corpus's mean clip length, so the headline does not depend on which `--steps` of
    the 24 bins get timed (a random shuffle moved it by ~1 %).  Audio is only synthesised for the bins a process
    actually uses (make_bin)."""
    rng = np.random.default_rng(seed)
    dur = np.sort(rng.uniform(1.0, 15.0, size=batch_size * num_bins))
    bins = [(seed * 100003 + i, dur[i * batch_size:(i + 1) * batch_size]) for i in range(num_bins)]
    ngroups = num_bins // world
    groups = [g // 2 if g % 2 == 0 else ngroups - 1 - g // 2 for g in range(ngroups)]
    return [bins[g * world + r] for g in groups for r in range(world)]


def make_bin(plan_entry, nalpha=29):
    """(wavs list, labels, label_lens): N(0, 0.1^2) audio clipped to [-1, 1], labels uniform on 1..A-1 at 14 chars/s."""
    bin_seed, durations = plan_entry
    rng = np.random.default_rng(bin_seed)
    wavs, labels, lens = [], [], []
    for d in durations:
        nsamp = int(round(d * 16000))
        wavs.append(np.clip(0.1 * rng.standard_normal(nsamp), -1.0, 1.0).astype(np.float32))
        ll = max(1, int(round(14.0 * d)))
        labels.append(rng.integers(1, nalpha, size=ll).astype(np.int32))
        lens.append(ll)
    return wavs, np.concatenate(labels), np.asarray(lens, np.int32)


def frames_of_plan(plan_entry):
    return int(sum(1 + int(round(d * 16000)) // HOP for d in plan_entry[1]))


def frames_of(bin_):
    return sum(1 + len(w) // HOP for w in bin_[0])


def out_steps_of(bin_):
    t_in = max(1 + len(w) // HOP for w in bin_[0])
    return ((t_in + 9) // 2 - 9) * len(bin_[0])      # padded output steps actually computed


def valid_out_steps_of(bin_):
    """Output steps the utterances THEMSELVES have (each clip's own T = floor((T_in + 9) / 2) - 9): the algorithmic work.
    ``out_steps_of`` counts what is computed (every utterance padded to the bin's longest clip)."""
    return sum(((1 + len(w) // HOP) + 9) // 2 - 9 for w in bin_[0])


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def gpu_parity_side(model, trainer, front, decoder, plan_entry, dev):
    """The HIP path's half of ``parity_vs_cpu_oracle``: on ONE bin (the longest of the corpus), from the model's CURRENT
    weights -- eval-mode probabilities -> greedy strings, then one training step with the learning rate set to 0 (loss,
    gradient norm; the weights stay put) and the train-mode logits.  Returns what cpu_baseline() needs to repeat it."""
    from ds2hip import ops
    bin_ = make_bin(plan_entry)
    flat, offs, labels, lens = make_resident(bin_, dev)
    trainer.flush()
    torch.cuda.synchronize()
    sd = {k: v.detach().to('cpu').clone() for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        inputs, pct = front(flat, offs)
        probs = model(inputs)
        sizes = (pct * probs.shape[1]).int()
        strings, _ = decoder.decode(probs, sizes)
    model.train()
    lrs = [g['lr'] for g in trainer.optimizer.param_groups]
    for g in trainer.optimizer.param_groups:
        g['lr'] = 0.0
    try:
        loss = trainer.update((inputs, labels, pct, lens), defer=False)
        gnorm = trainer.last_grad_norm
        with torch.no_grad():
            acts, _ = model._forward_impl(inputs, training=True, need_grad=False)
    finally:
        for g, lr in zip(trainer.optimizer.param_groups, lrs):
            g['lr'] = lr
    torch.cuda.synchronize()
    ops.check_async_errors()
    return {'state_dict': sd, 'plan_entry': plan_entry, 'logits': acts.transpose(0, 1).cpu().numpy(),
            'loss_sum': float(loss) * len(bin_[0]), 'gnorm': float(gnorm), 'strings': [s[0] for s in strings],
            'argmax': probs.argmax(-1).cpu().numpy(),
            'out_sizes': sizes.cpu().numpy(), 'labels_txt': None}


def cpu_baseline(plan, budget_s=60.0, full=True, parity=None):
    """The oracle (stock PyTorch CPU ops, numerically the reference) timed on the host cores: full training steps
    (frontend -> fwd -> CTC -> bwd -> clip -> SGD) on B=10 bins of the SAME workload.  Protocol:
    2 untimed warm-up steps (the shortest bin), then timed steps on five bins spread evenly over the length range
    (0/25/50/75/100 % of the sorted bins), shortest first; value = median of the per-step frames/s.  A CPU step on a
    median bin takes ~40 s; the DEFAULT is the complete protocol -- all five bins plus one 8 x 15 s step (BASELINE
    configs[3]'s per-GPU shape), ~4 min -- and --cpu-quick stops adding bins once `budget_s` of timed work is spent (and
    says how many of the five it timed)."""
    import torch.nn.functional as F
    from oracle import spectrogram as ospec
    from oracle.model import OracleDeepSpeech
    # threads: torch's default (= physical cores; 128 on the 256-hardware-thread GPU box).  Forcing os.cpu_count() = 256
    # threads, as BASELINE.md suggests, oversubscribes the cores: the oracle then needs minutes for ONE short step.
    torch.manual_seed(0)
    model = OracleDeepSpeech()
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    model.train()

    seen = {}

    def step(bin_):
        wavs, labels, lens = bin_
        t0 = time.time()
        x, pct = ospec.batch_log_spectrogram(wavs)
        logits = model(torch.from_numpy(x))
        out_sizes = (torch.from_numpy(pct) * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                          torch.from_numpy(lens).long(), blank=0, reduction='sum') / len(wavs)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 400)
        opt.step()
        dt = time.time() - t0
        seen.update(logits=logits.detach().numpy(), loss_sum=float(loss.item()) * len(wavs), gnorm=float(gn))
        return dt

    order = sorted(range(len(plan)), key=lambda i: frames_of_plan(plan[i]))
    picks = [order[int(round(q * (len(order) - 1)))] for q in (0.0, 0.25, 0.5, 0.75, 1.0)]
    warm = make_bin(plan[order[0]])
    for _ in range(2):
        step(warm)
    rates, used, spent = [], [], 0.0
    parity_out = None
    # the parity leg's bin (the longest) goes FIRST among the timed steps, so that --cpu-quick cannot skip it
    if parity is not None:
        want = [i for i in picks if plan[i] is parity['plan_entry'] or plan[i][0] == parity['plan_entry'][0]]
        picks = want + [i for i in picks if i not in want]
    for idx in picks:
        b = make_bin(plan[idx])
        check = parity is not None and plan[idx][0] == parity['plan_entry'][0]
        if check:
            # the SAME weights and BatchNorm buffers as the HIP model had for this bin (the oracle's momentum buffers do not
            # enter the forward pass, the loss or the gradient); eval-mode decode first (outside the timed step), then the
            # timed training step, whose logits / loss / gradient norm are what the HIP step is compared with
            from oracle import host as ohost
            model.load_state_dict(parity['state_dict'])
            model.eval()
            with torch.no_grad():
                xe, pe = ospec.batch_log_spectrogram(b[0])
                probs = model(torch.from_numpy(xe)).numpy()
            model.train()
            sizes = (torch.from_numpy(pe) * probs.shape[1]).int().numpy()
            cpu_strings, _ = ohost.greedy_decode(probs, sizes, parity['labels_txt'])
        dt = step(b)
        if check:
            gl, cl = parity['logits'], seen['logits']
            valid = np.arange(cl.shape[1])[None, :] < sizes[:, None]          # frames inside each utterance's out_size
            diff = np.abs(gl - cl).max(-1)
            parity_out = {
                'bin': '%d clips of %.1f-%.1f s, T_in = %d' % (len(b[0]), float(plan[idx][1].min()), float(plan[idx][1].max()),
                                                                xe.shape[1]),
                't_out': int(cl.shape[1]),
                'logits_max_abs_err': float(diff[valid].max()),
                'logits_max_abs_err_incl_padding_frames': float(diff.max()),
                'loss_rel_err': abs(parity['loss_sum'] - seen['loss_sum']) / abs(seen['loss_sum']),
                'loss_sum_hip': parity['loss_sum'], 'loss_sum_cpu_oracle': seen['loss_sum'],
                'grad_norm_rel_err': abs(parity['gnorm'] - seen['gnorm']) / seen['gnorm'],
                'greedy_strings_equal': bool(parity['strings'] == list(cpu_strings)),
                'greedy_strings_differing': int(sum(a != c for a, c in zip(parity['strings'], cpu_strings))),
                'greedy_string_chars': int(sum(len(a) for a in parity['strings'])),
                # every valid frame's eval-mode argmax (what the strings are collapsed from; the strings of a barely trained
                # model can be short): frames where the two paths pick different classes, and how close the oracle's two best
                # classes are there (a tie at fp32 round-off is not a disagreement)
                'argmax_frames': int(valid.sum()),
                'argmax_frames_differing': int(((parity['argmax'] != probs.argmax(-1)) & valid).sum()),
                'argmax_differing_frames_min_margin': (lambda m: float(m.min()) if m.size else None)(
                    (np.sort(probs, -1)[..., -1] - np.sort(probs, -1)[..., -2])[(parity['argmax'] != probs.argmax(-1)) & valid]),
                'out_sizes_equal': bool(np.array_equal(parity['out_sizes'], sizes)),
                'tolerances': 'north_star: logits 1e-3 abs, CTC loss 1e-4 rel, greedy strings identical',
                'unpinned_third_party': 'the oracle itself is pinned to the reference model (tests/golden/make_golden.py, T up to 746); '
                                        'what the reference delegates to code absent from its tree stays unpinned: warp-ctc (stand-in: '
                                        'F.ctc_loss + fp64 path enumeration; the infeasible-utterance rule is a recollection), librosa '
                                        '(stand-in: torch.stft), torchaudio / sox (amplitude scale and tempo resampling: README.md)',
                'what': 'HIP path (STFT frontend -> model -> CTC -> backward) against the CPU oracle (oracle/: numpy STFT, '
                        'torch CPU conv/BN/GRU/Linear, F.ctc_loss) on the SAME audio from the SAME weights: the HIP model\'s '
                        'state_dict after this run\'s training steps is loaded into the oracle before its step on this bin'}
            note('parity vs cpu oracle: %s' % json.dumps({k: v for k, v in parity_out.items() if k not in ('what', 'tolerances')}))
        note('cpu baseline: bin of %d frames in %.1f s' % (frames_of(b), dt))
        rates.append(frames_of(b) / dt)
        used.append('%.1f s clips: %d frames in %.1f s' % (float(np.mean(plan[idx][1])), frames_of(b), dt))
        spent += dt
        if not full and spent > budget_s:
            break
    out = {'value': round(float(np.median(rates)), 1), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'cpu_model': cpu_model_name(), 'host_cpus': os.cpu_count(),
           'threads_note': 'torch default = one thread per physical core; os.cpu_count() threads (two per core, as '
                           'BASELINE.md suggests) oversubscribes the cores: one short oracle step then takes minutes',
           'per_step_frames_per_s': [round(r, 1) for r in rates],
           'sample': 'oracle (torch CPU conv/BN/GRU/Linear + F.ctc_loss + clip + SGD) full training steps at B=10 on %d of '
                     'the 5 bins spread over the 1-15 s length range (2 untimed warm-up steps on the shortest bin first; '
                     'median of per-step frames/s): %s' % (len(rates), '; '.join(used))}
    if full:
        rng = np.random.default_rng(7)
        wavs = [np.clip(0.1 * rng.standard_normal(240000), -1, 1).astype(np.float32) for _ in range(8)]
        lens = np.full(8, 210, np.int32)
        b = (wavs, rng.integers(1, 29, size=int(lens.sum())).astype(np.int32), lens)
        dt = step(b)
        out['config3_8x15s'] = {'value': round(frames_of(b) / dt, 1), 'unit': 'frames/s', 'seconds': round(dt, 1)}
    if parity_out is not None:
        out['parity_vs_cpu_oracle'] = parity_out
    return out


def gru_pass_roofline(model, bsz, t):
    """Live HIP-event timing of the dominant kernels: one persistent BiGRU layer pass (ONE launch covering
    all T steps of both directions), forward and backward, on torch's current stream = the launch stream."""
    from ds2hip import ops
    hid = model._rnn_hidden_size
    dev = model._flat_p.device
    r = model.rnns[1].rnn
    w_hh = model._pair(r.weight_hh_l0, r.weight_hh_l0_reverse).view(2, 3 * hid, hid)
    w_hh_t = torch.stack([ops.transpose2d(w_hh[0], 3 * hid, hid), ops.transpose2d(w_hh[1], 3 * hid, hid)], 0)
    gates = 0.1 * torch.randn(t, bsz, 2, 3 * hid, device=dev)
    d_out = 0.01 * torch.randn(t, bsz, hid, device=dev)
    # The backward launch exists in forms that leave different numbers of CUs to the side stream's weight-gradient GEMMs
    # (codes/model.py: none under the top layer, _BWD_SPARE_CUS under the others): both are timed, and 'bwd' is what a step
    # launches -- one launch of the first, num_layers - 1 of the second.
    from codes.model import _BWD_SPARE_CUS
    res = {'fwd': [], 'bwd_top': [], 'bwd_below': []}
    for rep in range(10):
        g = gates.clone()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ghn, hout, coef = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid, want_coef=True)       # as a training pass launches them
        e[1].record()
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid, spare_cus=0 if rep % 2 == 0 else _BWD_SPARE_CUS, coef=coef)
        e[2].record()
        torch.cuda.synchronize()
        ops.check_async_errors()
        res['fwd'].append(e[0].elapsed_time(e[1]) * 1e-3)
        res['bwd_top' if rep % 2 == 0 else 'bwd_below'].append(e[1].elapsed_time(e[2]) * 1e-3)
    flop = 2.0 * 2 * bsz * hid * 3 * hid * t                           # both directions, all T steps
    out = {}
    for name in ('fwd', 'bwd_top', 'bwd_below'):
        dur = float(np.median(res[name]))
        out[name] = (flop / dur / 1e12, dur, flop)
    nl = model._num_rnn_layers
    dur = (out['bwd_top'][1] + (nl - 1) * out['bwd_below'][1]) / nl     # the mean launch of a step
    out['bwd'] = (flop / dur / 1e12, dur, flop)
    return out


def recurrence_floor(bsz, t, hid, dev, spare_below):
    """LIVE latency-model floor of one recurrence time step, measured in this run by ablating the SHIPPED kernels' own source:
    the fault-injection build of the library (csrc/build.py: the release objects with gru_persist.hip compiled with
    -DDS2_FAULT_INJECT=1, no timing stamps) honours DS2_GRU_DBG bits that switch phases of a step off -- 2: no hand-off loads
    and no MFMAs (what is left is the step's skeleton: barriers, LDS round trip, gate math, stores), 2048: loads issued but
    never validated, so nobody waits for anybody (matrix phase + skeleton), 4096: no MFMAs (hand-off + skeleton), 8192: no
    prefetch of the next step's saved activations (a full step never waits for those HBM loads; a shortened one would).  The
    hand-off and the matrix phase overlap in the shipped kernels (a fragment is multiplied as soon as it has landed), so
    floor = max(hand-off, matrix) + skeleton.  Results of the ablated launches are wrong by construction; nothing reads them."""
    import ctypes
    path = os.path.join(ROOT, 'aes-lac-2018_amd', 'ds2hip', 'libds2hip_faultinject.so')
    if not os.path.exists(path) or bsz < 9 or bsz > 12 or hid != 800:
        return None
    L = ctypes.CDLL(path)
    L.ds2_gru_sync_ws_bytes.restype = ctypes.c_size_t
    L.ds2_gru_sync_ws_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    vp, ci = ctypes.c_void_p, ctypes.c_int
    # the forms a training step launches: the forward pass with the coefficient planes as an output, the backward recurrence
    # that hands off dh and takes them (include/ds2hip.h, ABI revision 402)
    L.ds2_gru_bidir_fwd_persistent_ex.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
    L.ds2_gru_bidir_bwd_persistent_dh.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    ws = torch.zeros(L.ds2_gru_sync_ws_bytes(bsz, hid) // 4 + 64, dtype=torch.int32, device=dev)
    w = ((torch.rand(2, 3 * hid, hid, device=dev) * 2 - 1) / hid ** 0.5)
    wt = w.transpose(1, 2).contiguous()
    gates = 0.1 * torch.randn(t, bsz, 2, 3 * hid, device=dev)
    ghn = torch.zeros(t, bsz, 2, hid, device=dev)
    hout = torch.zeros(2, t, bsz, hid, device=dev)
    d_out = 0.01 * torch.randn(t, bsz, hid, device=dev)
    coef = torch.zeros(t, bsz, 2, 3 * hid, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    saved = {k: os.environ.get(k) for k in ('DS2_GRU_DBG',)}

    def timed(which, bits, spare=0):
        os.environ['DS2_GRU_DBG'] = str(bits)
        durs = []
        for _ in range(4):
            g = gates.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == 'fwd':
                rc = L.ds2_gru_bidir_fwd_persistent_ex(g.data_ptr(), ghn.data_ptr(), hout.data_ptr(), w.data_ptr(), coef.data_ptr(),
                                                       ws.data_ptr(), t, bsz, hid, st)
            else:
                rc = L.ds2_gru_bidir_bwd_persistent_dh(g.data_ptr(), ghn.data_ptr(), hout.data_ptr(), d_out.data_ptr(),
                                                       wt.data_ptr(), coef.data_ptr(), ws.data_ptr(), t, bsz, hid, spare, st)
            e1.record()
            torch.cuda.synchronize()
            if rc != 0:
                return None
            durs.append(e0.elapsed_time(e1) * 1e3 / t)
        ws.zero_()
        return float(np.median(durs[1:]))

    out = {}
    try:
        for name, which, spare in (('fwd', 'fwd', 0), ('bwd_top_layer_form', 'bwd', 0), ('bwd_lower_layers_form', 'bwd', spare_below)):
            full, skel, mat, hand = (timed(which, b, spare) for b in (0, 2 + 8192, 2048 + 8192, 4096 + 8192))
            if None in (full, skel, mat, hand):
                return None
            out[name] = {'skeleton': round(skel, 3), 'handoff': round(hand - skel, 3), 'matrix': round(mat - skel, 3),
                         'floor': round(max(hand, mat), 3), 'achieved_same_build': round(full, 3)}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out['note'] = ('us per time step, measured in THIS run on the fault-injection build of the shipped kernels (DS2_GRU_DBG '
                   'ablation bits: 2 skeleton only, 2048 no validation = matrix + skeleton, 4096 no MFMAs = hand-off + '
                   'skeleton, each with 8192 = no saved-activation prefetch, whose HBM latency a full step never waits for but a '
                   'shortened one would); hand-off and matrix both contain the hand-off loads\' own round trip; floor = max(hand-off, matrix) + skeleton, because the shipped kernels multiply a fragment as '
                   'soon as it has landed; achieved_same_build = the unablated launch of the same library)')
    return out


PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak; 6 bf16 products per fp32-equivalent one


def gemm_roofline(model, rows):
    """Live HIP-event timing of the model's GEMM shapes at the mean bin (rows = T * B): the input projection (NT), the dX
    GEMM (NN), dW_ih (TN) and the grouped dW_hh launch are the GEMM work of a layer.  fp32-equivalent TFLOP/s against the
    bf16 pipe's dense peak / 6 partial products (the default split-operand family; DS2_GEMM_SPLIT=0: the fp32 pipe)."""
    from ds2hip import ops
    hid = model._rnn_hidden_size
    dev = model._flat_p.device
    r = model.rnns[1].rnn
    w_ih = model._pair(r.weight_ih_l0, r.weight_ih_l0_reverse)
    x = 0.5 * torch.randn(rows, hid, device=dev)
    dg = 0.1 * torch.randn(rows, 6 * hid, device=dev)
    dw = torch.zeros(6 * hid, hid, device=dev)

    def timed(fn, reps=6):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    flop = 2.0 * rows * 6 * hid * hid
    shapes = {
        'input_projection_NT_%dx%dx%d' % (rows, 6 * hid, hid): timed(lambda: ops.gemm(x, w_ih, trans_b=True)),
        'dX_NN_%dx%dx%d' % (rows, hid, 6 * hid): timed(lambda: ops.gemm(dg, w_ih, split_k=0)),
        'dW_ih_TN_%dx%dx%d' % (6 * hid, hid, rows): timed(lambda: ops.gemm(dg, x, trans_a=True, out=dw, split_k=0)),
    }
    mode = ops.gemm_split_mode()
    peak = PEAK_BF16_MFMA_TFLOPS / mode if mode else PEAK_F32_MFMA_TFLOPS
    out = {'bound': 'mfma', 'unit': 'TFLOP/s (fp32-equivalent)', 'peak': round(peak, 1),
           'peak_note': ('dense bf16 MFMA peak 2500 / %d partial products per fp32-equivalent product' % mode) if mode
                        else 'fp32-input MFMA dense peak', 'shapes': {}}
    for k, dur in shapes.items():
        out['shapes'][k] = {'us': round(dur * 1e6, 1), 'achieved': round(flop / dur / 1e12, 1),
                            'frac': round(flop / dur / 1e12 / peak, 4)}
    first = next(iter(out['shapes'].values()))
    out.update(kernel='gemm_bf16x_kernel (input projection of a BiGRU layer)' if mode else 'gemm_f32_kernel',
               achieved=first['achieved'], frac=first['frac'])
    return out


def chain_time_share(model, trainer, front, resident_bin):
    """ONE training step on one bin with a timing event at every phase boundary of the main stream (model._tick): the live
    counterpart of profiles/*_kernel_stats.csv.  Times are CHAIN times -- what the main stream spends between two
    boundaries, side-stream work that runs beside it not counted, waits for it counted."""
    flat, offs, labels, lens = resident_bin
    trainer.flush()
    for _ in range(2):
        inputs, pct = front(flat, offs)
        trainer.update((inputs, labels, pct, lens), defer=False)
    torch.cuda.synchronize()
    model._ticks = []
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    inputs, pct = front(flat, offs)
    trainer.update((inputs, labels, pct, lens), defer=False)
    torch.cuda.synchronize()
    ticks, model._ticks = model._ticks, None
    share, prev = {}, e0
    for name, ev in ticks:
        if name == 'start':
            name = 'frontend: STFT + log + normalise'
        share[name] = share.get(name, 0.0) + prev.elapsed_time(ev)
        prev = ev
    total = sum(share.values())
    return {'total_ms': round(total, 3), 'T_in': int(inputs.shape[1]), 'B': int(inputs.shape[0]),
            'phases_ms': {k: round(v, 3) for k, v in share.items()},
            'phases_share': {k: round(v / total, 4) for k, v in share.items()}}


_T0 = time.time()


def _fallbacks():
    from ds2hip import ops
    return int(ops.fallback_count)


def note(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    sys.stderr.write('[bench %6.1f s] %s\n' % (time.time() - _T0, msg))
    sys.stderr.flush()


def make_resident(bin_, dev):
    wavs, labels, lens = bin_
    flat = torch.from_numpy(np.concatenate(wavs)).to(dev)
    offs = np.concatenate([[0], np.cumsum([len(w) for w in wavs])]).astype(np.int64)
    return flat, offs, torch.from_numpy(labels), torch.from_numpy(lens)


def timed_steps(step, n, warmup, use_dist):
    """W untimed steps, then EXACTLY n steps bracketed by barrier + synchronize; also the per-step times.  ``step`` may
    return a deferred loss (``Trainer.update(defer=True)``: the host reads step i's loss, the reference's one readback per
    step, codes/engine.py:92, after it has enqueued step i + 1): step i is then complete when call i + 1 returns, and the
    last one when its ``result()`` does -- inside the timed region."""
    for i in range(warmup):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    gc.collect()                  # (a generation-2 collection inside a 0.3 s timed region is a 5 % outlier: collect now, and
    gc.freeze()                   # keep what exists -- model, bins, modules -- out of the collections the timed steps trigger)
    stamps = []
    t0 = time.time()
    p0 = time.perf_counter()
    loss = None
    for i in range(warmup, warmup + n):
        loss = step(i)
        stamps.append(time.perf_counter())
    deferred = hasattr(loss, 'result')
    if deferred:
        loss = loss.result()
        stamps = stamps[1:] + [time.perf_counter()]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.time() - t0
    per = [b - a for a, b in zip([p0] + stamps[:-1], stamps)]
    return dt, per, loss


def secondary_shape(trainer, front, dev, bsz, bins_durations, steps=None, warmup=None, seed=900):
    """A few steps at another BASELINE shape on the same model / trainer; ``bins_durations``: one array of clip lengths
    per bin, visited round-robin.  One untimed pass over every bin first (each new shape grows the caching allocator's
    pools: +50 % on the first visit of a long B=32 bin), then ``steps`` timed steps.  Returns frames/s, ms/step and the
    whole-step fraction of the fp32-MFMA roof."""
    bins = [make_bin((seed + k, d)) for k, d in enumerate(bins_durations)]
    res = [make_resident(b, dev) for b in bins]
    steps = steps or len(bins)
    warmup = len(bins) if warmup is None else warmup

    def step(i):
        flat, offs, labels, lens = res[i % len(res)]
        inputs, pct = front(flat, offs)
        return trainer.update((inputs, labels, pct, lens), defer=False)        # the headline's protocol: a sync per step

    dt, per, _ = timed_steps(step, steps, warmup, False)
    idxs = [i % len(bins) for i in range(warmup, warmup + steps)]
    fr = float(sum(frames_of(bins[i]) for i in idxs))
    vsteps = float(sum(valid_out_steps_of(bins[i]) for i in idxs))
    tf = vsteps * TRAIN_FLOP_PER_OUT_STEP / dt / 1e12
    return {'frames_per_s': round(fr / dt, 1), 'ms_per_step': round(1e3 * dt / steps, 2),
            'whole_step_tflops': round(tf, 2), 'whole_step_frac_of_f32_mfma_peak': round(tf / PEAK_F32_MFMA_TFLOPS, 4)}


def loader_leg(trainer, plan, dev, workers=4):
    """End-to-end: the SAME bins as 16-bit WAV files on disk -> AudioDataset (ToTensor draws tempo + gain, hands on
    int16) -> DataLoader workers, page-locked batches -> DevicePrefetcher (upload one bin ahead) -> device decode +
    WSOLA tempo + gain + STFT -> training step.  frames/s of one pass over the bins (frames counted AFTER the tempo
    change, i.e. what the model sees), and of the loader + device frontend alone."""
    import shutil
    import tempfile
    import wave
    from codes.data import AudioDataLoader, AudioDataset, DevicePrefetcher
    from codes.sampler import BucketingSampler
    from codes.transforms import BatchSpectrogram, Compose, ToLabel, ToTensor
    tmp = tempfile.mkdtemp(prefix='ds2_bench_wavs_')
    try:
        rows, k = [], 0
        order = sorted(range(len(plan)), key=lambda i: frames_of_plan(plan[i]))     # a duration-sorted manifest
        for bi in order:
            wavs, labels, lens = make_bin(plan[bi])
            off = 0
            for w, n in zip(wavs, lens):
                with wave.open(os.path.join(tmp, 'u%d.wav' % k), 'wb') as f:
                    f.setnchannels(1)
                    f.setsampwidth(2)
                    f.setframerate(16000)
                    f.writeframes((w * 32767).astype('<i2').tobytes())
                txt = ''.join(chr(64 + int(c)) if c >= 3 else 'A' for c in labels[off:off + n])   # n label characters
                open(os.path.join(tmp, 'u%d.txt' % k), 'w').write(txt + '\n')
                off += n
                rows.append('u%d.wav,u%d.txt,%.4f' % (k, k, len(w) / 16000.0))
                k += 1
        open(os.path.join(tmp, 'm.csv'), 'w').write('\n'.join(rows) + '\n')
        ds = AudioDataset(tmp, os.path.join(tmp, 'm.csv'), Compose([ToTensor(augment=True, defer=True)]),
                          ToLabel(os.path.join(ROOT, 'data', 'labels.en.json')))
        bsz = len(plan[0][1])
        front = BatchSpectrogram(device=dev)
        out = {'workers': workers, 'clips': k, 'augment': 'tempo 0.85-1.15 (WSOLA) + gain -6..8 dB, on the device'}
        note('loader leg: %d wav files written' % k)
        sampler = BucketingSampler(ds, batch_size=bsz)
        loader = DevicePrefetcher(AudioDataLoader(ds, batch_sampler=sampler, raw_audio=True, num_workers=workers,
                                                  pin_memory=True, persistent_workers=workers > 0), dev, frontend=front)
        # pass 0 is untimed (it starts the worker processes); then one pass of the loader + device frontend alone and
        # two passes feeding the training step, each over all the bins in a fresh shuffled order
        for epoch, name in enumerate(('warmup', 'loader_and_frontend_only', 'train', 'train')):
            sampler.shuffle(epoch + 1)
            frames = padded = 0
            torch.cuda.synchronize()
            t0 = time.time()
            for inputs, targets, pct, sizes in loader:
                frames += int(round(float(pct.sum()) * inputs.shape[1]))
                padded += int(inputs.shape[0] * inputs.shape[1])           # what the step computes: B x T_max
                if name == 'train':
                    trainer.update((inputs, targets, pct, sizes), defer=True)
            trainer.flush()
            torch.cuda.synchronize()
            el = time.time() - t0
            rate = round(frames / el, 1)
            note('loader leg pass %d (%s): %s frames/s (%s padded)' % (epoch, name, rate, round(padded / el, 1)))
            if name != 'warmup' and rate > out.get(name + '_frames_per_s', 0.0):
                out[name + '_frames_per_s'] = rate
                out[name + '_padded_frames_per_s'] = round(padded / el, 1)      # the drawn tempo (0.85-1.15) un-sorts a bin
                out[name + '_padding_overhead'] = round(padded / max(frames, 1) - 1.0, 4)
        del loader
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch_command(ngpus, argv, port=None):
    """The command `bench.py --gpus N` (N > 1) runs when no launcher started it: one rank per GPU of this node through
    torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    if not (port or os.environ.get('MASTER_PORT')):
        # a free port of this host, so that runs launched back to back (N = 2, 4, 8 in a row) never meet a rendezvous port that
        # the previous run's store has not released yet
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    port = port or os.environ.get('MASTER_PORT')
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(ngpus, argv, runner=None):
    """Run the ranks as a child process (never an exec: this process must stay alive to return the code), hand the
    child's stdout -- rank 0's one JSON line -- through unchanged, return the child's exit code."""
    import subprocess
    cmd = self_launch_command(ngpus, argv)
    note('no launcher in the environment: starting %d ranks: %s' % (ngpus, ' '.join(cmd)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this pool's host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // ngpus)))
    proc = (runner or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE)
    out = proc.stdout.decode() if isinstance(proc.stdout, bytes) else (proc.stdout or '')
    sys.stdout.write(out)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--batch-size', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-full', action='store_true', help='(default since round 3) the complete CPU protocol')
    ap.add_argument('--cpu-quick', action='store_true', help='stop the CPU protocol after --cpu-budget-s of timed work')
    ap.add_argument('--cpu-budget-s', type=float, default=60.0)
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary shapes and the loader leg')
    ap.add_argument('--fixed-seconds', type=float, default=0.0,
                    help='profiling aid: every clip this long (e.g. --batch-size 64 --fixed-seconds 15 = the fixed worst case)')
    ap.add_argument('--protocol', choices=('driver', 'survey'), default='driver',
                    help="survey = SURVEY.md 8(d)'s literal protocol: 20 warm-up + 100 timed steps (sets --warmup / --steps); "
                         'the line always carries median / p10 / p90 of the per-step rates')
    ap.add_argument('--no-floor', action='store_true',
                    help="skip the recurrence-floor leg (ablated launches of the fault-injection library, which appear in a "
                         "kernel trace under the product kernels' names)")
    ap.add_argument('--no-f32-leg', action='store_true', help='skip the leg that repeats the steps on the f32-input GEMM kernels')
    args = ap.parse_args()
    if args.protocol == 'survey':
        args.warmup, args.steps = 20, 100

    # `python bench.py --gpus N` with N > 1 and no launcher around it (the way the driver runs `--gpus 1`): start the N
    # ranks ourselves, as a CHILD process -- this process has not touched the GPU and never will -- pass rank 0's one JSON
    # line through and leave with the child's return code (the reference: train.py:118-124 is started the same way)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly one line, the JSON result: RCCL prints a version banner to the C-level stdout (flushed at
    # exit, i.e. after our line), so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    from codes.utils.dist_utils import assert_no_fallbacks, data_parallel_env
    if world > 1 or os.environ.get('DS2_BENCH_FORCE_DIST') == '1':
        # hardware-queue count and RCCL channel cap: ONE function shared with train.py (codes/utils/dist_utils.py); read by
        # the runtime / RCCL when they start, so applied before the first torch.cuda call
        dp_env = data_parallel_env()
    else:
        dp_env = None
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X: the product path has no CPU fallback')
    # a benchmark must never report the launch-per-step fall-back's rate as the product's: a persistent recurrence launch
    # that cannot run (time-out, grid not co-resident) is an ERROR here, not a logged switch (ds2hip/ops.py)
    os.environ.setdefault('DS2_GRU_STRICT', '1')
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get('DS2_BENCH_FORCE_DIST') == '1'    # the latter: 1-rank RCCL group, for testing
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', init_method='env://')
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from codes.engine import Trainer
    from codes.model import DeepSpeech
    from codes.transforms import BatchSpectrogram

    bsz = args.batch_size
    plan = bin_plan(bsz, NUM_BINS * world, world=world)
    if args.fixed_seconds > 0:
        plan = [(seed, np.full(bsz, args.fixed_seconds)) for seed, _ in plan[:4 * world]]
    mine = [make_bin(p) for p in plan[rank::world]]                      # every world-th bin, starting from rank
    dev = torch.device('cuda', local)
    resident = [make_resident(b, dev) for b in mine]                      # inputs resident in HBM before timing

    torch.manual_seed(42)
    model = DeepSpeech().to(dev)                                          # 5 x BiGRU-800, A = 29, random init
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device=dev, max_norm=400)
    front = BatchSpectrogram(device=dev)

    # THE HEADLINE is SURVEY.md 8(d)'s literal protocol: the host synchronises at the end of EVERY step, as the reference's
    # codes/engine.py:92 does (round 3 reported the deferred-readback rate as the headline and this one beside it).
    def step_sync(i):
        flat, offs, labels, lens = resident[i % len(resident)]
        inputs, pct = front(flat, offs)
        return trainer.update((inputs, labels, pct, lens), defer=False)

    def step(i):       # the product's training loop (Trainer.run / train.py): the step's one readback deferred by one step
        flat, offs, labels, lens = resident[i % len(resident)]
        inputs, pct = front(flat, offs)
        return trainer.update((inputs, labels, pct, lens), defer=True)

    trainer.reserve(int(float(os.environ.get('DS2_BENCH_RESERVE_GB', '8')) * (1 << 30)))   # one allocator block up front (the longest
    # bin needs ~2 GB per step in flight, twice that while the side stream still holds the previous step's buffers)
    note('model built, inputs resident; timing')
    reserved0 = torch.cuda.memory_stats(dev).get('reserved_bytes.all.current', 0)
    dt, per, loss = timed_steps(step_sync, args.steps, args.warmup, use_dist)
    reserved_growth = torch.cuda.memory_stats(dev).get('reserved_bytes.all.current', 0) - reserved0   # > 0: a hipMalloc inside the timed region
    note('timed region done (a host synchronisation in every step): %.2f ms/step' % (1e3 * dt / args.steps))
    idxs = [i % len(mine) for i in range(args.warmup, args.warmup + args.steps)]
    frames = float(sum(frames_of(mine[i]) for i in idxs))
    osteps = float(sum(out_steps_of(mine[i]) for i in idxs))
    vsteps = float(sum(valid_out_steps_of(mine[i]) for i in idxs))
    pframes = float(sum(max(1 + len(w) // HOP for w in mine[i][0]) * len(mine[i][0]) for i in idxs))
    step_rates = sorted(frames_of(mine[i]) / t for i, t in zip(idxs, per))      # this rank's per-step frames/s
    per_sorted = sorted(per)
    if use_dist:
        t = torch.tensor([dt, frames, osteps, vsteps, pframes], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, frames, osteps, vsteps, pframes = float(tmax[0]), float(t[1]), float(t[2]), float(t[3]), float(t[4])

    # the same steps as the product's own loop runs them: each step's readback deferred by one step
    dt_defer, _, _ = timed_steps(lambda i: step(i + args.warmup), args.steps, 0, use_dist)
    if use_dist:
        ts = torch.tensor([dt_defer], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        dt_defer = float(ts[0])
    note('deferred-readback leg done: %.2f ms/step' % (1e3 * dt_defer / args.steps))

    # the same steps with every GEMM, conv2's forward pass and (B >= 17) the forward recurrence on the f32-input MFMA kernels
    # (ds2_gemm_split_mode 0, DS2_CONV_SPLIT=0, DS2_GRU_P2_BF16=0) instead of the default split-operand kernels on the bf16
    # matrix pipe: both families take fp32 operands and return fp32-accurate products
    # (include/ds2hip.h), this leg says what the choice of kernel family is worth
    from ds2hip import ops as _ops
    gemm_mode = _ops.gemm_split_mode()
    dt_f32 = None
    if gemm_mode != 0 and not args.no_f32_leg:
        _ops.gemm_split_mode(0)
        saved_env = {k: os.environ.get(k) for k in ('DS2_CONV_SPLIT', 'DS2_GRU_P2_BF16')}
        os.environ['DS2_CONV_SPLIT'] = '0'           # (read per call by the library: conv2 forward on the direct kernels,
        os.environ['DS2_GRU_P2_BF16'] = '0'          # the B >= 17 forward recurrence on the f32-input form)
        try:
            dt_f32, _, _ = timed_steps(lambda i: step(i + args.warmup - 2), args.steps, 2, use_dist)   # the same bins
        finally:
            _ops.gemm_split_mode(gemm_mode)
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if use_dist:
            ts = torch.tensor([dt_f32], dtype=torch.float64, device=dev)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            dt_f32 = float(ts[0])
        note('f32-MFMA GEMM leg done: %.2f ms/step' % (1e3 * dt_f32 / args.steps))

    # data-parallel self-diagnosis (every rank takes part): what the collective costs alone, what the step costs with the
    # per-layer all-reduce overlapped with backward and without, and whether any recurrence launch fell back
    ddp = None
    if use_dist:
        from ds2hip import ops as _ops
        gflat = model.flat_grad()
        for _ in range(2):
            dist.all_reduce(gflat)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dist.all_reduce(gflat)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5.0
        legs = {}
        for name, flag in (('overlap_1', True), ('overlap_0', False)):
            trainer.overlap = flag and trainer.distributed
            dleg, _, _ = timed_steps(step, 8, 2, True)
            legs[name] = dleg
        trainer.overlap = trainer.distributed and os.environ.get('DS2_ALLREDUCE_OVERLAP', '1') != '0'
        tl = torch.tensor([ar_ms, legs['overlap_1'], legs['overlap_0'], float(_ops.fallback_count)], dtype=torch.float64,
                          device=dev)
        tl_max = tl.clone()
        dist.all_reduce(tl_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        ddp = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(),
               'flat_gradient_bytes': int(gflat.numel() * 4),
               'allreduce_alone_ms': round(float(tl_max[0]), 3),
               'allreduce_alone_busbw_GBps': round(2.0 * (world - 1) / max(world, 1) * gflat.numel() * 4 / 1e6
                                                   / max(float(tl_max[0]), 1e-9), 1),
               'ms_per_step_overlap_1': round(1e3 * float(tl_max[1]) / 8, 3),
               'ms_per_step_overlap_0': round(1e3 * float(tl_max[2]) / 8, 3),
               'persistent_to_step_fallbacks': int(round(float(tl[3]))),
               'NCCL_MAX_NCHANNELS': os.environ.get('NCCL_MAX_NCHANNELS'),
               'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES')}
        note('data-parallel diagnostics done: %s' % json.dumps(ddp))
        # the --gpus N self-check: a fall-back anywhere FAILS the run (every rank raises: the count is the all-reduced sum)
        assert_no_fallbacks(ddp['persistent_to_step_fallbacks'], 'bench.py --gpus %d' % world)

    # BASELINE configs[3] at N > 1: 64 x 15 s over 8 GPUs = 8 clips of 15 s per rank (every rank, same barrier protocol)
    cfg3 = None
    if world > 1 and not args.no_extras:
        c3 = [make_resident(make_bin((7000 + rank * 10 + k, np.full(8, 15.0))), dev) for k in range(2)]

        def step3(i):
            flat, offs, labels, lens = c3[i % 2]
            inputs, pct = front(flat, offs)
            return trainer.update((inputs, labels, pct, lens), defer=True)

        dt3, _, _ = timed_steps(step3, 8, 2, use_dist)
        t3 = torch.tensor([dt3], dtype=torch.float64, device=dev)
        dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        cfg3 = {'workload': 'BASELINE configs[3]: 8 x 15 s per GPU (64 x 15 s at 8 GPUs)', 'steps': 8,
                'frames_per_s': round(8 * 8 * 1501 * world / float(t3[0]), 1),
                'ms_per_step': round(1e3 * float(t3[0]) / 8, 2)}
    if rank != 0:
        dist.destroy_process_group()
        return

    value = frames / dt

    # secondary metric: inference = frontend + eval forward + greedy decode, same bins, rank 0 only
    from codes.decoder import GreedyDecoder
    decoder = GreedyDecoder(['_', ' ', "'"] + [chr(65 + i) for i in range(26)])
    model.eval()
    inf_frames, n_inf = 0, min(12, len(resident))
    with torch.no_grad():
        for rep in range(2):                         # pass 0 is the warm-up (every bin is a new shape for the allocator)
            inf_frames = 0
            torch.cuda.synchronize()
            ti = time.time()
            for i in range(n_inf):
                inputs, pct = front(resident[i][0], resident[i][1])
                probs = model(inputs)
                sizes = (pct * probs.shape[1]).int()
                decoder.decode(probs, sizes)
                inf_frames += frames_of(mine[i])
            torch.cuda.synchronize()
            inf_dt = time.time() - ti
    model.train()
    step_tflops = vsteps * TRAIN_FLOP_PER_OUT_STEP / dt / 1e12 / world           # per GPU, from the utterances' OWN steps
    step_tflops_padded = osteps * TRAIN_FLOP_PER_OUT_STEP / dt / 1e12 / world    # what was computed (bins padded to the longest)
    t_mean = int(round(np.mean([out_steps_of(b) / len(b[0]) for b in mine])))
    note('inference leg done')
    roof = gru_pass_roofline(model, bsz, t_mean)
    from codes.model import _BWD_SPARE_CUS
    floor = None if args.no_floor else recurrence_floor(bsz, t_mean, model._rnn_hidden_size, dev, _BWD_SPARE_CUS)
    gemm_roof = gemm_roofline(model, t_mean * bsz)
    note('roofline legs done')
    mean_bin = min(range(len(mine)), key=lambda i: abs(out_steps_of(mine[i]) / len(mine[i][0]) - t_mean))
    share = chain_time_share(model, trainer, front, resident[mean_bin])
    note('chain time share: %s' % json.dumps(share['phases_ms']))
    parity = None
    if not args.no_cpu_baseline and world == 1 and args.fixed_seconds <= 0:
        longest = max(range(len(plan)), key=lambda i: frames_of_plan(plan[i]))
        parity = gpu_parity_side(model, trainer, front, decoder, plan[longest], dev)
        parity['labels_txt'] = ['_', ' ', "'"] + [chr(65 + i) for i in range(26)]
        note('parity leg (HIP side) done: loss_sum %.4f' % parity['loss_sum'])
    ach, dur, flop = roof['bwd']
    # HBM-side bytes per launch of the dominant kernel come from a SEPARATE rocprofv3 --pmc run of the same shape whose
    # summary is committed under profiles/ (PMC collection cannot run inside this process); the file is named below
    traffic, traffic_src, traffic_floor, traffic_alg = None, None, None, None
    for name in ('r06_traffic.json', 'r05_traffic.json'):
        try:
            rec = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if rec['shape'] == {'T': t_mean, 'B': bsz, 'H': 800}:
                traffic = rec['backward_recurrence_launch']['traffic_bytes_per_launch']
                traffic_floor = rec['backward_recurrence_launch'].get('xcd_replicated_floor_bytes_per_launch')
                traffic_alg = rec['backward_recurrence_launch'].get('algorithmic_hbm_bytes_per_launch')
                traffic_src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gru_step_timing.py, ' \
                              'not measured in this run)' % name
                break
        except (OSError, KeyError, ValueError):
            pass
    pct_of = lambda v, q: v[min(len(v) - 1, int(round(q * (len(v) - 1))))]          # noqa: E731
    result = {
        'metric': 'train frames/sec, DeepSpeech2 5xBiGRU-800',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: librispeech-from_scratch.json, 5xBiGRU-800 A=29, batch %d per GPU, '
                               '16 kHz clips %s, full train step incl. GPU STFT frontend, CTC, clip+SGD'
                               % (bsz, 'uniform 1-15 s in length-sorted bins' if args.fixed_seconds <= 0
                                  else 'all %.1f s long' % args.fixed_seconds),
                   'batch_per_gpu': bsz, 'global_batch': bsz * world, 'parallelism': 'dp%d' % world,
                   'last_loss': round(float(loss), 4),
                   'protocol': 'value / ms_per_step: a host synchronisation at the end of EVERY step (codes/engine.py:92, SURVEY.md '
                               '8d); frames = valid spectrogram frames 1 + L // 160 of every clip (padding not counted)',
                   'protocol_name': args.protocol + (' (SURVEY.md 8d: 20 warm-up + 100 timed steps; median / p10 / p90 in '
                                                     'frames_per_s_per_step_rank0 and ms_per_step_rank0)' if args.protocol == 'survey'
                                                     else " (the driver's command line: --steps / --warmup as given)"),
                   'padded_frames_per_s': round(pframes / dt, 1),
                   'padding_note': 'a bin is padded to its longest clip (codes/data.py:132-152): B x T_max frames are computed per '
                                   'step, the valid ones are counted',
                   'deferred_readback': {'frames_per_s': round(frames / dt_defer, 1),
                                         'ms_per_step': round(1e3 * dt_defer / args.steps, 3),
                                         'note': 'the same steps as Trainer.run / train.py run them: each step\'s one readback '
                                                 'deferred by one step (round 3\'s headline)'},
                   'chain_time_share': share,
                   'gemm_arithmetic': {
                       'mode': gemm_mode,
                       'note': ('fp32 operands, fp32 accumulator, fp32 result; products on the bf16 matrix pipe after an '
                                'error-free three-way split of every operand element (%d partial products, each exact; '
                                'what is left out is below one fp32 rounding of the product: 2^-24.2 at most and 6e-9 rms over 1e6 random products, a '
                                'correctly rounded fp32 multiply: 2^-24 and 2.5e-8): error against fp64 equal to the f32-input MFMA '
                                'kernels\' (tests/test_kernels_gpu.py, tools/gemm_split_check.py)' % gemm_mode)
                               if gemm_mode else 'f32-input MFMA kernels (v_mfma_f32_32x32x2_f32)',
                       'kernels_on_the_bf16_pipe': 'GEMMs, conv2 forward, forward recurrence from B = 17',
                       'same_steps_on_f32_input_mfma_gemms': None if dt_f32 is None else {
                           'frames_per_s': round(frames / dt_f32, 1), 'ms_per_step': round(1e3 * dt_f32 / args.steps, 3),
                           'note': 'deferred readback: compare with config.deferred_readback'}},
                   'persistent_to_step_fallbacks': (_fallbacks() if ddp is None
                                                    else ddp['persistent_to_step_fallbacks']),
                   'ms_per_step_rank0': {'median': round(1e3 * pct_of(per_sorted, 0.5), 3),
                                         'p10': round(1e3 * pct_of(per_sorted, 0.1), 3),
                                         'p90': round(1e3 * pct_of(per_sorted, 0.9), 3),
                                         'max': round(1e3 * per_sorted[-1], 3),
                                         'per_step': [round(1e3 * v, 2) for v in per],
                                         'allocator_growth_bytes_in_timed_region': int(reserved_growth)},
                   'frames_per_s_per_step_rank0': {'median': round(pct_of(step_rates, 0.5), 1),
                                                   'p10': round(pct_of(step_rates, 0.1), 1),
                                                   'p90': round(pct_of(step_rates, 0.9), 1)},
                   'inference_frames_per_s_rank0': round(inf_frames / inf_dt, 1)},
        'roofline': {'bound': 'latency' if bsz <= 16 else 'mfma',
                     'bound_note': 'one launch = T dependent time steps; a step is a chain of three phases on every CU -- the '
                                   'inter-CU hand-off of the new state through memory, the matrix phase at the fp32 MFMA rate '
                                   'of the CUs it occupies, and the gate / reduction skeleton (DESIGN.md section 6).  achieved / '
                                   'peak / frac price the launch against the fp32-input MFMA roof (the contract\'s yardstick); '
                                   'floor_us_per_step is what the latency model allows',
                     'kernel': 'the backward recurrence launch: gru_bwd_persistent6_kernel<25, 7> (d(h) hand-off, 28 units per workgroup) '
                               'under the layers below the top one, <25, 5> (20 units) under the top layer (one launch = all T=%d steps '
                               'of a BiGRU layer, both directions, B=%d)' % (t_mean, bsz),
                     'achieved': round(ach, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
                     'traffic_algorithmic': traffic_alg,
                     'traffic_floor_of_an_exchange_through_memory': traffic_floor,
                     'traffic_note': 'bytes at the memory side per launch (2 x FETCH_SIZE + WRITE_SIZE).  traffic_algorithmic: the saved '
                                     'activations once (an exchange that stayed on chip would move nothing else).  An exchange through memory '
                                     'between CUs on EIGHT XCDs with non-coherent L2s cannot go below the payload fetched once into '
                                     'EVERY XCD (traffic_floor_...): the ratio to that, not to the algorithmic bytes, is the waste',
                     'avg_launch_us': round(dur * 1e6, 1), 'us_per_time_step': round(dur * 1e6 / t_mean, 3),
                     'floor_us_per_step': floor,
                     'flop_per_launch': flop,
                     'launch_forms_us': {
                         'top layer (nothing queued beside it: the widest grid)': round(roof['bwd_top'][1] * 1e6, 1),
                         'layers below (CUs left to the weight-gradient GEMMs of the layer above)': round(roof['bwd_below'][1] * 1e6, 1),
                         'note': 'avg_launch_us = the mean launch of a step (1 : num_layers - 1), each form timed stand-alone; '
                                 'ds2_gru_bidir_bwd_persistent_ex, include/ds2hip.h'},
                     'fwd_kernel_tflops': round(roof['fwd'][0], 3),
                     'fwd_us_per_time_step': round(roof['fwd'][1] * 1e6 / t_mean, 3),
                     'whole_step_tflops_per_gpu': round(step_tflops, 3),
                     'whole_step_frac_of_f32_mfma_peak': round(step_tflops / PEAK_F32_MFMA_TFLOPS, 5),
                     'whole_step_tflops_per_gpu_padded': round(step_tflops_padded, 3),
                     'whole_step_note': 'FLOP = 261.94 M x the utterances\' own output steps (valid frames); _padded counts the '
                                        'steps computed (every clip padded to its bin\'s longest)',
                     'gemm': gemm_roof},
    }
    if cfg3 is not None:
        result['config']['config3'] = cfg3
    if ddp is not None:
        ddp['environment'] = dp_env          # queue / channel settings and the CPU placement applied (codes/utils/dist_utils.py)
        result['config']['data_parallel'] = ddp
    if world == 1 and not args.no_extras:
        # the other single-GPU BASELINE shapes on the same model, a few steps each
        p32 = bin_plan(32, 8, seed=43)                      # a length-sorted corpus cut into bins of 32, as BucketingSampler does
        result['config']['other_shapes'] = {
            'B32_1to15s (configs[2])': secondary_shape(trainer, front, dev, 32, [p[1] for p in p32]),
            'B8_x_15s (configs[3] per GPU)': secondary_shape(trainer, front, dev, 8, [np.full(8, 15.0)] * 2),
            'B64_x_15s (fixed worst case)': secondary_shape(trainer, front, dev, 64, [np.full(64, 15.0)] * 2, steps=4),
        }
        note('other shapes done')
        result['config']['loader'] = loader_leg(trainer, plan, dev)
        note('loader leg done')
    if not args.no_cpu_baseline and world == 1:      # the CPU oracle is timed beside the GPU at N = 1 only
        result['cpu_baseline'] = cpu_baseline(plan, budget_s=args.cpu_budget_s, full=not args.cpu_quick, parity=parity)
        if 'parity_vs_cpu_oracle' in result['cpu_baseline']:
            result['parity_vs_cpu_oracle'] = result['cpu_baseline'].pop('parity_vs_cpu_oracle')
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(result) + '\n').encode())
    os.close(result_fd)


if __name__ == '__main__':
    main()
