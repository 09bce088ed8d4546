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
    ADJACENT bins are then shuffled as units, so that the bins the `world` ranks consume in the same step under the
    every-world-th rule (codes/sampler.py:119-125) hold clips of similar length (minimises straggling).
    Audio is only synthesised for the bins a process actually uses (make_bin)."""
    rng = np.random.default_rng(seed)
    dur = np.sort(rng.uniform(1.0, 15.0, size=batch_size * num_bins))
    bins = [(seed * 100003 + i, dur[i * batch_size:(i + 1) * batch_size]) for i in range(num_bins)]
    groups = np.random.default_rng(seed + 1).permutation(num_bins // world)   # epoch-2-style shuffled bins
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


def cpu_baseline(plan, budget_s=25.0):
    """The oracle (stock PyTorch CPU ops, numerically the reference) timed on the host cores: one bounded
    training step (frontend -> fwd -> CTC -> bwd -> clip -> SGD) on the shortest bin(s)."""
    import torch.nn.functional as F
    from oracle import spectrogram as ospec
    from oracle.model import OracleDeepSpeech
    torch.manual_seed(0)
    model = OracleDeepSpeech()
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    model.train()
    order = sorted(range(len(plan)), key=lambda i: frames_of_plan(plan[i]))
    bins = {i: make_bin(plan[i]) for i in order[:3]}
    frames, secs, used = 0, 0.0, []
    for idx in order[:3]:
        wavs, labels, lens = bins[idx]
        t0 = time.time()
        x, pct = ospec.batch_log_spectrogram(wavs)
        logits = model(torch.from_numpy(x))
        out_sizes = (torch.from_numpy(pct) * logits.shape[1]).int()
        loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(labels).long(), out_sizes.long(),
                          torch.from_numpy(lens).long(), blank=0, reduction='sum') / len(wavs)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 400)
        opt.step()
        dt = time.time() - t0
        if used or len(order) == 1:            # the first step is the warm-up unless it is all we can afford
            frames += frames_of(bins[idx])
            secs += dt
        used.append(idx)
        if secs + dt > budget_s and frames > 0:
            break
        if dt > budget_s:                       # even the warm-up blew the budget: count it
            frames, secs = frames_of(bins[idx]), dt
            break
    return {'value': round(frames / secs, 1), 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'oracle (torch CPU conv/BN/GRU/Linear + F.ctc_loss + clip + SGD) full training step on the '
                      '%d shortest bins of the same workload (B=10; first step = warm-up, untimed), %d frames in %.1f s'
                      % (len(used), frames, secs)}


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
    res = {'fwd': [], 'bwd': []}
    for _ in range(5):
        g = gates.clone()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ghn, hout = ops.gru_bidir_fwd(g, w_hh, t, bsz, hid)
        e[1].record()
        ops.gru_bidir_bwd(g, ghn, hout, d_out, w_hh_t, t, bsz, hid)
        e[2].record()
        torch.cuda.synchronize()
        ops.check_async_errors()
        res['fwd'].append(e[0].elapsed_time(e[1]) * 1e-3)
        res['bwd'].append(e[1].elapsed_time(e[2]) * 1e-3)
    flop = 2.0 * 2 * bsz * hid * 3 * hid * t                           # both directions, all T steps
    out = {}
    for name in ('fwd', 'bwd'):
        dur = float(np.median(res[name]))
        out[name] = (flop / dur / 1e12, dur, flop)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--batch-size', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: RCCL prints a version banner to the C-level stdout (flushed at
    # exit, i.e. after our line), so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('DS2_BENCH_FORCE_DIST') == '1':
        # With a process group there are more streams (the all-reduce stream, RCCL's own) than the HIP runtime's default
        # four hardware queues serve well: measured on one rank, the step loses 12 % (308 k vs 352 k frames/s) with the
        # default and nothing with two or three queues.  Read by the runtime at its first call, so set before any.
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '3')
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get('DS2_BENCH_FORCE_DIST') == '1'    # the latter: 1-rank RCCL group, for testing
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # The recurrence kernels need ~204 of the 256 CUs co-resident; cap RCCL's kernels at 32 workgroups so a gradient
        # all-reduce running beside them (on the 52 CUs they leave free) can never keep one from starting.
        os.environ.setdefault('NCCL_MAX_NCHANNELS', '32')
        dist.init_process_group('nccl', init_method='env://')
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    from codes.engine import Trainer
    from codes.model import DeepSpeech
    from codes.transforms import BatchSpectrogram

    bsz = args.batch_size
    plan = bin_plan(bsz, NUM_BINS * world, world=world)
    mine = [make_bin(p) for p in plan[rank::world]]                      # every world-th bin, starting from rank
    dev = torch.device('cuda', local)
    resident = []
    for wavs, labels, lens in mine:                                       # inputs resident in HBM before timing
        flat = torch.from_numpy(np.concatenate(wavs)).to(dev)
        offs = np.concatenate([[0], np.cumsum([len(w) for w in wavs])]).astype(np.int64)
        resident.append((flat, offs, torch.from_numpy(labels), torch.from_numpy(lens)))

    torch.manual_seed(42)
    model = DeepSpeech().to(dev)                                          # 5 x BiGRU-800, A = 29, random init
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device=dev, max_norm=400)
    front = BatchSpectrogram(device=dev)

    def step(i):
        flat, offs, labels, lens = resident[i % len(resident)]
        inputs, pct = front(flat, offs)
        return trainer.update((inputs, labels, pct, lens))

    for i in range(args.warmup):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.time() - t0
    idxs = [i % len(mine) for i in range(args.warmup, args.warmup + args.steps)]
    frames = float(sum(frames_of(mine[i]) for i in idxs))
    osteps = float(sum(out_steps_of(mine[i]) for i in idxs))
    if use_dist:
        t = torch.tensor([dt, frames, osteps], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, frames, osteps = float(tmax[0]), float(t[1]), float(t[2])
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
        for i in range(2):                                               # warm-up
            inputs, pct = front(resident[i][0], resident[i][1])
            model(inputs)
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
    step_tflops = osteps * TRAIN_FLOP_PER_OUT_STEP / dt / 1e12 / world    # per GPU, padded steps included
    t_mean = int(round(np.mean([out_steps_of(b) / len(b[0]) for b in mine])))
    roof = gru_pass_roofline(model, bsz, t_mean)
    ach, dur, flop = roof['bwd']
    traffic = None                      # HBM bytes per launch from a committed rocprofv3 --pmc run of the same shape
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')))
        if rec['shape'] == {'T': t_mean, 'B': bsz, 'H': 800}:
            traffic = rec['gru_bwd_persistent4_kernel']['traffic_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass
    result = {
        'metric': 'train frames/sec, DeepSpeech2 5xBiGRU-800',
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: librispeech-from_scratch.json, 5xBiGRU-800 A=29, batch %d per GPU, '
                               '16 kHz clips uniform 1-15 s in length-sorted bins, full train step incl. GPU STFT '
                               'frontend, CTC, clip+SGD' % bsz,
                   'batch_per_gpu': bsz, 'global_batch': bsz * world, 'parallelism': 'dp%d' % world,
                   'last_loss': round(float(loss), 4),
                   'inference_frames_per_s_rank0': round(inf_frames / inf_dt, 1)},
        'roofline': {'bound': 'mfma',
                     'kernel': 'gru_bwd_persistent4_kernel (one launch = all T=%d steps of a BiGRU layer, both '
                               'directions, B=%d)' % (t_mean, bsz),
                     'achieved': round(ach, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 5), 'traffic': traffic,
                     'avg_launch_us': round(dur * 1e6, 1), 'us_per_time_step': round(dur * 1e6 / t_mean, 3),
                     'flop_per_launch': flop,
                     'fwd_kernel_tflops': round(roof['fwd'][0], 3),
                     'fwd_us_per_time_step': round(roof['fwd'][1] * 1e6 / t_mean, 3),
                     'whole_step_tflops_per_gpu': round(step_tflops, 3),
                     'whole_step_frac_of_f32_mfma_peak': round(step_tflops / PEAK_F32_MFMA_TFLOPS, 5)},
    }
    if not args.no_cpu_baseline and world == 1:      # the CPU oracle is timed beside the GPU at N = 1 only
        result['cpu_baseline'] = cpu_baseline(plan)
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(result) + '\n').encode())
    os.close(result_fd)


if __name__ == '__main__':
    main()
