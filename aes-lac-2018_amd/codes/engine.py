"""Training / evaluation step (reference ``codes/engine.py``), without ignite.

``create_trainer(model, optimizer, criterion, device, **kwargs)`` keeps the reference's signature and
step semantics (``codes/engine.py:35-100``):
    out = model(inputs); loss = criterion(out^T, targets, (pct*T).int(), sizes) / B; +-inf -> 0;
    zero_grad; backward; clip_grad_norm_(max_norm); optimizer.step(); synchronize; return loss.item()
but the whole step is HIP kernels on flat buffers: one fused forward/backward pass (no autograd graph),
the CTC gradient already scaled by 1/B, one sum-of-squares pass and one fused clip + Nesterov-SGD pass that
derives the clip coefficient on the device.  With ``torch.distributed`` initialised (one process per GPU,
RCCL over xGMI) the flat gradient is sum-all-reduced in per-layer buckets on a side stream while backward
is still running, and the 1/world average folds into the update kernel; BatchNorm statistics stay
per-replica, as under the reference's DDP (``train.py:172-175``).

The optimizer object is the reference's ``torch.optim.SGD`` (built from the JSON config, with any number of
parameter groups -- the per-layer learning rates of ``training_utils.py:133-159`` -- and with frozen parameters left
out): learning rate, momentum and the LR scheduler keep working through ``optimizer.param_groups``; its momentum buffers are
views of the trainer's flat momentum buffer so ``optimizer.state_dict()`` checkpoints as before.  Any
other optimizer falls back to ``loss.backward()`` through the model's autograd node + ``optimizer.step()``.
"""
import logging
import os
import time

import torch
import torch.distributed as dist

from ds2hip import ops

from .ctc import ctc_costs_and_grad
from .data import wait_ready

LOG = logging.getLogger('aes-lac-2018')


def sanitize_inputs(out_seq_length, input_percentages):
    """(input_percentages * T).int(): float32 multiply then truncation -- codes/engine.py:12-16."""
    return (input_percentages.to('cpu', torch.float32) * out_seq_length).int()


_STATS_DIRECT = os.environ.get('DS2_STATS_DIRECT', '1') != '0'


class PendingLoss(object):
    """The host half of a training step whose device work is already enqueued (``Trainer.update(defer=True)``)."""

    def __init__(self, trainer, host, done, bsz, scale):
        self.trainer, self.host, self.done, self.bsz, self.scale = trainer, host, done, bsz, scale
        self.stream = torch.cuda.current_stream() if done is None else None   # the stream the step was queued on
        self._value = None

    def result(self):
        if self._value is None:
            if self.done is None:                 # the kernel wrote the slot itself: poll the memory
                loss_sum, sumsq, timed_out, n_inf = ops.wait_step_stats(self.host, stream=self.stream)
            else:
                ops.spin_wait(self.done)
                loss_sum, sumsq, timed_out, n_inf = self.host.tolist()
            if timed_out != 0:
                ops.raise_async_error()
            self.trainer.last_grad_norm = float(sumsq) ** 0.5 * self.scale
            loss_v = float(loss_sum) / self.bsz
            if n_inf != 0 or loss_v in (float('inf'), float('-inf')):
                # codes/engine.py:27-30: the loss becomes 0 * loss -- reported as 0, and no utterance of the batch
                # contributed a gradient (the CTC kernel zeroed it); the update ran on momentum alone, as
                # optimizer.step() does
                LOG.warning('WARNING: received an inf loss, setting loss value to 0')
                loss_v = 0.0
            self._value = loss_v
        return self._value

    def __float__(self):
        return float(self.result())


class Trainer(object):
    def __init__(self, model, optimizer, criterion=None, device='cuda', max_norm=400, skip_n=0, frontend=None,
                 overlap_allreduce=True):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.device = torch.device(device)
        self.max_norm = max_norm
        self.skip_n = skip_n
        self.frontend = frontend
        self.iteration = 0
        self.data_time = 0.0
        self.last_grad_norm = None
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.overlap = overlap_allreduce and self.distributed and os.environ.get('DS2_ALLREDUCE_OVERLAP', '1') != '0'
        self._comm_stream = None
        self._buf = None
        self._first = True
        self._sumsq = None
        self._stats = None
        self._pending = None
        self._host_stats = None
        model._ensure_flat()
        if self.distributed:                     # DDP construction: rank 0's parameters and buffers win
            dist.broadcast(model._flat_p, 0)
            for b in model.buffers():
                if b.is_cuda:                    # (num_batches_tracked counters live on the host)
                    dist.broadcast(b, 0)
        self._fused = self._fused_ok()
        if self._fused:
            self._bind_momentum()
        self._bound_flat = model._flat_p           # the spans, the momentum views and the optimizer state are tied to THIS buffer

    def reserve(self, nbytes, small_blocks=32):
        """Grow the caching allocator's pool by one block of ``nbytes`` now, so that the first minibatch of every new
        (longer) shape carves its activations out of it instead of paying a ``hipMalloc`` per tensor inside the step.
        ``small_blocks``: the same for the allocator's SMALL pool (requests up to 1 MB live in 2 MB blocks of their own,
        which the big block does not serve): a minibatch shape whose mix of small buffers needs one block more than any
        shape before it costs a ``hipMalloc`` -- a device synchronisation, ~19 ms seen inside a 9 ms step -- wherever in the
        run it first appears (the driver-style bench run: 20 timed steps, one of them 27.5 instead of 8.7 ms, 0.9 ms on the
        mean)."""
        torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        small = [torch.empty(1 << 20, dtype=torch.uint8, device=self.device) for _ in range(2 * int(small_blocks))]
        del small

    # ---------------------------------------------------------------- optimizer plumbing
    def _fused_ok(self):
        """The fused clip + Nesterov-SGD pass covers ``torch.optim.SGD`` with any number of parameter groups (the
        reference's per-layer learning rates, ``training_utils.py:133-159``) as long as every group is Nesterov
        momentum without dampening or weight decay; frozen parameters (fine-tuning with ``freeze_layers``) are simply
        left out of the update.  ``self._spans`` = [(lo, hi, group index)] runs of the flat buffer to update."""
        opt = self.optimizer
        if not isinstance(opt, torch.optim.SGD):
            return False
        m = self.model
        where = {id(p): (o, p.numel()) for p, o in zip(m._plist, m._offsets)}
        spans, seen = [], set()
        for gi, g in enumerate(opt.param_groups):
            if not (g.get('nesterov', False) and g.get('momentum', 0) > 0 and g.get('dampening', 0) == 0 and
                    g.get('weight_decay', 0) == 0):
                return False
            for p in g['params']:
                if id(p) not in where or id(p) in seen:
                    return False
                seen.add(id(p))
                if not p.requires_grad:
                    continue
                o, n = where[id(p)]
                spans.append((o, o + (n + 3) // 4 * 4, gi))    # parameters are padded to 4 floats in the flat buffer
        if any(p.requires_grad and id(p) not in seen for p in m._plist):
            return False                                         # a trainable parameter the optimizer does not know
        merged = []
        for lo, hi, gi in sorted(spans):                         # flat order; adjacent runs of one group become one launch
            hi = min(hi, m._flat_p.numel())
            if merged and merged[-1][2] == gi and merged[-1][1] == lo:
                merged[-1] = (merged[-1][0], hi, gi)
            else:
                merged.append((lo, hi, gi))
        self._spans = merged
        return len(merged) > 0

    def _bind_momentum(self):
        """Flat momentum buffer; the torch optimizer's per-parameter state entries are views of it."""
        m = self.model
        self._buf = torch.zeros_like(m._flat_p)
        loaded = False
        for p, o in zip(m._plist, m._offsets):
            view = self._buf[o:o + p.numel()].view(p.shape)
            st = self.optimizer.state[p]
            old = st.get('momentum_buffer', None)
            if old is not None:                  # resumed optimizer state (train.py:142-150)
                view.copy_(old)
                loaded = True
            st['momentum_buffer'] = view
        self._first = not loaded

    # ---------------------------------------------------------------- one optimisation step
    def update(self, batch, defer=False):
        """One optimisation step (``codes/engine.py:19-101``); returns the minibatch loss.

        ``defer=True`` returns a ``PendingLoss`` instead: the step's device work is enqueued, the host does NOT wait for
        its one readback (loss sum, gradient norm, time-out flags) until the NEXT call to ``update`` has enqueued the next
        step (or ``result()`` is called).  The values are the same; the GPU no longer idles through the host's round trip
        and the next step's first launches (0.3 ms of a 20 ms step at B = 10)."""
        if self.skip_n > 0:                      # codes/engine.py:46-49 (resume mid-epoch)
            self.skip_n -= 1
            return 'Skipped'
        model = self.model
        # ``model.train()`` at the top of every step (codes/engine.py:51) -- which also returns the BatchNorm modules that
        # ``_freeze_layers`` put in eval mode to training mode.  The recursive call costs ~60 us of host time with the GPU idle;
        # the flags it would change are looked at instead (the model and its seven BatchNorm modules)
        bns = self.__dict__.get('_bn_modules')
        if bns is None:
            bns = self._bn_modules = [m for m in model.modules() if hasattr(m, 'num_batches_tracked')]
        if not model.training or not all(m.training for m in bns):
            model.train()
        model._ensure_flat()        # the re-pack is lazy: run its (cheap) check NOW, so that a swap just before this call is seen
        if model._flat_p is not self._bound_flat:
            # (a sub-module or parameter was replaced after this Trainer was built -- fine-tuning surgery belongs BEFORE the
            # optimizer and the trainer are created, as train.py does it: the flat spans of the fused update, the momentum
            # views and the optimizer's parameter list all point at the old buffer)
            raise RuntimeError('the model\'s parameters were re-packed after this Trainer was built (a swapped sub-module or '
                               're-assigned parameter): create the optimizer and the Trainer after the last change to the model')
        t0 = time.time()
        inputs, targets, input_percentages, target_sizes = batch
        if self.frontend is not None and not isinstance(inputs, torch.Tensor):   # list of clips or a RawAudioBatch
            inputs, input_percentages = self.frontend(inputs)          # raw clips -> device spectrograms
        inputs = wait_ready(inputs).to(self.device, non_blocking=True)
        self.data_time = time.time() - t0
        bsz = inputs.shape[0]

        if not self._fused:
            return self._update_autograd(inputs, targets, input_percentages, target_sizes)

        def loss_fn(acts):
            out_sizes = sanitize_inputs(acts.shape[0], input_percentages)
            costs, d_acts = ctc_costs_and_grad(acts, targets, out_sizes, target_sizes, grad_scale=1.0 / bsz,
                                               zero_batch_if_inf=True)      # codes/engine.py:24-30
            return costs, d_acts

        hook = self._bucket_hook() if self.overlap else None
        costs, _ = self._forward_backward(inputs, loss_fn, hook)
        gflat = model.flat_grad()
        if self.distributed:
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            else:
                dist.all_reduce(gflat)
        scale = 1.0 / self.world
        self._sumsq = ops.sumsq(gflat, self._sumsq)
        for lo, hi, gi in self._spans:                              # one run per parameter group (one in all, usually)
            g = self.optimizer.param_groups[gi]
            ops.clip_sgd_nesterov(model._flat_p[lo:hi], gflat[lo:hi], self._buf[lo:hi], self._sumsq, scale,
                                  self.max_norm, g['lr'], g['momentum'], self._first)
        self._first = False
        model._tick('gradient norm + clip + Nesterov SGD')
        # one launch gathers what the host needs (loss sum, grad norm^2, sticky kernel-timeout flags, inf count),
        # one device->host copy brings it over
        if self._host_stats is None:
            self._host_stats = [torch.empty(4, dtype=torch.float64).pin_memory() for _ in range(2)]
        slot = self._host_stats[self.iteration & 1]                       # two page-locked slots: a deferred readback
        if _STATS_DIRECT:                                                 # survives the next step's
            # the kernel writes the slot in host memory itself and the host polls it: no copy, no event (DS2_STATS_DIRECT=0
            # restores them)
            ops.arm_step_stats(slot)
            ops.step_stats(costs, self._sumsq, slot)
            done = None
        else:
            self._stats = ops.step_stats(costs, self._sumsq, self._stats)
            slot.copy_(self._stats.reshape(-1), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        self.iteration += 1
        pend = PendingLoss(self, slot, done, bsz, scale)
        prev, self._pending = self._pending, pend
        if prev is not None:
            prev.result()                                                 # the previous step's sync, one step late
        if defer:
            return pend
        self._pending = None
        return pend.result()                                              # the step's one sync (codes/engine.py:92)

    def flush(self):
        """Resolve a deferred step's readback (call before reading parameters on the host or ending a run)."""
        prev, self._pending = self._pending, None
        return prev.result() if prev is not None else None

    def _forward_backward(self, inputs, loss_fn, hook):
        m = self.model
        state = {}
        # ONE fill of the whole flat gradient, on the side stream beside the forward pass (the step's first side-stream work:
        # everything the backward pass puts there, and the main stream through the w_hh_t_ready event, is ordered behind
        # it), instead of a fill in front of every split-K weight-gradient GEMM
        prezero = os.environ.get('DS2_PREZERO', '1') != '0'

        def fill():
            # (issued by the forward pass right BEHIND its conv-block launches, not in front of them: after a step that ended
            # with a synchronisation the GPU idles until the first long kernel is enqueued, and the stream switch + fill cost
            # the host ~25 us there.  Still ahead of everything that writes the gradient: the side stream's own work follows
            # it in stream order, the main stream's through the w_hh_t_ready event recorded after it.)
            # (the check that the parameters are still views of the flat buffers belongs here too: the conv block reads its
            # weights through the modules, everything that relies on the flat layout -- the paired GRU weights, the gradients,
            # the optimizer -- comes after this point)
            gflat = state['gflat'] = m.flat_grad()           # (checks the flat layout itself)
            main = torch.cuda.current_stream()
            side = m._side_stream(gflat.device) if m.overlap_wgrad else None
            if side is not None:
                side.wait_stream(main)
            if prezero:
                with torch.cuda.stream(side if side is not None else main):
                    gflat.zero_()

        acts, sv = m._forward_impl(inputs.contiguous().float(), training=True, need_grad=True, after_conv=fill)
        costs, d_acts = loss_fn(acts)
        m._backward_impl(sv, d_acts, state['gflat'], grad_ready=hook, prezeroed=prezero)
        return costs, acts

    def _bucket_hook(self):
        """all-reduce each finished slice of the flat gradient on a side stream (overlaps with backward)."""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        comm = self._comm_stream
        gflat = self.model.flat_grad()

        def ready(lo, hi, also_wait=None, after=None):
            # ``after``: an event on the main stream behind which the slice's main-stream writers lie (the backward pass
            # releases a layer's weight-gradient GEMMs one recurrence launch late: waiting for the whole main stream would
            # hold the all-reduce back until that recurrence has finished)
            if after is not None:
                comm.wait_event(after)
            else:
                comm.wait_stream(torch.cuda.current_stream())
            if also_wait is not None:              # weight-gradient GEMMs of this slice run on a side stream
                comm.wait_stream(also_wait)
            with torch.cuda.stream(comm):
                dist.all_reduce(gflat[lo:hi])
        return ready

    def _update_autograd(self, inputs, targets, input_percentages, target_sizes):
        """Reference-shaped step through autograd, for optimizers the fused kernel does not cover."""
        out = self.model(inputs)                                        # (B,T,A)
        out_sizes = sanitize_inputs(out.shape[1], input_percentages)
        loss = self.criterion(out.transpose(0, 1), targets, out_sizes, target_sizes) / inputs.shape[0]
        loss = loss.sum()
        is_inf = float(loss.item()) in (float('inf'), float('-inf'))
        if is_inf:                                   # codes/engine.py:27-30: 0 * loss -> a zero gradient, the step still runs
            LOG.warning('WARNING: received an inf loss, setting loss value to 0')
            loss = 0 * loss
        self.optimizer.zero_grad()
        loss.backward()
        if self.world > 1:
            for p in self.model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
                    p.grad.div_(self.world)
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_norm)
        self.optimizer.step()
        torch.cuda.synchronize()
        ops.check_async_errors()
        self.iteration += 1
        return 0.0 if is_inf else float(loss.item())

    def run(self, loader, num_epochs=1, on_iteration=None, on_epoch=None):
        """Epoch loop; the per-step readback is deferred by one step (``update(defer=True)``), so ``on_iteration`` for
        step i runs once step i + 1 has been enqueued."""
        for epoch in range(num_epochs):
            prev = None
            for i, batch in enumerate(loader):
                cur = self.update(batch, defer=True)
                if prev is not None and on_iteration is not None:
                    on_iteration(self, epoch, prev[0], _resolved(prev[1]))
                prev = (i, cur)
            if prev is not None:
                self.flush()
                if on_iteration is not None:
                    on_iteration(self, epoch, prev[0], _resolved(prev[1]))
            if on_epoch is not None:
                on_epoch(self, epoch)


def _resolved(loss):
    return loss.result() if isinstance(loss, PendingLoss) else loss


class Evaluator(object):
    """Inference step + metrics (reference ``codes/engine.py:103-130``, ``codes/metrics.py``)."""

    def __init__(self, model, decoder=None, device='cuda'):
        self.model, self.decoder, self.device = model, decoder, torch.device(device)

    def inference(self, batch):
        self.model.eval()
        with torch.no_grad():
            inputs, targets, input_percentages, target_sizes = batch
            out = self.model(wait_ready(inputs).to(self.device))         # (B,T,A) probabilities
            out_sizes = sanitize_inputs(out.shape[1], input_percentages)
            return out, targets.to('cpu'), out_sizes, target_sizes.to('cpu')

    def run(self, loader):
        """Returns dict(ctcloss, wer, cer) with the training-time metric definitions:
        ctcloss = sum of CTC costs on the eval-mode outputs (probabilities, soft-maxed again inside the loss,
        ``codes/metrics.py:49-51``) / utterances; wer / cer = mean over utterances of distance / reference
        length, x100 (``codes/metrics.py:114-132,143-162``)."""
        tot_loss, n_utt, wer_sum, cer_sum = 0.0, 0, 0.0, 0.0
        for batch in loader:
            out, targets, out_sizes, target_sizes = self.inference(batch)
            costs, _ = ctc_costs_and_grad(out.transpose(0, 1).contiguous(), targets, out_sizes, target_sizes)
            tot_loss += float(costs.sum().item())
            n_utt += out.shape[0]
            if self.decoder is not None:
                hyps, _ = self.decoder.decode(out, out_sizes)
                off = 0
                for i in range(out.shape[0]):
                    n = int(target_sizes[i])
                    ref = self.decoder.convert_to_strings([targets[off:off + n]])[0][0]
                    off += n
                    hyp = hyps[i][0]
                    nw, nc = len(ref.split()), len(ref)
                    w, c = self.decoder.wer(hyp, ref), self.decoder.cer(hyp, ref)
                    wer_sum += w / nw if nw else w
                    cer_sum += c / nc if nc else c
        ops.check_async_errors()                     # (every .item() above synchronised; the timeout flags are sticky)
        n = max(n_utt, 1)
        return {'ctcloss': tot_loss / n, 'wer': 100.0 * wer_sum / n, 'cer': 100.0 * cer_sum / n}


def create_trainer(model, optimizer, criterion, device, **kwargs):
    """Same call as the reference (``train.py:199-200``): kwargs carry max_norm, skip_n, task_weights..."""
    if len(kwargs.get('task_weights', [1])) > 1:
        raise NotImplementedError('multi-task training is out of scope (SURVEY.md section 2)')
    crit = criterion[0] if isinstance(criterion, (list, tuple)) else criterion
    return Trainer(model, optimizer, crit, device, max_norm=kwargs.get('max_norm', 400),
                   skip_n=kwargs.get('skip_n', 0), frontend=kwargs.get('frontend', None))


def create_evaluator(model, metrics=None, device='cuda', decoder=None):
    return Evaluator(model, decoder, device)
