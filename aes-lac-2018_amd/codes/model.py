"""DeepSpeech2 (conv -> BiGRU -> FC) for MI355X: the reference's ``codes.model.DeepSpeech`` surface over
hand-written HIP kernels.

Drop-in contract (reference ``codes/model.py:116-207``):
  * same constructor keywords (``:119-127``), same ``forward(x: (B,T_in,161)) -> (B,T,num_classes)``:
    un-normalised activations in training mode, softmax in eval mode (``:201-205``);
  * same ``state_dict()`` key names and shapes, so checkpoints interchange by name
    (``codes/utils/model_utils.py:70``): ``conv.{0,3}.{weight,bias}``, ``conv.{1,4}.*``,
    ``rnns.N.rnn.weight_{ih,hh}_l0[_reverse]``, ``rnns.N.batch_norm.module.*``, ``fc.0.module.{0,1}.*``;
  * same arithmetic quirks: no sequence packing (``:62``), BatchNorm over padded frames (``:59-60``),
    directions summed (``:64-67``), time padding on conv1 only (``:143-144``).

What differs is underneath: every tensor op is a kernel of libds2hip.so (``include/ds2hip.h``).  The
modules below only HOLD parameters under the reference's names; none of them has a PyTorch forward.
Parameters live in one flat fp32 buffer (views), laid out so that both directions' W_ih (and W_hh)
of a layer are adjacent: one MFMA GEMM produces the gate pre-activations of both directions and one
fused kernel applies clip + Nesterov SGD to everything.

Out of scope: unidirectional + Lookahead, LSTM/RNN cells, multi-task heads.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from ds2hip import ops

from .utils.dist_utils import top_layer_spare_cus


# ------------------------------------------------------------------------------------ parameter holders
class _Conv2dParams(nn.Module):
    """weight (Cout,Cin,KF,KT) + bias, initialised like torch.nn.Conv2d.reset_parameters."""

    def __init__(self, cin, cout, kf, kt):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, kf, kt))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * kf * kt)
        nn.init.uniform_(self.bias, -bound, bound)


class _BatchNormParams(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def _apply(self, fn, *args, **kwargs):
        """``num_batches_tracked`` is bookkeeping the kernels never read: it stays on the HOST whatever device the
        module moves to, so bumping it after a training forward is a host add, not a kernel launch (7 per step)."""
        out = super()._apply(fn, *args, **kwargs)
        self._buffers['num_batches_tracked'] = self._buffers['num_batches_tracked'].to('cpu')
        return out


class _Placeholder(nn.Module):
    """Keeps the reference's Sequential indices (Hardtanh slots hold no parameters)."""


class _GRUParams(nn.Module):
    """weight_{ih,hh}_l0[_reverse], gate order r,z,n, initialised like torch.nn.GRU (uniform +-1/sqrt(H))."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        for suffix in ('', '_reverse'):
            for nm, cols in (('weight_ih_l0', input_size), ('weight_hh_l0', hidden_size)):
                p = nn.Parameter(torch.empty(3 * hidden_size, cols))
                nn.init.uniform_(p, -k, k)
                self.register_parameter(nm + suffix, p)


class _LinearParams(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class SequenceWise(nn.Module):
    """Holder named like the reference's SequenceWise (codes/model.py:13-34): ``.module``."""

    def __init__(self, module):
        super().__init__()
        self.module = module


class BatchRNN(nn.Module):
    """Holder named like the reference's BatchRNN (codes/model.py:43-69): ``.batch_norm.module``, ``.rnn``."""

    def __init__(self, input_size, hidden_size, batch_norm=True):
        super().__init__()
        self.batch_norm = SequenceWise(_BatchNormParams(input_size)) if batch_norm else None
        self.rnn = _GRUParams(input_size, hidden_size)


# ------------------------------------------------------------------------------------ autograd bridge
class _DS2Function(torch.autograd.Function):
    """Whole-network forward/backward as ONE autograd node, so ``loss.backward()`` works as in the reference."""

    @staticmethod
    def forward(ctx, model, x, *params):
        acts, saved = model._forward_impl(x, training=True, need_grad=True)
        ctx.model, ctx.saved = model, saved
        return acts

    @staticmethod
    def backward(ctx, d_acts):
        model = ctx.model
        gflat = torch.empty_like(model._flat_p)
        model._backward_impl(ctx.saved, d_acts.contiguous(), gflat)
        ctx.saved = None
        grads = [gflat[o:o + p.numel()].view_as(p) for p, o in zip(model._plist, model._offsets)]
        return (None, None) + tuple(grads)


# Compute units a backward recurrence launch leaves to the side stream's weight-gradient GEMMs (ds2_gru_bidir_bwd_persistent_ex):
# same-box A/B of the whole B = 10 step, ms with a synchronisation per step, 48 steps: 52 CUs free (24 units per workgroup)
# 15.66, 82 free (28 units) 15.57 -- the slower recurrence form wins because dW_ih of the layer above then hides beside it too.
_BWD_SPARE_CUS = int(os.environ.get('DS2_GRU_BWD_SPARE', '82'))


class DeepSpeech(nn.Module):
    __version__ = '0.0.1'

    def __init__(self, rnn_type=nn.GRU, num_classes=29, rnn_hidden_size=800, num_rnn_layers=5, window_size=320,
                 bidirectional=True, context=20, include_classifier=True):
        super().__init__()
        if isinstance(rnn_type, str):
            rnn_type = getattr(torch.nn, rnn_type.upper())
        if rnn_type is not nn.GRU or not bidirectional or not include_classifier:
            raise NotImplementedError('the MI355X path implements the bidirectional-GRU classifier model only '
                                      '(every BASELINE config); see SURVEY.md section 2 row 1')
        if rnn_hidden_size % 8 != 0:
            raise ValueError('rnn_hidden_size must be a multiple of 8')
        self._rnn_type = rnn_type
        self._num_classes = num_classes
        self._rnn_hidden_size = rnn_hidden_size
        self._num_rnn_layers = num_rnn_layers
        self._window_size = window_size
        self._bidirectional = bidirectional
        self._context = context
        self._include_classifier = include_classifier

        self.conv = nn.Sequential(_Conv2dParams(1, 32, 41, 11), _BatchNormParams(32), _Placeholder(),
                                  _Conv2dParams(32, 32, 21, 11), _BatchNormParams(32), _Placeholder())
        feat = window_size // 2 + 1
        feat = (feat - 41) // 2 + 1
        feat = (feat - 21) // 2 + 1
        if window_size != 320:
            raise NotImplementedError('window_size=%r: the HIP conv / BatchNorm-layout / STFT kernels are specialised for 161 '
                                      'frequency bins (window_size=320, every shipped config); the reference accepts any '
                                      'window (codes/model.py:124,148-151) -- README.md "What the drop-in does not cover"'
                                      % (window_size,))
        self._rnn_input_size = 32 * feat
        rnns = [('0', BatchRNN(self._rnn_input_size, rnn_hidden_size, batch_norm=False))]
        for i in range(num_rnn_layers - 1):
            rnns.append((str(i + 1), BatchRNN(rnn_hidden_size, rnn_hidden_size)))
        self.rnns = nn.Sequential(OrderedDict(rnns))
        self.lookahead = None
        self.fc = nn.Sequential(SequenceWise(nn.Sequential(_BatchNormParams(rnn_hidden_size),
                                                           _LinearParams(rnn_hidden_size, num_classes))))
        self._flat_p = None
        self._flat_g = None
        self._plist, self._offsets = [], []
        self.overlap_wgrad = os.environ.get('DS2_OVERLAP_WGRAD', '1') != '0'
        self.defer_wgrad = os.environ.get('DS2_DEFER_WGRAD', '1') != '0'
        self._side = None

    # ------------------------------------------------------------------ flat parameter storage
    def _flat_order(self):
        c = self.conv
        order = [c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[3].weight, c[3].bias, c[4].weight, c[4].bias]
        for layer in self.rnns:
            if layer.batch_norm is not None:
                order += [layer.batch_norm.module.weight, layer.batch_norm.module.bias]
            r = layer.rnn
            order += [r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse]
        head = self.fc[0].module
        order += [head[0].weight, head[0].bias, head[1].weight]
        return order

    def flatten_parameters(self):
        """(Re)pack every parameter into one contiguous fp32 buffer; parameters become views of it."""
        order = self._flat_order()
        dev = order[0].device
        offsets, n = [], 0
        for p in order:
            offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(order, offsets):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1).to(device=dev, dtype=torch.float32))
                p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat_p, self._plist, self._offsets = flat, order, offsets
        self._flat_g = None
        self.__dict__['_flat_sig'] = None
        return self

    def _flat_signature(self):
        """What the fast path of ``_ensure_flat`` checks: every (parent, child name, child) edge of the module tree and, for
        every parameter of the flat order, (owner module, name, parameter object, offset)."""
        edges, owners = [], {}
        for parent in self.modules():
            for nm, child in parent._modules.items():
                edges.append((parent, nm, child))
            for nm, prm in parent._parameters.items():
                if prm is not None:
                    owners[id(prm)] = (parent, nm)
        params = [(owners[id(p)][0], owners[id(p)][1], p, o) for p, o in zip(self._plist, self._offsets)]
        return edges, params

    def _ensure_flat(self):
        """The parameters are views of ONE flat buffer, in the flat order.  Called at the top of every step, with the GPU idle
        under the reference's synchronise-per-step protocol, so the common case is cheap (~70 dict lookups and pointer compares:
        the module tree has its recorded edges, every parameter object is where it was and points where it should); anything
        else -- ``.to()``, a swapped sub-module (fine-tuning), a re-assigned parameter -- takes the full walk below."""
        sig = self.__dict__.get('_flat_sig')
        if sig is not None and self._flat_p is not None:
            base = self._flat_p.data_ptr()
            ok = True
            for parent, nm, child in sig[0]:
                if parent._modules.get(nm) is not child:
                    ok = False
                    break
            if ok:
                for owner, nm, p, o in sig[1]:
                    if owner._parameters.get(nm) is not p or p.data_ptr() != base + 4 * o:
                        ok = False
                        break
            if ok:
                return
        self.__dict__.pop('_bn_walk', None)                  # a swapped sub-module: the cached BatchNorm list is stale
        order = self._flat_order()
        ok = self._flat_p is not None and len(order) == len(self._plist)
        if ok:
            base = self._flat_p.data_ptr()
            for p, q, o in zip(order, self._plist, self._offsets):
                if p is not q or p.data_ptr() != base + 4 * o:
                    ok = False
                    break
        if not ok:
            self.flatten_parameters()
        self.__dict__['_flat_sig'] = self._flat_signature()

    def flat_grad(self):
        """Persistent flat gradient buffer; each parameter's .grad is a view of it."""
        self._ensure_flat()
        if self._flat_g is None or self._flat_g.device != self._flat_p.device:
            self._flat_g = torch.zeros_like(self._flat_p)
            for p, o in zip(self._plist, self._offsets):
                p.grad = self._flat_g[o:o + p.numel()].view(p.shape)
        return self._flat_g

    def _gview(self, gflat, p):
        for q, o in zip(self._plist, self._offsets):
            if q is p:
                return gflat[o:o + p.numel()].view(p.shape)
        raise KeyError('parameter not in flat buffer')

    # ------------------------------------------------------------------ chain timing (bench.py's live time-share table)
    def _tick(self, name):
        """Phase boundary on the main stream.  A no-op unless ``self._ticks`` is a list (bench.py sets it for one step):
        then a timing event is recorded, and the elapsed time between consecutive events is the CHAIN time of the phase
        that ends here (waits for the side stream included)."""
        ticks = self.__dict__.get('_ticks')
        if ticks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            ticks.append((name, ev))

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        dev = self.conv[0].weight.device
        if dev.type != 'cuda':
            raise RuntimeError('DeepSpeech runs on a ROCm device only (model.to("cuda")); there is no CPU path')
        if x.device != dev:            # the reference hands CPU batches to DataParallel (codes/engine.py:58)
            x = x.to(dev)
        x = x.contiguous().float()
        self._ensure_flat()
        if self.training and torch.is_grad_enabled():
            acts = _DS2Function.apply(self, x, *self._plist)
        else:
            acts, _ = self._forward_impl(x, training=self.training, need_grad=False)
        out = acts.transpose(0, 1)                                   # (B,T,A) view, as codes/model.py:201
        if not self.training:
            t, b, a = acts.shape
            out = ops.softmax_rows(acts.view(t * b, a), t * b, a).view(t, b, a).transpose(0, 1)
        return out

    def _forward_impl(self, x, training, need_grad, after_conv=None):
        """x (B,T_in,161) -> acts (T,B,A); returns the tensors backward needs.  ``after_conv``: called once the conv block's
        launches are enqueued (the trainer's gradient fill: host work that need not sit in front of the step's first kernels)."""
        bsz, t_in, nfreq = x.shape
        assert nfreq == 161, 'expected 161 frequency bins'
        hid, nlayers = self._rnn_hidden_size, self._num_rnn_layers
        t1, t = ops.conv_out_frames(t_in)
        assert t > 0, 'input too short for the conv stack'
        sv = {'t_in': t_in, 't1': t1, 't': t, 'bsz': bsz}
        c = self.conv
        # Each BatchNorm follows its OWN ``training`` flag, as torch's does.  The reference's ``_freeze_layers`` puts the
        # BatchNorm modules of a frozen layer in eval mode once, at set-up (training_utils.py:52-54,73) -- and its update
        # step calls ``model.train()`` at the top of EVERY step (codes/engine.py:51), which puts them back: under the
        # reference's trainer a frozen conv block still normalises with batch statistics and keeps moving its running
        # estimates (Trainer.update repeats that call).  A caller that drives forward_backward itself with
        # ``model.conv[1].eval()`` gets inference-mode BatchNorm in the conv block; in the recurrent stack and the head only
        # the all-training case exists.
        # (plain instance-dict lookups: a missing attribute on an nn.Module costs a failed walk through its parameter /
        # buffer / sub-module tables and an exception -- ~3 us each, ~30 of them here, at the top of a step with the GPU idle)
        # (a BatchNorm is "held" when it is in eval mode inside a model that is in training mode; ``training`` itself -- the
        # caller's choice of batch or running statistics for the pass -- is independent of the flags, as before)
        model_training = self.training
        c1_train = training and not (model_training and not c[1].training)
        c4_train = training and not (model_training and not c[4].training)
        if training and model_training:
            mods = self.__dict__.get('_bn_walk')
            if mods is None:                               # the module tree is static (_ensure_flat drops this on a swap)
                mods = self.__dict__['_bn_walk'] = [m for m in list(self.rnns.modules()) + list(self.fc.modules())
                                                    if isinstance(m, _BatchNormParams)]
            for mod in mods:
                if not mod.training:
                    raise NotImplementedError('BatchNorm in inference mode inside a training step is supported for the '
                                              'conv block only')
        self._tick('start')
        xt = ops.transpose_btf(x)                                                   # (B,161,T_in)
        y1 = ops.conv_fwd(1, xt, c[0].weight, c[0].bias, t_in)                      # (B,32,61,T1)
        mi1 = ops.bn2d_stats(y1, c[1].running_mean, c[1].running_var, c1_train)
        a1 = ops.bn2d_apply_htanh(y1, mi1, c[1].weight, c[1].bias, layout_tbf=False)
        y2 = ops.conv_fwd(2, a1, c[3].weight, c[3].bias, t1)                        # (B,32,21,T)
        mi2 = ops.bn2d_stats(y2, c[4].running_mean, c[4].running_var, c4_train)
        xin = ops.bn2d_apply_htanh(y2, mi2, c[4].weight, c[4].bias, layout_tbf=True)  # (T,B,672)
        if c1_train:
            c[1].num_batches_tracked += 1
        if c4_train:
            c[4].num_batches_tracked += 1
        # a conv block with no trainable parameter (freeze_layers: ["conv"]): nothing upstream of the recurrent stack needs
        # a gradient, whatever mode its BatchNorm runs in
        sv['conv_frozen'] = need_grad and not (c[0].weight.requires_grad or c[0].bias.requires_grad or
                                               c[1].weight.requires_grad or c[1].bias.requires_grad or
                                               c[3].weight.requires_grad or c[3].bias.requires_grad or
                                               c[4].weight.requires_grad or c[4].bias.requires_grad)
        if need_grad and not (c1_train and c4_train) and not sv['conv_frozen']:
            raise NotImplementedError('inference-mode BatchNorm in a conv block with trainable parameters has no backward here')
        sv.update(xt=xt, y1=y1, mi1=mi1, a1=a1, y2=y2, mi2=mi2)
        self._tick('conv block forward (transpose, conv1, conv2, 2 x BatchNorm + clip)')
        if after_conv is not None:
            after_conv()
        rows = t * bsz
        layers = []
        prev_h = None
        if need_grad:
            # The backward recurrence reads W_hh transposed.  The weights do not change during the step, so the ten
            # transposes run now, on the side stream, on CUs the forward recurrence leaves idle -- not between
            # BatchNorm backward and the recurrence kernel of every layer, where they sit on the critical chain.
            wt = torch.empty((nlayers, 2, hid, 3 * hid), dtype=torch.float32, device=x.device)
            main = torch.cuda.current_stream()
            side = self._side_stream(x.device) if self.overlap_wgrad else main
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                # (one launch per 8 layers instead of two per layer: ten ~8-us launches sat in the host's path right at the
                # start of the step, where the GPU waits for the host)
                pairs = [self._pair(layer.rnn.weight_hh_l0, layer.rnn.weight_hh_l0_reverse).view(2, 3 * hid, hid)
                         for layer in self.rnns]
                for lo in range(0, nlayers, 8):
                    ops.transpose2d_group(pairs[lo:lo + 8], 2, 3 * hid, hid, wt[lo:lo + 8])
                ready = torch.cuda.Event()
                ready.record()
            sv['w_hh_t'], sv['w_hh_t_ready'] = wt, ready
        for li, layer in enumerate(self.rnns):
            n_in = self._rnn_input_size if li == 0 else hid
            rec = {}
            if layer.batch_norm is not None:
                bn = layer.batch_norm.module
                mi = ops.bn1d_stats(prev_h[0], prev_h[1], rows, hid, bn.running_mean, bn.running_var, training)
                xin = ops.bn1d_apply(prev_h[0], prev_h[1], mi, bn.weight, bn.bias, rows, hid)
                if training:
                    bn.num_batches_tracked += 1
                rec['mi'] = mi
            r = layer.rnn
            w_ih = self._pair(r.weight_ih_l0, r.weight_ih_l0_reverse)               # (6H, In) view
            w_hh = self._pair(r.weight_hh_l0, r.weight_hh_l0_reverse)               # (2*3H, H) view
            gates = ops.gemm(xin.view(rows, n_in), w_ih, trans_b=True)              # (T*B, 6H)
            self._tick('BatchNorm + input projection GEMM (forward)')
            # (a training pass also takes the backward recurrence's coefficient planes where its d(h)-hand-off form will run)
            ghn, hout, coef = ops.gru_bidir_fwd(gates, w_hh, t, bsz, hid, want_coef=True) if need_grad else \
                (ops.gru_bidir_fwd(gates, w_hh, t, bsz, hid) + (None,))
            self._tick('BiGRU recurrence forward')
            rec.update(xin=xin, gates=gates, ghn=ghn, hout=hout, coef=coef)
            layers.append(rec)
            prev_h = hout
        head = self.fc[0].module
        mi = ops.bn1d_stats(prev_h[0], prev_h[1], rows, hid, head[0].running_mean, head[0].running_var, training)
        xf = ops.bn1d_apply(prev_h[0], prev_h[1], mi, head[0].weight, head[0].bias, rows, hid)
        if training:
            head[0].num_batches_tracked += 1
        acts = ops.gemm(xf, head[1].weight, trans_b=True).view(t, bsz, self._num_classes)
        self._tick('head: BatchNorm + FC (forward)')
        sv.update(layers=layers, mi_fc=mi, xf=xf)
        return acts, (sv if need_grad else None)

    def _pair(self, p_fwd, p_rev):
        """The two directions' weights are adjacent in the flat buffer: return them as one (2*rows, cols) matrix."""
        n = p_fwd.numel()
        assert p_rev.data_ptr() == p_fwd.data_ptr() + 4 * n, 'flat layout broken; call flatten_parameters()'
        off = (p_fwd.data_ptr() - self._flat_p.data_ptr()) // 4
        return self._flat_p[off:off + 2 * n].view(2 * p_fwd.shape[0], p_fwd.shape[1])

    # ------------------------------------------------------------------ backward
    def _span(self, first, last):
        """[lo, hi) of the flat buffer covering parameters first..last (inclusive, flat order)."""
        lo = hi = None
        for q, o in zip(self._plist, self._offsets):
            if q is first:
                lo = o
            if q is last:
                hi = o + (q.numel() + 3) // 4 * 4
        return lo, hi

    def _backward_impl(self, sv, d_acts, gflat, grad_ready=None, prezeroed=False):
        """d_acts (T,B,A) -> gradients of every parameter written into ``gflat`` (same layout as the flat params).

        ``grad_ready(lo, hi)`` is called as soon as the slice [lo, hi) of ``gflat`` is final (head, then each
        GRU layer top-down, then the conv stack) so a data-parallel trainer can start its all-reduce early.
        """
        hid, ncls = self._rnn_hidden_size, self._num_classes
        t, bsz, t1, t_in = sv['t'], sv['bsz'], sv['t1'], sv['t_in']
        rows = t * bsz
        gv = lambda p: self._gview(gflat, p)                                        # noqa: E731
        # everything the forward pass put on the side stream (the trainer's one fill of ``gflat``, the W_hh transposes) is
        # complete before the first gradient is written
        main = torch.cuda.current_stream()
        self._tick('CTC loss + gradient')
        main.wait_event(sv['w_hh_t_ready'])
        head = self.fc[0].module
        d2 = d_acts.reshape(rows, ncls)
        ops.gemm(d2, sv['xf'], trans_a=True, out=gv(head[1].weight), split_k=0)               # dW_fc = d^T xf
        dxf = ops.gemm(d2, head[1].weight, split_k=0)                                          # (rows,H)
        last = sv['layers'][-1]['hout']
        dy = ops.bn1d_bwd(last[0], last[1], dxf, sv['mi_fc'], head[0].weight, rows, hid, gv(head[0].weight),
                          gv(head[0].bias))
        if grad_ready is not None:
            grad_ready(*self._span(head[0].weight, head[1].weight))
        self._tick('head: FC + BatchNorm (backward)')
        nl = len(sv['layers'])
        f4 = 4
        # The weight-gradient GEMMs of a layer (dW_ih, dW_hh) are not on the chain that feeds the next (lower) layer:
        # chain = d(gi) -> dX GEMM -> BatchNorm backward -> recurrence of the layer below.  They go to a LOW-priority
        # side stream and fill the ~56 CUs the persistent recurrence kernel (200 workgroups) leaves idle.
        side = self._side_stream(gflat.device) if self.overlap_wgrad else None
        keepalive = []
        # The side stream's GEMMs of layer l are released only once the recurrence kernel of layer l - 1 has been LAUNCHED
        # (an event recorded on the main stream right in front of it): a low-priority GEMM that is already running when the
        # recurrence arrives keeps back-filling every CU that is only partly free, and the recurrence's 200 whole-CU
        # workgroups are not all resident -- the ones that are spin -- until that GEMM has run out of workgroups
        # (measured, rocprofv3 timeline at T = 495: 1.50 ms for the top layer's backward kernel, 1.88-1.94 ms for the four
        # that started behind a 0.38 ms dW_hh GEMM).
        acc_beta = 1.0 if prezeroed else 0.0
        pending = None
        # The dX GEMMs split K and add their partial products with atomics, so their outputs must start from zero.  ONE buffer
        # for the five layers, cleared by ONE fill on its own normal-priority stream here, at the start of the backward pass
        # (done long before the first dX GEMM, a recurrence launch away), with ONE pair of events -- a fill and two events per
        # layer put two marker packets in front of every recurrence launch and a cross-stream wait in front of every dX GEMM
        # (~30 us of idle chain per layer in the kernel trace; the chain must not wait for the low-priority weight-gradient
        # stream's backlog either: hence the stream of its own).
        dx_all, dx_ready, dx_off = None, None, 0
        if side is not None:
            fill = self._fill_stream(gflat.device)
            total = rows * (hid * (nl - 1) + self._rnn_input_size)
            dx_all = torch.empty((total,), dtype=torch.float32, device=gflat.device)
            dx_all.record_stream(fill)
            alloc_ev = torch.cuda.Event()
            alloc_ev.record(main)
            fill.wait_event(alloc_ev)
            with torch.cuda.stream(fill):
                dx_all.zero_()
                dx_ready = torch.cuda.Event()
                dx_ready.record(fill)
        for li in range(nl - 1, -1, -1):
            rec = sv['layers'][li]
            layer = self.rnns[li]
            r = layer.rnn
            n_in = self._rnn_input_size if li == 0 else hid
            w_ih = self._pair(r.weight_ih_l0, r.weight_ih_l0_reverse)
            w_hh_t = sv['w_hh_t'][li]                                               # transposed during forward
            gates, ghn, hout = rec['gates'], rec['ghn'], rec['hout']
            dx_buf = None
            if dx_all is not None:
                dx_buf = dx_all[dx_off:dx_off + rows * n_in].view(rows, n_in)
                dx_off += rows * n_in
            if pending is not None:
                gate = torch.cuda.Event()
                gate.record(main)
            # CUs to leave free beside the launch: none for the top layer (nothing is queued beside it yet -- the widest, fastest
            # grid), room for the side stream's weight-gradient GEMMs of the layer above under the others
            # (data-parallel runs: the head's all-reduce is in flight beside the top layer's launch, and RCCL's channel
            # kernels -- up to NCCL_MAX_NCHANNELS = 32 workgroups that may be WAITING for a peer -- must not share the chip with
            # a grid that needs 240 of its 256 CUs resident at once: room for them there too)
            top_spare = 0 if grad_ready is None else top_layer_spare_cus()   # (a collective beside it: one CU per RCCL channel + 8)
            spare = (top_spare if li == nl - 1 else _BWD_SPARE_CUS) if side is not None else top_spare
            ops.gru_bidir_bwd(gates, ghn, hout, dy, w_hh_t, t, bsz, hid, spare_cus=spare, coef=rec.get('coef'))   # gates -> d(gi), ghn -> d(gh_n)
            self._tick('BiGRU recurrence backward (weight-gradient GEMMs beside it)')
            if pending is not None:
                pending(gate)
                pending = None
            dgi = gates.view(rows, 6 * hid)
            if dx_buf is not None:
                if dx_ready is not None:
                    main.wait_event(dx_ready)                                       # once: every later layer's slice is behind it
                    dx_ready = None
                dxin = ops.gemm(dgi, w_ih, out=dx_buf, beta=1.0, split_k=0)         # (rows, n_in): on the chain
            else:
                dxin = ops.gemm(dgi, w_ih, split_k=0)
            g_hh = (gv(r.weight_hh_l0), gv(r.weight_hh_l0_reverse))
            g_ih = self._pair_view(gflat, r.weight_ih_l0, r.weight_ih_l0_reverse)

            def wgrad(gate, gates=gates, ghn=ghn, hout=hout, dgi=dgi, xin=rec['xin'], n_in=n_in, g_hh=g_hh, g_ih=g_ih,
                      layer=layer, r=r):
                if side is not None:
                    side.wait_event(gate) if gate is not None else side.wait_stream(main)
                with torch.cuda.stream(side if side is not None else main):
                    # dW_hh[d] = dGH[d]^T h_prev[d]; forward dir pairs step t with h[t-1], reverse with h[t+1]
                    if t > 1:
                        k = (t - 1) * bsz
                        step_g, step_n = bsz * 6 * hid * f4, bsz * 2 * hid * f4
                        group = []                        # the four dW_hh problems of the layer in ONE launch
                        for d in (0, 1):
                            a_g = gates.data_ptr() + d * 3 * hid * f4 + (step_g if d == 0 else 0)
                            a_n = ghn.data_ptr() + d * hid * f4 + (step_n if d == 0 else 0)
                            hp = hout[d].data_ptr() + (0 if d == 0 else bsz * hid * f4)
                            group.append((a_g, 6 * hid, 2 * hid, hp, hid, g_hh[d].data_ptr(), hid))
                            group.append((a_n, 2 * hid, hid, hp, hid, g_hh[d].data_ptr() + 2 * hid * hid * f4, hid))
                        # prezeroed: the caller cleared the whole flat gradient with ONE fill (beside the forward pass), so
                        # the split-K GEMMs add into it instead of each clearing its own output first (45 fills per step)
                        # one grouped launch while K = rows is moderate (same-box A/B of the whole step: B = 10 +0.6 %,
                        # B = 32 (K up to 24 k) +0.6 %, B = 64 x 15 s (K = 47 k) -1.6 %: there one launch per problem, each
                        # with its own finer split of K, is faster)
                        if k <= 32768:
                            ops.gemm_tn_group(group, hid, k, accumulate=prezeroed)
                        else:
                            for a_p, lda, m_p, b_p, ldb, c_p, ldc in group:
                                ops.gemm_raw(1, 0, m_p, hid, k, a_p, lda, b_p, ldb, c_p, ldc, beta=acc_beta, split_k=0)
                    elif not prezeroed:
                        g_hh[0].zero_()
                        g_hh[1].zero_()
                    ops.gemm(dgi, xin.view(rows, n_in), trans_a=True, out=g_ih, beta=acc_beta, split_k=0)  # dW_ih (both dirs)
                if grad_ready is not None:
                    first = layer.batch_norm.module.weight if layer.batch_norm is not None else r.weight_ih_l0
                    if gate is not None:
                        grad_ready(*self._span(first, r.weight_hh_l0_reverse), also_wait=side, after=gate)
                    else:
                        grad_ready(*self._span(first, r.weight_hh_l0_reverse), also_wait=side)

            if layer.batch_norm is not None:
                bn = layer.batch_norm.module
                below = sv['layers'][li - 1]['hout']
                dy = ops.bn1d_bwd(below[0], below[1], dxin, rec['mi'], bn.weight, rows, hid, gv(bn.weight),
                                  gv(bn.bias))
            else:
                dy = dxin
            self._tick('dX GEMM + BatchNorm (backward)')
            if side is not None and self.defer_wgrad:
                pending = wgrad                    # released behind the next layer's recurrence launch
            else:
                wgrad(None)
            if side is not None:
                keepalive.append(dict(rec))        # the side stream still reads these; freed after the join below
            rec.clear()
        if pending is not None:                    # the bottom layer's: beside the conv block's backward
            pending(None)
        # The side stream may still be working through weight-gradient GEMMs (it only gets the CUs the recurrence
        # kernels leave free): the conv block's backward below runs beside that backlog, and the ONE join with the
        # side stream is at the end of this function.  `keepalive` keeps the tensors it reads allocated until then.
        c = self.conv
        if sv.get('conv_frozen', False):             # frozen conv block: nothing upstream needs a gradient
            lo, hi = self._span(c[0].weight, c[4].bias)
            gflat[lo:hi].zero_()
            if side is not None:
                main.wait_stream(side)
            keepalive.clear()
            if grad_ready is not None:
                grad_ready(lo, hi)
            return
        d_a2 = ops.transpose2d(dy, t, bsz * self._rnn_input_size).view(bsz, 32, 21, t)   # (T,B,672) -> (B,32,21,T)
        d_y2 = ops.bn2d_htanh_bwd(sv['y2'], d_a2, sv['mi2'], c[4].weight, c[4].bias, gv(c[4].weight), gv(c[4].bias))
        if side is not None:                         # conv2's filter gradient is off the dgrad -> BN1 -> conv1 chain
            side.wait_stream(main)
        with torch.cuda.stream(side if side is not None else main):
            ops.conv_wgrad(2, sv['a1'], d_y2, t1, gv(c[3].weight), gv(c[3].bias))
        d_a1 = ops.conv2_dgrad(d_y2, c[3].weight, t1)
        d_y1 = ops.bn2d_htanh_bwd(sv['y1'], d_a1, sv['mi1'], c[1].weight, c[1].bias, gv(c[1].weight), gv(c[1].bias))
        ops.conv_wgrad(1, sv['xt'], d_y1, t_in, gv(c[0].weight), gv(c[0].bias))
        self._tick('conv block backward (main stream)')
        if side is not None:
            main.wait_stream(side)                  # every later consumer of the gradients sits behind this join
            self._tick('tail: waiting for the side stream (bottom layer dW, conv2 wgrad)')
        keepalive.clear()
        if grad_ready is not None:
            grad_ready(*self._span(c[0].weight, c[4].bias))

    def _side_stream(self, dev):
        """Low-priority stream for the weight-gradient GEMMs (default-priority work always gets free CUs first)."""
        st = getattr(self, '_side', None)
        if st is None or st.device != dev:
            if os.environ.get('DS2_SIDE_PRIORITY', 'low') == 'low':
                st = ops.low_priority_stream(dev)
            else:
                st = torch.cuda.Stream(device=dev)
            self._side = st
        return st

    def _fill_stream(self, dev):
        """Normal-priority stream for the dX buffers' zero fills (they run beside the recurrence launch; the chain waits for
        them, so they must not queue behind the low-priority weight-gradient stream's backlog)."""
        st = self.__dict__.get('_fill')
        if st is None or st.device != dev:
            st = self.__dict__['_fill'] = torch.cuda.Stream(device=dev)
        return st

    def _pair_view(self, gflat, p_fwd, p_rev):
        n = p_fwd.numel()
        off = (p_fwd.data_ptr() - self._flat_p.data_ptr()) // 4
        return gflat[off:off + 2 * n].view(2 * p_fwd.shape[0], p_fwd.shape[1])

    # ------------------------------------------------------------------ training fast path (no autograd)
    def forward_backward(self, x, loss_fn):
        """One fused pass for the trainer: acts -> ``loss_fn(acts) -> (loss, d_acts)`` -> gradients into flat_grad().

        Mirrors model(inputs) + criterion + backward of codes/engine.py:62-84 without building an autograd graph.
        """
        dev = self.conv[0].weight.device
        if x.device != dev:
            x = x.to(dev)
        self._ensure_flat()
        gflat = self.flat_grad()
        acts, sv = self._forward_impl(x.contiguous().float(), training=True, need_grad=True)
        loss, d_acts = loss_fn(acts)
        self._backward_impl(sv, d_acts, gflat)
        return loss, acts
