"""CTC criterion with the warp-ctc call signature the reference uses.

Reference call sites: ``train.py:12,179`` (``from warpctc_pytorch import CTCLoss``), ``codes/engine.py:22``
``criterion(out, targets, out_sizes, target_sizes)`` and ``codes/metrics.py:43,51``.  Contract kept:
  acts (T,B,A) un-normalised device activations (softmax applied inside), labels 1-D int32 (CPU or device),
  act_lens / label_lens int32 (B) (CPU or device)  ->  1-element tensor = SUM over the batch of -log p,
  differentiable w.r.t. acts only, blank = 0.
The arithmetic is ``ds2_ctc_loss_grad`` (csrc/ctc.hip); the gradient is produced in the same pass and
handed to autograd in backward.  An infeasible utterance contributes +inf, which the trainer zeroes like
``codes/engine.py:27-30``.
"""
import torch

from ds2hip import ops


def _dev_i32(x, dev):
    return torch.as_tensor(x).to(device=dev, dtype=torch.int32).contiguous()


def ctc_costs_and_grad(acts, labels, act_lens, label_lens, grad_scale=1.0):
    """Raw kernel call: returns (costs (B,), grad (T,B,A)) on the device of ``acts``."""
    dev = acts.device
    label_lens_c = torch.as_tensor(label_lens).to('cpu', torch.int64)
    max_len = int(label_lens_c.max().item()) if label_lens_c.numel() else 0
    offsets = torch.zeros_like(label_lens_c)
    if label_lens_c.numel() > 1:
        offsets[1:] = torch.cumsum(label_lens_c, 0)[:-1]
    labels_d = _dev_i32(labels, dev).reshape(-1)
    if labels_d.numel() == 0:
        labels_d = torch.zeros(1, dtype=torch.int32, device=dev)
    return ops.ctc_loss_grad(acts.contiguous().float(), labels_d, _dev_i32(offsets, dev),
                             _dev_i32(label_lens_c, dev), _dev_i32(act_lens, dev), max_len, grad_scale)


class _CTCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens):
        costs, grad = ctc_costs_and_grad(acts, labels, act_lens, label_lens)
        ctx.save_for_backward(grad)
        return costs.sum().reshape(1)

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return grad * grad_output.reshape(1, 1, 1), None, None, None


class CTCLoss(torch.nn.Module):
    """Drop-in for ``warpctc_pytorch.CTCLoss()`` as used by the reference."""

    def forward(self, acts, labels, act_lens, label_lens):
        return _CTCFunction.apply(acts, labels, act_lens, label_lens)
