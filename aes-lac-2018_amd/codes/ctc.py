"""CTC criterion with the warp-ctc call signature the reference uses.

Reference call sites: ``train.py:12,179`` (``from warpctc_pytorch import CTCLoss``), ``codes/engine.py:22``
``criterion(out, targets, out_sizes, target_sizes)`` and ``codes/metrics.py:43,51``.  Contract kept:
  acts (T,B,A) un-normalised device activations (softmax applied inside), labels 1-D int32 (CPU or device),
  act_lens / label_lens int32 (B) (CPU or device)  ->  1-element tensor = SUM over the batch of -log p,
  differentiable w.r.t. acts only, blank = 0.
The arithmetic is ``ds2_ctc_loss_grad`` (csrc/ctc.hip); the gradient is produced in the same pass and
handed to autograd in backward.  An infeasible utterance contributes +inf (and, like warp-ctc, no gradient of its
own); the trainer then applies ``codes/engine.py:24-30``: the batch loss becomes ``0 * loss``, i.e. the value is
reported as 0 and the WHOLE batch's gradient is zero (``zero_batch_if_inf`` does that inside the gradient kernel
for the fused step; the autograd path gets it from the 0 factor).
"""
import torch

from ds2hip import ops


def ctc_costs_and_grad(acts, labels, act_lens, label_lens, grad_scale=1.0, zero_batch_if_inf=False):
    """Raw kernel call: returns (costs (B,), grad (T,B,A)) on the device of ``acts``.

    The four small integer arrays the warp-ctc signature passes on the host (labels, lengths) travel to the
    device in ONE upload through a reusable pinned staging buffer."""
    dev = acts.device
    label_lens_c = torch.as_tensor(label_lens).to('cpu', torch.int32).reshape(-1)
    act_lens_c = torch.as_tensor(act_lens).to('cpu', torch.int32).reshape(-1)
    labels_c = torch.as_tensor(labels).to('cpu', torch.int32).reshape(-1)
    bsz = label_lens_c.numel()
    max_len = int(label_lens_c.max().item()) if bsz else 0
    nlab = max(int(labels_c.numel()), 1)
    packed = torch.zeros(nlab + 3 * bsz, dtype=torch.int32)
    packed[:labels_c.numel()] = labels_c
    if bsz > 1:
        packed[nlab + 1:nlab + bsz] = torch.cumsum(label_lens_c, 0)[:-1]          # start of each utterance's labels
    packed[nlab + bsz:nlab + 2 * bsz] = label_lens_c
    packed[nlab + 2 * bsz:] = act_lens_c
    d = ops.upload_small(packed, dev)
    return ops.ctc_loss_grad(acts.contiguous().float(), d[:nlab], d[nlab:nlab + bsz], d[nlab + bsz:nlab + 2 * bsz],
                             d[nlab + 2 * bsz:], max_len, grad_scale, zero_batch_if_inf)


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
