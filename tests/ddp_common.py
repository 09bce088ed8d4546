"""Synthetic utterances and the DDP-emulating oracle for the 2-rank data-parallel step test.

What the reference does with W processes (``train.py:172-175``, ``codes/sampler.py:113-125``, ``codes/engine.py:79-90``
under ``DistributedDataParallel``): rank r takes bins r, r+W, ...; every replica normalises with ITS OWN batch
statistics; each rank's loss is divided by its local batch size; DDP averages the gradients over ranks; then every
rank clips the (identical) averaged gradient to ``max_norm`` and takes the same Nesterov step.  ``oracle_ddp_steps``
restates exactly that on W oracle replicas in one process.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import host
from oracle.model import OracleDeepSpeech, seeded_state_dict

MODEL_KW = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
NUM_UTTS, BATCH = 22, 4          # 6 bins (the last one short): 3 per rank at world 2
LR, MOMENTUM, MAX_NORM = 2e-2, 0.9, 2.0     # the clip engages on these steps


def utterance(i):
    """Utterance i of a length-sorted synthetic manifest: (frames (T_in,161), labels)."""
    rng = np.random.Generator(np.random.PCG64(1000 + i))
    t_in = 60 + 3 * i
    x = rng.standard_normal((t_in, 161)).astype(np.float32)
    lab = rng.integers(1, 29, size=2 + i % 4).astype(np.int32)
    return x, lab


def batch_of(ids):
    """collate (codes/data.py:107-164) of the given utterances."""
    return host.collate([utterance(i) for i in ids])


def oracle_ddp_steps(world, seed=7):
    """Returns (list over ranks of per-step losses, final state dict of rank 0, final weights of every rank)."""
    replicas = []
    for _ in range(world):
        m = OracleDeepSpeech(**MODEL_KW)
        m.load_state_dict(seeded_state_dict(m, seed))          # DDP construction: rank 0's weights everywhere
        m.train()
        replicas.append(m)
    opts = [torch.optim.SGD(m.parameters(), lr=LR, momentum=MOMENTUM, nesterov=True) for m in replicas]
    bins = [host.ddp_bins(NUM_UTTS, BATCH, world, r) for r in range(world)]
    losses = [[] for _ in range(world)]
    for step in range(len(bins[0])):
        for r, m in enumerate(replicas):
            inputs, targets, pct, sizes = batch_of(bins[r][step])
            x = torch.from_numpy(inputs)
            logits = m(x)
            out_sizes = torch.from_numpy(host.out_sizes(pct, logits.shape[1]))
            loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), torch.from_numpy(targets).long(), out_sizes.long(),
                              torch.from_numpy(sizes).long(), blank=0, reduction='sum') / x.shape[0]
            opts[r].zero_grad()
            loss.backward()
            losses[r].append(float(loss.item()))
        with torch.no_grad():                                    # DDP: gradients averaged over ranks
            for ps in zip(*[list(m.parameters()) for m in replicas]):
                mean = sum(p.grad for p in ps) / world
                for p in ps:
                    p.grad.copy_(mean)
        for m, o in zip(replicas, opts):
            torch.nn.utils.clip_grad_norm_(m.parameters(), MAX_NORM)
            o.step()
    return losses, replicas[0].state_dict(), [[p.detach().numpy().copy() for p in m.parameters()] for m in replicas]
