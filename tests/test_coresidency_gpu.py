"""Co-residency stress: full trainer steps on the PERSISTENT recurrence kernels while a stand-in
for RCCL's channel kernels -- 32 long-lived workgroups streaming HBM (tests/co_resident_kernel.hip), re-launched back to
back on a third stream -- shares the chip for the whole forward and backward pass.  That is the condition of a
data-parallel step, which no one has been able to run with more than one rank on this pool: the recurrence kernels need
~204 whole CUs co-resident, the side stream's weight-gradient GEMMs fill the rest, and the collective's workgroups hold
their CUs for milliseconds.  Checked: no time-out flag, no fall-back to the launch-per-step kernels, and the same losses
and weights as the launch-per-step kernels give without any neighbour."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_kernels_gpu import co_resident_load  # noqa: E402


def _batches(bsz, seed):
    rng = np.random.default_rng(seed)
    out = []
    for t_max in (601, 481):
        t_ins = [t_max] + sorted((int(v) for v in rng.integers(t_max // 2, t_max, size=bsz - 1)), reverse=True)
        x = rng.standard_normal((bsz, t_max, 161)).astype(np.float32)
        for b, n in enumerate(t_ins):
            x[b, n:] = 0.0
        lens = [max(1, int(0.1 * n)) for n in t_ins]
        labels = rng.integers(1, 29, size=sum(lens)).astype(np.int32)
        pct = torch.tensor([n / float(t_max) for n in t_ins], dtype=torch.float32)
        out.append((torch.from_numpy(x), torch.from_numpy(labels), pct, torch.tensor(lens, dtype=torch.int32)))
    return out


def _run(mode, batches, monkeypatch, load):
    from codes.engine import Trainer
    from codes.model import DeepSpeech
    from ds2hip import ops
    monkeypatch.setattr(ops, 'GRU_MODE', mode)

    def fresh():
        torch.manual_seed(7)
        model = DeepSpeech().to('cuda')                                  # the full 5 x BiGRU-800 model
        opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
        trainer = Trainer(model, opt, device='cuda', max_norm=400)
        assert trainer._fused
        return model, trainer

    # Warm-up on a throw-away copy, on EVERY batch shape, from an emptied allocator cache: a block the caching allocator has to
    # get from (or, when the earlier tests of the process have filled the device, give back to) the driver inside the measured
    # steps is a hipMalloc / hipFree -- and hipFree waits for the whole device, i.e. for the stand-in's queued second of
    # launches: the steps then "take" a second and the stand-in is gone when they end (round 4: failed in the full suite,
    # never stand-alone).  The measured steps start from the seeded weights themselves.
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    warm = fresh()[1]
    for b in batches:
        warm.update(b)
    del warm
    torch.cuda.synchronize()
    model, trainer = fresh()
    trainer.update(batches[0])                                           # (the fresh trainer's own buffers: momentum, readback slots)
    model, trainer = fresh()
    torch.cuda.synchronize()
    held = load() if load is not None else None
    losses, norms, grads = [], [], []
    t_wall = time.time()
    for b in batches:                                                    # synchronous steps: each reads its own flags
        losses.append(trainer.update(b))
        norms.append(trainer.last_grad_norm)
        grads.append(model.flat_grad().detach().clone())                 # raw (unclipped) gradient of that step
    torch.cuda.current_stream().synchronize()                            # (not the device: the stand-in is still running)
    t_wall = time.time() - t_wall
    still_running = held is not None and not held[1].query()
    torch.cuda.synchronize()
    ops.check_async_errors()
    if held is not None:
        held[1].synchronize()
    c = model.conv
    conv_span = model._span(c[0].weight, c[4].bias)
    return dict(losses=losses, norms=norms, grads=grads, weights=model._flat_p.detach().clone(), conv=conv_span,
                still_running=still_running, wall_ms=1e3 * t_wall)


@pytest.mark.parametrize('bsz', [8, 10])
def test_persistent_steps_beside_a_collective_stand_in(bsz, monkeypatch):
    """Tolerances.  Step 1 starts from the same seeded weights in both runs, so its loss and everything the recurrence
    kernels and the side stream's GEMMs produce (the GRU / FC slices of the gradient) are compared tightly.  The conv
    block's slices are not: its activation is a hard clip, the BatchNorm statistics in front of it are atomic sums whose
    last bits depend on arrival order, and with ~8 M conv activations per step one of them sits within 1e-7 of the clip
    boundary in about one seed in four (B = 10, seed 10: element 1621927 of conv2's output normalises to -1.4e-8 ..
    -7.9e-8) -- whether its gradient passes is then decided by those last bits, in either kernel family and with or
    without a neighbour (tools/attic/clip_boundary_probe.py).  One such flip moves conv2's filter gradient by 3 % of its
    largest entry and the NEXT step's loss by 4e-5; the second step is therefore held to 1e-4."""
    from ds2hip import ops
    batches = _batches(bsz, bsz)
    ref = _run('step', batches, monkeypatch, None)
    before = ops.fallback_count
    # 1000 ms of stand-in for ~40 ms of steps: the first update of a fresh trainer allocates (flat gradient, momentum, a
    # page-locked readback slot) while the stand-in already runs, and such an allocation beside a busy chip has taken hundreds of
    # milliseconds (round 4: one failure in five full-suite runs at 400 ms, none stand-alone)
    got = _run('persistent', batches, monkeypatch, lambda: co_resident_load(duration_ms=1000.0))
    assert got['still_running'], ('the stand-in finished before the steps did (%.0f ms of wall time for two steps): it did '
                                  'not share the chip for the whole pass' % got['wall_ms'])
    assert ops.fallback_count == before and not ops._persistent_off     # nothing fell back to the per-step kernels
    np.testing.assert_allclose(got['losses'][0], ref['losses'][0], rtol=2e-6)
    lo, hi = ref['conv']
    g, r = got['grads'][0], ref['grads'][0]
    rest = torch.ones_like(r, dtype=torch.bool)
    rest[lo:hi] = False
    assert float((g - r)[rest].abs().max()) <= 2e-5 * float(r[rest].abs().max())      # GRU, BatchNorm1d and FC slices
    assert float((g - r)[lo:hi].abs().max()) <= 0.05 * float(r[lo:hi].abs().max())    # conv block: admits a clip flip
    np.testing.assert_allclose(got['norms'][0], ref['norms'][0], rtol=2e-4)
    np.testing.assert_allclose(got['losses'][1], ref['losses'][1], rtol=1e-4)
    np.testing.assert_allclose(got['norms'][1], ref['norms'][1], rtol=1e-3)
    assert float((got['weights'] - ref['weights']).abs().max()) <= 2e-4
