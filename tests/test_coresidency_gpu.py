"""Co-residency stress: full trainer steps on the PERSISTENT recurrence kernels while a stand-in
for RCCL's channel kernels -- 32 long-lived workgroups streaming HBM (tests/co_resident_kernel.hip), re-launched back to
back on a third stream -- shares the chip for the whole forward and backward pass.  That is the condition of a
data-parallel step, which no one has been able to run with more than one rank on this pool: the recurrence kernels need
~204 whole CUs co-resident, the side stream's weight-gradient GEMMs fill the rest, and the collective's workgroups hold
their CUs for milliseconds.  Checked: no time-out flag, no fall-back to the launch-per-step kernels, and the same losses
and weights as the launch-per-step kernels give without any neighbour."""
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
    torch.manual_seed(7)
    model = DeepSpeech().to('cuda')                                      # the full 5 x BiGRU-800 model
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=400)
    assert trainer._fused
    trainer.update(batches[0])                                           # warm-up (allocations, first-launch checks)
    torch.cuda.synchronize()
    held = load() if load is not None else None
    losses = [trainer.update(b) for b in batches]                        # synchronous steps: each reads its own flags
    torch.cuda.current_stream().synchronize()                            # (not the device: the stand-in is still running)
    still_running = held is not None and not held[1].query()
    torch.cuda.synchronize()
    ops.check_async_errors()
    if held is not None:
        held[1].synchronize()
    return losses, trainer.last_grad_norm, model._flat_p.detach().clone(), still_running


@pytest.mark.parametrize('bsz', [8, 10])
def test_persistent_steps_beside_a_collective_stand_in(bsz, monkeypatch):
    from ds2hip import ops
    batches = _batches(bsz, bsz)
    ref = _run('step', batches, monkeypatch, None)
    before = ops.fallback_count
    got = _run('persistent', batches, monkeypatch, lambda: co_resident_load(duration_ms=400.0))
    assert got[3], 'the stand-in finished before the steps did: it did not share the chip for the whole pass'
    assert ops.fallback_count == before and not ops._persistent_off     # nothing fell back to the per-step kernels
    np.testing.assert_allclose(got[0], ref[0], rtol=2e-5)
    np.testing.assert_allclose(got[1], ref[1], rtol=2e-4)
    assert float((got[2] - ref[2]).abs().max()) <= 5e-6
