"""Shared checks for the full-size golden fixtures (tests/golden/ref_full_b*.npz, ref_ft43_b16.npz).

The fixtures hold what the REFERENCE's own ``codes/model.py`` computed (tests/golden/make_golden.py) for the
5xBiGRU-800 model at the batch sizes of BASELINE configs[1..4]: every ``tstride``-th output frame of the train
logits and eval probabilities, the eval-mode argmax (+ runner-up, + near-tie mask) of EVERY frame, the CTC loss,
per-parameter gradient norms and strided gradient samples, and the BatchNorm running statistics after the step.
``check_against_golden`` compares any implementation's numbers with them; the oracle test (CPU) and the HIP test
(-m gpu) both call it.
"""
import os

import numpy as np

from tests.golden.make_golden import CASES, seeded_inputs, seeded_labels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FT43_SEED = 4343          # seed of the un-mapped rows of the swapped FC layer (make_golden.run_case)


def case_inputs(name):
    kw = CASES[name]
    nalpha = 43 if kw.get('finetune43') else kw['model_kwargs'].get('num_classes', 29)
    x = seeded_inputs(77, kw['bsz'], kw['t_in'], lengths=kw['lengths'])
    labels = seeded_labels(78, kw['label_lens'], nalpha)
    return kw, x, labels, nalpha


def check_against_golden(g, logits, loss_sum, grads, buffers, probs, logit_tol, prob_tol, gnorm_rtol, gsample_rtol,
                         buf_rtol=1e-4, buf_atol=1e-5):
    """logits / probs: full (B,T,A) numpy arrays of the implementation under test; grads / buffers: name -> array."""
    ts = int(g['tstride']) if 'tstride' in g.files else 1
    np.testing.assert_allclose(logits[:, ::ts], g['logits'], rtol=0, atol=logit_tol)
    ref_loss = float(g['loss_sum'])
    assert abs(loss_sum - ref_loss) <= 1e-4 * abs(ref_loss), (loss_sum, ref_loss)
    for k, gr in grads.items():
        gn = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        ref_gn = float(g['gnorm_' + k])
        if k in ('conv.0.bias', 'conv.3.bias'):
            # a bias in front of a BatchNorm has an exactly-zero gradient; both sides hold only round-off -- of sums whose
            # terms scale with the layer's filter gradient (2e-5 of it in the reference's own fp32 run at 15 s)
            scale = max(1.0, float(g['gnorm_' + k.replace('bias', 'weight')]) / 100.0)
            assert gn < 1e-2 * scale and ref_gn < 1e-2 * scale, k
            continue
        assert abs(gn - ref_gn) <= gnorm_rtol * ref_gn + 1e-6, (k, gn, ref_gn)
        flat = gr.reshape(-1)
        stride = max(1, flat.shape[0] // 1024)
        ref = g['gsample_' + k]
        # the reference's own fp32 gradient is ``gnoise`` away from the exact (float64) one: allow the same distance
        noise = 2.0 * float(g['gnoise_' + k]) if 'gnoise_' + k in g.files else 0.0
        np.testing.assert_allclose(flat[::stride][:1024], ref, rtol=gsample_rtol,
                                   atol=max(gsample_rtol * np.abs(ref).max(), noise) + 1e-7, err_msg=k)
    for k, v in buffers.items():
        np.testing.assert_allclose(v, g['buf_' + k], rtol=buf_rtol, atol=buf_atol, err_msg=k)
    np.testing.assert_allclose(probs[:, ::ts], g['probs'], rtol=0, atol=prob_tol)
    if 'argmax' in g.files:
        am = probs.argmax(-1)
        sizes = g['out_sizes']
        valid = np.arange(am.shape[1])[None, :] < sizes[:, None]
        exact = (am == g['argmax'])
        tie_ok = g['near_tie'] & (am == g['argmax2'])
        bad = valid & ~(exact | tie_ok)
        assert not bad.any(), 'argmax differs from the reference on %d confident frames' % int(bad.sum())
        assert (exact | ~valid).mean() > 0.97
