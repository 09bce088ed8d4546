"""Smoke check used by __graft_entry__.smoke(): one tiny training step on cuda:0 vs the CPU oracle."""
import numpy as np
import torch
import torch.nn.functional as F


def run_smoke():
    from codes.engine import Trainer
    from codes.model import DeepSpeech
    from oracle.model import OracleDeepSpeech, seeded_state_dict
    from tests.golden.make_golden import seeded_inputs, seeded_labels

    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2, num_classes=29)
    bsz, t_in, lengths, label_lens = 3, 121, [121, 97, 64], [9, 6, 4]
    oracle = OracleDeepSpeech(**kwargs)
    sd = seeded_state_dict(oracle, 1234)
    oracle.load_state_dict(sd)
    model = DeepSpeech(**kwargs)
    model.load_state_dict(sd)
    model.to('cuda')
    x = torch.from_numpy(seeded_inputs(77, bsz, t_in, lengths=lengths))
    labels = torch.from_numpy(seeded_labels(78, label_lens, 29))
    pct = torch.tensor([n / float(t_in) for n in lengths], dtype=torch.float32)
    sizes = torch.tensor(label_lens, dtype=torch.int32)

    # oracle step on the CPU
    oracle.train()
    logits = oracle(x)
    out_sizes = (pct * logits.shape[1]).int()
    loss = F.ctc_loss(logits.transpose(0, 1).log_softmax(-1), labels.long(), out_sizes.long(), sizes.long(),
                      blank=0, reduction='sum') / bsz
    opt_o = torch.optim.SGD(oracle.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    opt_o.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(oracle.parameters(), 400)
    opt_o.step()

    # the HIP path
    opt = torch.optim.SGD(model.parameters(), lr=3e-4, momentum=0.9, nesterov=True)
    trainer = Trainer(model, opt, device='cuda', max_norm=400)
    got = trainer.update((x, labels, pct, sizes))
    assert abs(got - float(loss.item())) <= 1e-4 * abs(float(loss.item())), (got, float(loss.item()))
    worst = 0.0
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        worst = max(worst, float((p.detach().cpu() - q.detach()).abs().max()))
    assert worst < 1e-5, worst
    model.eval()
    oracle.eval()
    with torch.no_grad():
        pg = model(x.to('cuda')).cpu().numpy()
        po = oracle(x).numpy()
    assert np.abs(pg - po).max() < 1e-3
    print('smoke ok: loss %.5f (oracle %.5f), max post-step weight diff %.2e, max prob diff %.2e'
          % (got, float(loss.item()), worst, np.abs(pg - po).max()))
