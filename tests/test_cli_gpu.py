"""CLI surface on the GPU: train.py (one tiny epoch, checkpoint) -> test.py (greedy decode, WER/CER), and the
trainer's distributed code path under a 1-rank RCCL group."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus(tmp_path, n=6):
    rng = np.random.default_rng(0)
    rows = []
    words = ['hello', 'world', 'speech', 'test', 'amd', 'gpu']
    for i in range(n):
        ns = 16000 + 1700 * i
        with wave.open(str(tmp_path / ('u%d.wav' % i)), 'wb') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.clip(0.1 * rng.standard_normal(ns), -1, 1) * 32767).astype('<i2').tobytes())
        (tmp_path / ('u%d.txt' % i)).write_text(words[i % 6] + ' ' + words[(i + 1) % 6] + '\n')
        rows.append('u%d.wav,u%d.txt,%.3f' % (i, i, ns / 16000.0))
    (tmp_path / 'train.csv').write_text('\n'.join(rows) + '\n')
    (tmp_path / 'val.csv').write_text('\n'.join(rows[:3]) + '\n')
    for f in ('labels.en.json', 'labels.pt_BR.json'):
        (tmp_path / f).write_text(open(os.path.join(ROOT, 'data', f)).read())
    cfg = json.load(open(os.path.join(ROOT, 'scripts', 'librispeech-from_scratch.json')))
    cfg['model']['name'] = 'tiny'
    cfg['model']['params'] = {'rnn_hidden_size': 32, 'num_rnn_layers': 2}
    cfg['training'].update(num_epochs=2, batch_size=3, augment=True)      # tempo + gain on the training set
    (tmp_path / 'tiny.json').write_text(json.dumps(cfg))


def test_train_then_test_cli(tmp_path):
    _corpus(tmp_path)
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'tiny.json'), '--data-dir',
                          str(tmp_path), '--train-manifest', str(tmp_path / 'train.csv'), '--val-manifest',
                          str(tmp_path / 'val.csv'), '--local', '--checkpoint', '--num-workers', '0', '--save-folder',
                          str(tmp_path / 'results')], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    log = out.stderr + out.stdout
    assert 'Validation Summary Epoch: [2]' in log and 'Epoch: [1][1/2]' in log
    ckpt = tmp_path / 'results' / 'tiny' / 'model_ckpt_2.pth'
    assert ckpt.exists()
    payload = torch.load(str(ckpt), map_location='cpu', weights_only=False)
    assert payload['epoch'] == 2 and 'rnns.1.batch_norm.module.running_mean' in payload['state_dict']
    # epoch-end behaviour of the reference (train.py:272-374): train + val metric histories, metrics-log, best-CER files
    assert payload['iteration'] == 4
    for hist in (payload['metrics'], payload['val_metrics']):
        assert sorted(hist) == ['cer', 'ctcloss', 'wer'] and all(len(v) == 2 for v in hist.values())
    lines = (tmp_path / 'results' / 'tiny' / 'metrics-log').read_text().splitlines()
    assert len(lines) == 2 and lines[1].startswith('Epoch [2] | Train ctcloss ') and '| Val ctcloss ' in lines[1]
    assert (tmp_path / 'results' / 'tiny' / 'model_best-ckpt_1.pth').exists()
    assert 'Training Summary Epoch: [1]' in log and 'Annealing learning rate' in log
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'test.py'), '--model-path', str(ckpt), '--data-dir',
                          str(tmp_path), '--manifest', str(tmp_path / 'val.csv'), '--batch-size', '2', '--num-workers',
                          '0', '--cuda'], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'Test Summary' in out.stdout and 'Average CER' in out.stdout
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'test.py'), '--model-path', str(ckpt), '--data-dir',
                          str(tmp_path), '--manifest', str(tmp_path / 'val.csv'), '--batch-size', '2', '--num-workers',
                          '0', '--decoder', 'beam', '--beam-width', '8'], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'Test Summary' in out.stdout and 'Average CER' in out.stdout


def test_resume_from_a_mid_epoch_checkpoint(tmp_path):
    """--checkpoint-per-batch + --continue-from (train.py:127-157,389-399): the resumed run skips the batches the
    checkpoint already covers, counts iterations once, starts from shuffled bins, and appends to the histories."""
    _corpus(tmp_path)
    base = [sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'tiny.json'), '--data-dir', str(tmp_path),
            '--train-manifest', str(tmp_path / 'train.csv'), '--val-manifest', str(tmp_path / 'val.csv'), '--local',
            '--num-workers', '0', '--save-folder', str(tmp_path / 'results')]
    out = subprocess.run(base + ['--checkpoint-per-batch', '1'], capture_output=True, text=True, env=dict(os.environ),
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    mid = tmp_path / 'results' / 'tiny' / 'model_batch-ckpt_3.pth'           # epoch 2 (of 2), one of its two batches done
    payload = torch.load(str(mid), map_location='cpu', weights_only=False)
    assert payload['epoch'] == 1 and payload['iteration'] == 3 and len(payload['val_metrics']['cer']) == 1
    out = subprocess.run(base + ['--continue-from', str(mid), '--checkpoint', '--save-folder',
                                 str(tmp_path / 'resumed')], capture_output=True, text=True, env=dict(os.environ),
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    log = out.stderr + out.stdout
    assert 'Start epoch: 1. Start iteration 3' in log and 'Shuffling batches for the following epochs' in log
    assert 'Epoch: [2][2/2]' in log and 'Epoch: [2][1/2]' not in log and 'Epoch: [1][' not in log
    final = torch.load(str(tmp_path / 'resumed' / 'tiny' / 'model_ckpt_2.pth'), map_location='cpu', weights_only=False)
    assert final['epoch'] == 2 and final['iteration'] == 4
    assert len(final['val_metrics']['cer']) == 2 and final['val_metrics']['cer'][0] == payload['val_metrics']['cer'][0]


def test_train_cli_distributed_launch(tmp_path):
    """train.py without --local under torch.distributed.run (one rank): RCCL group, DistributedBucketingSampler,
    rank-0 checkpointing."""
    _corpus(tmp_path)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                          '--master-addr', '127.0.0.1', '--master-port', '29641', os.path.join(ROOT, 'train.py'),
                          str(tmp_path / 'tiny.json'), '--data-dir', str(tmp_path), '--train-manifest',
                          str(tmp_path / 'train.csv'), '--val-manifest', str(tmp_path / 'val.csv'), '--checkpoint',
                          '--num-workers', '0', '--save-folder', str(tmp_path / 'results_ddp')],
                         capture_output=True, text=True, env=dict(os.environ), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'Validation Summary Epoch: [2]' in (out.stderr + out.stdout)
    assert (tmp_path / 'results_ddp' / 'tiny' / 'model_ckpt_2.pth').exists()


def test_trainer_distributed_path_single_rank():
    """With a process group initialised the trainer broadcasts, all-reduces per-layer slices on a side stream and
    folds 1/world into the update: with world = 1 the result must equal the non-distributed step (up to atomic-add ordering)."""
    import torch.distributed as dist
    from codes.engine import Trainer
    from codes.model import DeepSpeech
    from oracle.model import OracleDeepSpeech, seeded_state_dict
    from tests.golden.make_golden import seeded_inputs
    kwargs = dict(rnn_hidden_size=32, num_rnn_layers=2)
    sd = seeded_state_dict(OracleDeepSpeech(**kwargs), 5)
    x = torch.from_numpy(seeded_inputs(3, 3, 100, lengths=[100, 80, 60]))
    labels = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8, 9], dtype=torch.int32)
    pct = torch.tensor([1.0, 0.8, 0.6])
    sizes = torch.tensor([4, 3, 2], dtype=torch.int32)

    def run():
        model = DeepSpeech(**kwargs)
        model.load_state_dict(sd)
        model.to('cuda')
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
        tr = Trainer(model, opt, device='cuda', max_norm=400)
        losses = [tr.update((x, labels, pct, sizes)) for _ in range(2)]
        return losses, model._flat_p.clone(), tr

    ref_losses, ref_p, _ = run()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29617')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        losses, p, tr = run()
        assert tr.distributed and tr.overlap and tr._comm_stream is not None
    finally:
        dist.destroy_process_group()
    assert losses == pytest.approx(ref_losses, rel=1e-6)
    # split-K GEMMs and the conv wgrad accumulate with float atomics: the last bits depend on arrival order
    assert float((p - ref_p).abs().max()) < 1e-6
