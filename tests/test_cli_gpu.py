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


def test_finetune_cli_en_checkpoint_to_pt_br_head(tmp_path):
    """BASELINE configs[4] through the CLI (reference README.md:227-250, codes/utils/training_utils.py:87-122): train a tiny
    EN model, then ``train.py --continue-from <EN ckpt> scripts/pt_BR-finetune-accents-map-fc.json``: the 29-way head is
    swapped for the 43-way one, the rows named by data/map_en-pt_BR.json (found through --data-dir) are copied, the run
    starts as a NEW run (epoch / iteration / optimizer state not resumed), and test.py decodes the result with the pt_BR
    labels."""
    _corpus(tmp_path)
    (tmp_path / 'map_en-pt_BR.json').write_text(open(os.path.join(ROOT, 'data', 'map_en-pt_BR.json')).read())
    common = ['--data-dir', str(tmp_path), '--train-manifest', str(tmp_path / 'train.csv'), '--val-manifest',
              str(tmp_path / 'val.csv'), '--local', '--checkpoint', '--num-workers', '0']
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'tiny.json')] + common +
                         ['--save-folder', str(tmp_path / 'en')], capture_output=True, text=True, env=dict(os.environ),
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    en_ckpt = tmp_path / 'en' / 'tiny' / 'model_ckpt_2.pth'
    en = torch.load(str(en_ckpt), map_location='cpu', weights_only=False)
    cfg = json.load(open(os.path.join(ROOT, 'scripts', 'pt_BR-finetune-accents-map-fc.json')))
    assert cfg['training']['finetune'] is True and cfg['model']['map_fc'] == 'map_en-pt_BR.json'
    cfg['model']['name'] = 'ft'
    cfg['model']['params'] = {'rnn_hidden_size': 32, 'num_rnn_layers': 2}
    cfg['training'].update(num_epochs=1, batch_size=3)
    cfg['optimizer']['params']['lr'] = 0.0                # the weights stay what the surgery made them
    (tmp_path / 'ft.json').write_text(json.dumps(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'ft.json')] + common +
                         ['--continue-from', str(en_ckpt), '--save-folder', str(tmp_path / 'ft')], capture_output=True,
                         text=True, env=dict(os.environ), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    log = out.stderr + out.stdout
    assert 'Changing the last FC layer' in log and 'Mapping FC weights' in log
    assert 'Start epoch:' not in log                       # a fine-tune run does not resume the EN run's counters
    ft = torch.load(str(tmp_path / 'ft' / 'ft' / 'model_ckpt_1.pth'), map_location='cpu', weights_only=False)
    key = 'fc.0.module.1.weight'
    assert tuple(en['state_dict'][key].shape) == (29, 32) and tuple(ft['state_dict'][key].shape) == (43, 32)
    pairs = json.load(open(os.path.join(ROOT, 'data', 'map_en-pt_BR.json')))
    old_idx, new_idx = zip(*pairs)
    assert torch.equal(ft['state_dict'][key][list(new_idx)], en['state_dict'][key][list(old_idx)])
    rest = sorted(set(range(43)) - set(new_idx))
    assert float(ft['state_dict'][key][rest].abs().max()) < 0.1          # the other rows: N(0, 0.01)
    assert ft['epoch'] == 1 and ft['iteration'] == 2 and len(ft['val_metrics']['cer']) == 1
    assert ft['optimizer']['param_groups'][0]['lr'] == 0.0 and ft['args']['config']['model']['langs'] == ['pt_BR']
    for k, v in en['state_dict'].items():                                # lr = 0: the backbone is the EN backbone
        if k != key and 'running' not in k and 'num_batches' not in k:
            assert torch.equal(v, ft['state_dict'][k]), k
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'test.py'), '--model-path',
                          str(tmp_path / 'ft' / 'ft' / 'model_ckpt_1.pth'), '--data-dir', str(tmp_path), '--manifest',
                          str(tmp_path / 'val.csv'), '--batch-size', '2', '--num-workers', '0', '--verbose'],
                         capture_output=True, text=True, env=dict(os.environ), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'Test Summary' in out.stdout and 'Average CER' in out.stdout
    sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd'))
    from codes.utils.model_utils import load_model
    model, _, _, target_t = load_model(str(tmp_path / 'ft' / 'ft' / 'model_ckpt_1.pth'), return_transforms=True,
                                       data_dir=str(tmp_path))
    assert model._num_classes == 43 and len(target_t[0].label_encoder.classes_) == 43


def test_freeze_config_cli_from_an_en_checkpoint(tmp_path):
    """``train.py scripts/pt_BR-finetune-freeze.json --continue-from <EN ckpt>`` (reference README.md:193, the 30.80 % row of
    BASELINE.md) with the shipped file's own ``model`` / ``training.finetune`` / ``optimizer`` / ``scheduler`` values (only the
    model size, epochs and batch size are shrunk): ``finetune: true`` makes the run call ``finetune_model`` -- the one caller of
    ``_freeze_layers`` (training_utils.py:57-91) -- and start as a NEW run: epoch 0, iteration 0, fresh optimizer and scheduler
    (train.py:142-167).  The conv block's parameters come out bit-identical; its BatchNorm running statistics do move, because
    the update step's ``model.train()`` (codes/engine.py:51) returns the frozen BatchNorm to training mode -- the reference's
    literal behaviour (tests/test_host2_cpu.py::test_reference_freeze_leaves_batchnorm_training_under_its_update_step)."""
    _corpus(tmp_path)
    common = ['--data-dir', str(tmp_path), '--train-manifest', str(tmp_path / 'train.csv'), '--val-manifest',
              str(tmp_path / 'val.csv'), '--local', '--checkpoint', '--num-workers', '0']
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'tiny.json')] + common +
                         ['--save-folder', str(tmp_path / 'en')], capture_output=True, text=True, env=dict(os.environ),
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    en_ckpt = tmp_path / 'en' / 'tiny' / 'model_ckpt_2.pth'
    en = torch.load(str(en_ckpt), map_location='cpu', weights_only=False)
    cfg = json.load(open(os.path.join(ROOT, 'scripts', 'pt_BR-finetune-freeze.json')))
    assert cfg['training']['finetune'] is True and cfg['model']['freeze_layers'] == ['conv'] and cfg['model']['langs'] == ['en']
    cfg['model']['params'] = {'rnn_hidden_size': 32, 'num_rnn_layers': 2}
    cfg['training'].update(num_epochs=1, batch_size=3)
    (tmp_path / 'freeze.json').write_text(json.dumps(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), str(tmp_path / 'freeze.json')] + common +
                         ['--continue-from', str(en_ckpt), '--save-folder', str(tmp_path / 'fr')], capture_output=True,
                         text=True, env=dict(os.environ), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    log = out.stderr + out.stdout
    conv_keys = ['conv.0.weight', 'conv.0.bias', 'conv.1.weight', 'conv.1.bias', 'conv.3.weight', 'conv.3.bias',
                 'conv.4.weight', 'conv.4.bias']
    n_frozen = sum(en['state_dict'][k].numel() for k in conv_keys)
    assert 'Freezed {} parameters'.format(n_frozen) in log
    assert 'Changing the last FC layer' not in log           # langs: ["en"]: the 29-way head stays
    assert 'Start epoch:' not in log and 'Epoch: [1][1/2]' in log
    ft = torch.load(str(tmp_path / 'fr' / 'pt_BR-finetune-freeze' / 'model_ckpt_1.pth'), map_location='cpu',
                    weights_only=False)
    assert ft['epoch'] == 1 and ft['iteration'] == 2 and len(ft['val_metrics']['cer']) == 1
    for k in conv_keys:
        assert torch.equal(ft['state_dict'][k], en['state_dict'][k]), k
    moved = [k for k, v in en['state_dict'].items()
             if k not in conv_keys and 'running' not in k and 'num_batches' not in k and not torch.equal(v, ft['state_dict'][k])]
    assert len(moved) == len([k for k in en['state_dict'] if 'running' not in k and 'num_batches' not in k]) - len(conv_keys)
    # BatchNorm of the frozen block: training mode under the update step (2 more batches seen, statistics moved)
    assert int(ft['state_dict']['conv.1.num_batches_tracked']) == int(en['state_dict']['conv.1.num_batches_tracked']) + 2
    assert not torch.equal(ft['state_dict']['conv.1.running_mean'], en['state_dict']['conv.1.running_mean'])
    # a fresh optimizer and scheduler: the freeze file's lr annealed once by ITS gamma, not the EN run's annealed lr
    assert ft['optimizer']['param_groups'][0]['lr'] == pytest.approx(3e-4 * 0.99, rel=1e-12)
    assert en['optimizer']['param_groups'][0]['lr'] == pytest.approx(3e-4 * 0.909090909 ** 2, rel=1e-9)
    # frozen parameters (the first eight of model.parameters(): the conv block) have no momentum that could move them
    state = ft['optimizer']['state']
    params = ft['optimizer']['param_groups'][0]['params']
    assert len(params) == len([k for k in en['state_dict'] if 'running' not in k and 'num_batches' not in k])
    for i in params[:8]:
        buf = state.get(i, {}).get('momentum_buffer')
        assert buf is None or not bool(buf.any())
    assert bool(state[params[8]]['momentum_buffer'].any())
