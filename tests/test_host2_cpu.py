"""CPU checks of the host-side neighbours of the hot path: labels, manifests, samplers, configs, checkpoints,
and the data-parallel partition + bucketed all-reduce under a 2-process gloo group."""
import json
import os
import wave

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_wav(path, samples):
    with wave.open(path, 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes((np.clip(samples, -1, 1) * 32767).astype('<i2').tobytes())


def test_to_label_and_label_files():
    from codes.transforms import ToLabel
    from codes.utils.io_utils import read_labels
    en = read_labels(os.path.join(ROOT, 'data', 'labels.en.json'))
    pt = read_labels(os.path.join(ROOT, 'data', 'labels.pt_BR.json'))
    assert len(en) == 29 and len(pt) == 43 and en[0] == '_' and pt[0] == '_' and en[2] == "'"
    t_en = ToLabel(os.path.join(ROOT, 'data', 'labels.en.json'), lang='en', remove_accents=True)
    ids = t_en("Olá, it's 42!")
    assert ids.shape == (len("OLA ITS "), 1) or ids.ndim == 2
    assert ''.join(en[i] for i in ids[:, 0]) == "OLA IT'S "
    t_pt = ToLabel(os.path.join(ROOT, 'data', 'labels.pt_BR.json'), lang='pt_BR', remove_accents=False)
    assert ''.join(pt[i] for i in t_pt('ação')[:, 0]) == 'AÇÃO'
    pairs = json.load(open(os.path.join(ROOT, 'data', 'map_en-pt_BR.json')))
    assert all(en[a] == pt[b] for a, b in pairs) and len(pairs) == 28


def test_dataset_loader_and_samplers(tmp_path):
    from codes.data import AudioDataLoader, AudioDataset
    from codes.sampler import BucketingSampler, DistributedBucketingSampler
    from codes.transforms import Compose, ToLabel, ToTensor
    rng = np.random.default_rng(0)
    rows = []
    for i, n in enumerate((16000, 17000, 20000, 24000, 30000)):
        _write_wav(str(tmp_path / ('a%d.wav' % i)), 0.1 * rng.standard_normal(n))
        (tmp_path / ('a%d.txt' % i)).write_text('hello world %d\n' % i)
        rows.append('a%d.wav,a%d.txt,%.3f' % (i, i, n / 16000.0))
    (tmp_path / 'm.csv').write_text('\n'.join(rows) + '\n')
    ds = AudioDataset(str(tmp_path), str(tmp_path / 'm.csv'), Compose([ToTensor()]),
                      ToLabel(os.path.join(ROOT, 'data', 'labels.en.json')))
    wav, lab = ds[2]
    assert wav.shape == (20000,) and wav.dtype == torch.float32 and lab.shape == (12, 1)
    sampler = BucketingSampler(ds, batch_size=2)
    assert [sorted(b) for b in sampler.bins] == [[0, 1], [2, 3], [4]]
    loader = AudioDataLoader(ds, batch_sampler=sampler, raw_audio=True)
    wavs, targets, pct, sizes = next(iter(loader))
    assert len(wavs) == 2 and pct is None and sizes.tolist() == [12, 12] and targets.dtype == torch.int32
    for w in range(2):
        s = DistributedBucketingSampler(ds, batch_size=2, num_replicas=2, rank=w)
        assert list(s) == host.ddp_bins(5, 2, 2, w)


def test_configs_and_model_factory():
    from codes.utils import training_utils as tu
    from codes.utils.io_utils import AttrDict, expand_values
    for name in ('librispeech-from_scratch', 'pt_BR-from_scratch', 'pt_BR-finetune', 'pt_BR-finetune-freeze'):
        cfg = AttrDict(json.load(open(os.path.join(ROOT, 'scripts', name + '.json'))))
        assert set(cfg.keys()) == {'model', 'training', 'optimizer', 'scheduler'}
        assert cfg.optimizer.params.nesterov is True and cfg.training.max_norm == 400
    cfg = AttrDict(json.load(open(os.path.join(ROOT, 'scripts', 'librispeech-from_scratch.json'))))
    assert cfg.training.batch_size == 10
    cfg.model['params'] = {'rnn_hidden_size': 32, 'num_rnn_layers': 2}
    model = tu.get_model(cfg.model)
    assert model.fc[0].module[1].weight.shape == (29, 32)
    opt = tu.get_optimizer(tu.get_per_params_lr(model, cfg.optimizer), cfg.optimizer)
    sch = tu.get_scheduler(opt, cfg.scheduler)
    assert isinstance(opt, torch.optim.SGD) and opt.param_groups[0]['nesterov']
    assert abs(sch.gamma - 0.909090909) < 1e-12
    assert expand_values({'a': ['{x}/b', {'c': '{x}'}]}, x='r') == {'a': ['r/b', {'c': 'r'}]}


def test_finetune_surgery_and_freeze():
    from codes.utils import training_utils as tu
    from codes.utils.io_utils import AttrDict
    cfg = AttrDict({'langs': ['pt_BR'], 'freeze_layers': ['conv'], 'map_fc': os.path.join(ROOT, 'data', 'map_en-pt_BR.json'),
                    'params': {'rnn_hidden_size': 32, 'num_rnn_layers': 2, 'num_classes': 29}})
    from codes.model import DeepSpeech
    model = DeepSpeech(**cfg.params)
    old_w = model.fc[0].module[1].weight.detach().clone()
    model = tu.finetune_model(model, cfg)
    new_w = model.fc[0].module[1].weight
    assert new_w.shape == (43, 32)
    assert torch.equal(new_w[2], old_w[3]) and torch.equal(new_w[0], old_w[0])      # 'A': EN row 3 -> PT row 2
    assert float(new_w[30].detach().abs().max()) < 0.1                                       # unmapped rows ~ N(0, 0.01)
    assert not any(p.requires_grad for p in model.conv.parameters())
    assert model.conv[1].frozen_stats and all(p.requires_grad for p in model.rnns.parameters())
    assert 'fc.0.module.1.weight' in model.state_dict() and model.state_dict()['fc.0.module.1.weight'].shape[0] == 43


def test_checkpoint_roundtrip(tmp_path):
    from codes.model import DeepSpeech
    from codes.utils import model_utils as mu
    from codes.utils.io_utils import AttrDict
    args = AttrDict({'data_dir': os.path.join(ROOT, 'data'),
                     'config': {'model': {'name': 'x', 'langs': ['en'], 'params': {'rnn_hidden_size': 32,
                                                                                  'num_rnn_layers': 2}}}})
    model = DeepSpeech(rnn_hidden_size=32, num_rnn_layers=2)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True)
    ckpt = mu.make_checkpoint(args, model, opt, None, 3, 77, val_metrics={'cer': 12.5})
    assert set(ckpt) == {'args', 'state_dict', 'optimizer', 'scheduler', 'epoch', 'iteration', 'metrics', 'val_metrics'}
    path = str(tmp_path / 'model_ckpt_3.pth')
    torch.save(ckpt, path)
    again, back = mu.load_model(path, return_ckpt=True)
    assert back['epoch'] == 3 and back['iteration'] == 77 and back['val_metrics']['cer'] == 12.5
    for k, v in model.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])


def _ddp_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from codes.model import DeepSpeech
        from codes.sampler import DistributedBucketingSampler
        sampler = DistributedBucketingSampler(list(range(23)), batch_size=4)       # rank / world from the group
        mine = list(sampler)
        assert mine == host.ddp_bins(23, 4, world, rank)
        # the trainer's exchange: per-layer slices of ONE flat gradient buffer, summed, then averaged in the update
        torch.manual_seed(0)
        model = DeepSpeech(rnn_hidden_size=32, num_rnn_layers=2)
        model.flatten_parameters()
        dist.broadcast(model._flat_p, 0)
        g = torch.full_like(model._flat_p, float(rank + 1))
        spans = [model._span(model.fc[0].module[0].weight, model.fc[0].module[1].weight)]
        for layer in reversed(list(model.rnns)):
            first = layer.batch_norm.module.weight if layer.batch_norm is not None else layer.rnn.weight_ih_l0
            spans.append(model._span(first, layer.rnn.weight_hh_l0_reverse))
        spans.append(model._span(model.conv[0].weight, model.conv[4].bias))
        covered = torch.zeros_like(g)
        for lo, hi in spans:
            dist.all_reduce(g[lo:hi])
            covered[lo:hi] += 1
        assert bool((covered == 1).all()), 'slices must tile the flat buffer exactly once'
        assert bool((g == sum(range(1, world + 1))).all())
        out.put((rank, len(mine), float(model._flat_p.sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_partition_and_bucketed_allreduce():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=10) for _ in range(2))
    assert got[0][1] == got[1][1] == 3                 # ceil(6 bins / 2 ranks)
    assert got[0][2] == got[1][2]                      # parameters identical after the rank-0 broadcast
