"""CPU checks of the host-side neighbours of the hot path: labels, manifests, samplers, configs, checkpoints,
and the data-parallel partition + bucketed all-reduce under a 2-process gloo group."""
import json
import os
import sys
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
    # workers hand on the int16 samples (+ drawn augmentation); decode / tempo / gain / STFT happen on the device
    ds = AudioDataset(str(tmp_path), str(tmp_path / 'm.csv'), Compose([ToTensor(defer=True)]),
                      ToLabel(os.path.join(ROOT, 'data', 'labels.en.json')))
    wav, lab = ds[2]
    assert wav.pcm.shape == (20000,) and wav.pcm.dtype == torch.int16 and wav.tempo is None and lab.shape == (12, 1)
    sampler = BucketingSampler(ds, batch_size=2)
    assert [sorted(b) for b in sampler.bins] == [[0, 1], [2, 3], [4]]
    loader = AudioDataLoader(ds, batch_sampler=sampler, raw_audio=True, num_workers=2)
    from codes.transforms import RawAudioBatch
    wavs, targets, pct, sizes = next(iter(loader))
    assert isinstance(wavs, RawAudioBatch) and len(wavs) == 2 and wavs.pcm.dtype == torch.int16
    assert sorted(wavs.offsets[i + 1] - wavs.offsets[i] for i in range(2)) == [16000, 17000] and wavs.tempos is None
    assert pct is None and sizes.tolist() == [12, 12] and targets.dtype == torch.int32
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

# Every leaf value of the eight shipped configs, as /root/reference/scripts/<same name>.json has it (a literal table: the
# reference tree does not travel).  `training.finetune` of pt_BR-finetune-freeze is `true` there
# (/root/reference/scripts/pt_BR-finetune-freeze.json:12): it is what makes train.py call finetune_model, the only caller of
# _freeze_layers, and start from epoch 0 with a fresh optimizer (train.py:142-167).
_SGD = {'optimizer.name': 'SGD', 'optimizer.params.lr': 3e-4, 'optimizer.params.momentum': 0.9,
        'optimizer.params.nesterov': True, 'scheduler.name': 'ExponentialLR'}
_PT = dict(_SGD, **{'training.num_epochs': 100, 'training.batch_size': 32, 'training.max_norm': 400, 'training.augment': True,
                    'scheduler.params.gamma': 0.99})
SHIPPED_CONFIGS = {
    'example': dict(_PT, **{'model.name': 'example', 'model.freeze_layers': 'all', 'model.langs': ['pt_BR'],
                            'model.map_fc': 'map_en-pt_BR.json', 'model.params.rnn_hidden_size': 800,
                            'training.finetune': False, 'training.task_weights': [1],
                            'optimizer.per_layer_lr': [['base'], ['fc', 3e-3]]}),
    'librispeech-from_scratch': dict(_SGD, **{'model.name': 'librispeech-from_scratch', 'model.langs': ['en'],
                                              'training.num_epochs': 15, 'training.batch_size': 10, 'training.max_norm': 400,
                                              'training.augment': False, 'scheduler.params.gamma': 0.909090909}),
    'pt_BR-finetune-accents-map-fc': dict(_PT, **{'model.name': 'pt_BR-finetune-accents-map-fc', 'model.langs': ['pt_BR'],
                                                  'model.map_fc': 'map_en-pt_BR.json', 'training.finetune': True}),
    'pt_BR-finetune-accents-random-fc': dict(_PT, **{'model.name': 'pt_BR-finetune-accents-random-fc',
                                                     'model.langs': ['pt_BR'], 'training.finetune': True}),
    'pt_BR-finetune-freeze': dict(_PT, **{'model.name': 'pt_BR-finetune-freeze', 'model.freeze_layers': ['conv'],
                                          'model.langs': ['en'], 'training.finetune': True}),
    'pt_BR-finetune': dict(_PT, **{'model.name': 'pt_BR-finetune', 'model.langs': ['en'], 'training.finetune': False}),
    'pt_BR-from_scratch-accents': dict(_PT, **{'model.name': 'pt_BR-from_scratch-accents', 'model.langs': ['pt_BR'],
                                               'training.finetune': False}),
    'pt_BR-from_scratch': dict(_PT, **{'model.name': 'pt_BR-from_scratch', 'model.langs': ['en']}),
}


def _leaves(d, prefix=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_leaves(v, prefix + k + '.'))
        else:
            out[prefix + k] = v
    return out


def test_shipped_configs_hold_the_reference_values():
    names = sorted(f[:-5] for f in os.listdir(os.path.join(ROOT, 'scripts')) if f.endswith('.json'))
    assert names == sorted(SHIPPED_CONFIGS)
    for name, want in SHIPPED_CONFIGS.items():
        got = _leaves(json.load(open(os.path.join(ROOT, 'scripts', name + '.json'))))
        assert got == want, (name, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)})
        for k, v in want.items():                                  # 400 vs 400.0, true vs 1 would compare equal
            assert type(got[k]) is type(v), (name, k)
    if os.path.isdir('/root/reference/scripts'):                   # in the build container: the table against the files
        for name, want in SHIPPED_CONFIGS.items():
            assert _leaves(json.load(open('/root/reference/scripts/%s.json' % name))) == want, name


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
    # BatchNorm of the frozen layer in eval mode (training_utils.py:52-54,73) -- until the next model.train(), which the
    # reference's update step calls before every forward pass (codes/engine.py:51)
    assert not model.conv[1].training and not model.conv[4].training and model.fc[0].module[0].training
    assert all(p.requires_grad for p in model.rnns.parameters())
    model.train()
    assert model.conv[1].training and not any(p.requires_grad for p in model.conv.parameters())
    assert 'fc.0.module.1.weight' in model.state_dict() and model.state_dict()['fc.0.module.1.weight'].shape[0] == 43


def test_reference_freeze_leaves_batchnorm_training_under_its_update_step():
    """What ``freeze_layers`` does to BatchNorm in the reference, on stock torch modules (and, in the build container, on the
    reference's own ``DeepSpeech``): ``_freeze_layers`` applies ``batch_norm_eval_mode`` to the layer
    (training_utils.py:52-54,73), and ``_update`` calls ``model.train()`` before every forward pass (codes/engine.py:51) --
    ``Module.train`` recurses, so the frozen layer's BatchNorm normalises with batch statistics and moves its running
    estimates during training.  The product does the same (Trainer.update), see test_fused_step_with_param_groups_and_frozen_conv."""
    nets = []
    from oracle.model import OracleDeepSpeech
    nets.append(OracleDeepSpeech(rnn_hidden_size=16, num_rnn_layers=1))
    if os.path.isdir('/root/reference/codes'):
        import importlib.util
        spec = importlib.util.spec_from_file_location('ref_model_for_freeze', '/root/reference/codes/model.py')
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        nets.append(ref.DeepSpeech(rnn_hidden_size=16, num_rnn_layers=1))
    for net in nets:
        def batch_norm_eval_mode(m):
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.eval()
        net.conv.apply(batch_norm_eval_mode)
        for p in net.conv.parameters():
            p.requires_grad = False
        assert not net.conv[1].training
        net.train()                                              # codes/engine.py:51
        assert net.conv[1].training and net.conv[4].training
        before = net.conv[1].running_mean.clone()
        net(torch.randn(2, 60, 161))
        assert int(net.conv[1].num_batches_tracked) == 1 and not torch.equal(net.conv[1].running_mean, before)


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


def test_older_checkpoint_formats(tmp_path):
    """The two older payloads the reference still reads (model_utils.py:26-54 and :80-124)."""
    from codes.model import DeepSpeech
    from codes.utils import model_utils as mu
    # (1) version 0.0.1: geometry in seconds, inline labels, stray BatchNorm keys on the first recurrent layer
    labels = "_'ABCDEFGHIJKLMNOPQRSTUVWXYZ "
    model = DeepSpeech(rnn_hidden_size=32, num_rnn_layers=2, num_classes=len(labels))
    sd = dict(model.state_dict())
    for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
        sd['rnns.0.batch_norm.module.' + leaf] = torch.zeros(3)
    legacy = {'version': '0.0.1', 'audio_conf': {'sample_rate': 16000, 'window_size': 0.02, 'window_stride': 0.01},
              'hidden_size': 32, 'hidden_layers': 2, 'rnn_type': 'gru', 'labels': labels, 'bidirectional': True,
              'state_dict': sd}
    path = str(tmp_path / 'legacy.pth')
    torch.save(legacy, path)
    again, front, to_label = mu.load_model(path)
    for k, v in model.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])
    assert again._num_classes == 29 and len(front.transforms) == 2
    assert to_label('ab c').reshape(-1).tolist() == [2, 3, 28, 4]
    # (2) current payload whose args.config is still in the network/transforms schema
    old_cfg = {'network': {'name': 'deepspeech', 'params': {'rnn_hidden_size': 32, 'num_rnn_layers': 2}},
               'transforms': {'label': [{'params': {'labels': '{data_dir}/labels.en.json'}}],
                              'train': [{'params': {'augment': False}}]},
               'training': {'num_epochs': 7, 'max_norm': 400, 'learning_rate': 3e-4, 'momentum': 0.9,
                            'learning_anneal': 0.99}}
    ckpt = {'args': {'data_dir': os.path.join(ROOT, 'data'), 'batch_size': 10, 'finetune': False, 'config': old_cfg},
            'state_dict': model.state_dict()}
    path2 = str(tmp_path / 'old_schema.pth')
    torch.save(ckpt, path2)
    m2, back = mu.load_model(path2, return_ckpt=True)
    assert torch.equal(m2.state_dict()['fc.0.module.1.weight'], model.state_dict()['fc.0.module.1.weight'])
    from codes.utils.io_utils import AttrDict
    up = mu.upgrade_config(AttrDict(ckpt['args'])).config
    assert up.model.langs == ['en'] and up.model.params.num_classes == 29 and up.training.batch_size == 10
    assert up.optimizer.params.nesterov is True and up.scheduler.params.gamma == 0.99 and up.training.num_epochs == 7


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
        out.put((rank, [tuple(b) for b in mine], float(model._flat_p.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_gloo_partition_and_bucketed_allreduce(world):
    """The data-parallel step's two host-side rules on a real process group: the bin partition (codes/sampler.py:113-125 --
    every world-th bin, the list wrapped so that each rank gets ceil(nbins / world); world = 3: not a power of two,
    world = 8 > 6 bins: two ranks take wrapped bins) and the flat gradient's per-layer slices, summed over the ranks."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=10) for _ in range(world))
    nbins = 6                                           # 23 utterances in bins of 4
    per = -(-nbins // world)
    assert all(len(g[1]) == per for g in got)           # every rank runs the same number of steps (no rank starves a collective)
    all_bins = [tuple(range(4 * i, min(4 * i + 4, 23))) for i in range(nbins)]
    seen = [b for g in got for b in g[1]]
    assert set(seen) == set(all_bins)                   # every bin is trained on
    assert len(seen) == per * world                     # ... the wrapped ones twice
    assert len({g[2] for g in got}) == 1                # parameters identical after the rank-0 broadcast


# ------------------------------------------------------------------------------------------- host C++ helpers
def _py_edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def test_edit_distance_host_entry_point():
    from codes.decoder import Decoder, _levenshtein
    rng = np.random.default_rng(3)
    assert _levenshtein('', '') == 0 and _levenshtein('abc', '') == 3 and _levenshtein('', 'xy') == 2
    assert _levenshtein('kitten', 'sitting') == 3
    for _ in range(200):
        a = ''.join(rng.choice(list('abcd '), size=rng.integers(0, 30)))
        b = ''.join(rng.choice(list('abcd '), size=rng.integers(0, 30)))
        assert _levenshtein(a, b) == _py_edit_distance(a, b)
    d = Decoder("_'ABCDEFGHIJKLMNOPQRSTUVWXYZ ")
    assert d.wer('THE CAT SAT', 'THE CAT SAT DOWN') == 1 and d.wer('A B C', 'C B A') == 2
    assert d.cer('HE LLO', 'HELLO') == 0 and d.cer('HELLO', 'HALLO W') == 2


def _best_labelling_by_enumeration(probs, blank=0):
    """Exact most probable labelling of a tiny (T, A) CTC output: sum the path probabilities per collapsed string."""
    import itertools
    t, a = probs.shape
    table = {}
    for path in itertools.product(range(a), repeat=t):
        p = float(np.prod([probs[i, c] for i, c in enumerate(path)]))
        lab, prev = [], None
        for c in path:
            if c != blank and c != prev:
                lab.append(c)
            prev = c
        table[tuple(lab)] = table.get(tuple(lab), 0.0) + p
    best = max(table.items(), key=lambda kv: kv[1])
    return list(best[0]), best[1]


def test_ctc_prefix_beam_search_host_entry_point():
    from codes.decoder import BeamCTCDecoder
    rng = np.random.default_rng(11)
    labels = '_AB'
    for case in range(40):
        t = int(rng.integers(1, 7))
        probs = rng.dirichlet(np.ones(3) * 0.7, size=t).astype(np.float32)
        want, p = _best_labelling_by_enumeration(probs.astype(np.float64))
        dec = BeamCTCDecoder(labels, beam_width=64)
        (got,), (offs,) = dec.decode(torch.from_numpy(probs[None]), None)
        assert got[0] == ''.join(labels[c] for c in want), (case, probs)
        assert abs(dec.last_log_probs[0] - np.log(p)) < 1e-4
        assert len(offs[0]) == len(want) and all(int(offs[0][i]) <= int(offs[0][i + 1]) for i in range(len(want) - 1))
        # log-probability input gives the same answer
        dec_log = BeamCTCDecoder(labels, beam_width=64, log_input=True)
        assert dec_log.decode(torch.from_numpy(np.log(probs)[None]), None)[0][0][0] == got[0]
    # peaked outputs: every beam width agrees with the frame-wise collapse; `sizes` truncates
    ids = np.array([0, 1, 1, 0, 1, 2, 2, 0, 0, 2, 1, 0])
    probs = np.full((1, len(ids), 3), 0.005, dtype=np.float32)
    probs[0, np.arange(len(ids)), ids] = 0.99
    for width in (1, 4, 32):
        dec = BeamCTCDecoder(labels, beam_width=width)
        assert dec.decode(torch.from_numpy(probs), None)[0][0][0] == 'AABBA'
        assert dec.decode(torch.from_numpy(probs), [6])[0][0][0] == 'AAB'
    # the most probable labelling differs from the greedy path when probability mass is split over alignments
    split = np.array([[[0.4, 0.6, 0.0], [0.55, 0.45, 0.0]]], dtype=np.float32)     # greedy path: A,_ -> "A" (also best)
    assert BeamCTCDecoder(labels, beam_width=8).decode(torch.from_numpy(split), None)[0][0][0] == 'A'
    with pytest.raises(ValueError):
        BeamCTCDecoder(labels, beam_width=0)


def test_tempo_and_gain_augmentation(tmp_path):
    """The augmentation's specification (oracle/audio.py: WSOLA tempo keeps the pitch and scales the duration, the gain is
    in dB, 16-bit requantisation clips) and the product's draws: ``ToTensor(augment=True, defer=True)`` draws tempo then
    gain with np.random like the reference (codes/transforms.py:171-182) and hands the int16 samples on untouched."""
    from codes.transforms import RawAudioBatch, ToTensor
    from ds2hip import ops
    from oracle import audio as oa
    sr = 16000
    t = np.arange(3 * sr) / sr
    tone = (0.2 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    for tempo in (0.85, 1.0, 1.15, 1.3):
        y = oa.wsola_tempo(tone, tempo, sr)
        assert len(y) == oa.wsola_out_len(len(tone), tempo, sr) == ops.wsola_schedule(len(tone), tempo, sr)[1]
        assert abs(len(y) - len(tone) / tempo) < 0.1 * sr                     # duration scales by 1 / tempo
        spec = np.abs(np.fft.rfft(y * np.hanning(len(y))))
        peak = np.argmax(spec) * sr / len(y)
        assert abs(peak - 440.0) < 3.0, (tempo, peak)                          # pitch unchanged
        assert 0.8 < float(np.sqrt(np.mean(y ** 2))) / float(np.sqrt(np.mean(tone ** 2))) < 1.1
    assert np.array_equal(oa.wsola_tempo(tone, 1.0, sr), tone)
    # the product's data-independent segment schedule == the oracle's plan, over the augmentation range
    for n in (1500, 16000, 123457, 240000):
        for tempo in (0.85, 0.9004, 0.9995, 1.0005, 1.137, 1.15):
            plan, out_len = oa.wsola_plan(n, tempo, sr)
            bases, m = ops.wsola_schedule(n, tempo, sr)
            assert m == out_len and len(bases) == (0 if plan is None else len(plan)), (n, tempo)
    loud = np.asarray([0.9, -0.9, 0.25, 1e-5, -2e-5], np.float32)
    q = oa.gain_requantize(loud, 8.0)
    assert q[0] == np.float32(32767 / 32768.0) and q[1] == -1.0 and abs(q[2] - 0.25 * 10 ** 0.4) < 1e-4
    assert np.all(q * 32768 == np.round(q * 32768))                            # on the 16-bit grid
    path = str(tmp_path / 'a.wav')
    _write_wav(path, tone)
    plain = ToTensor(defer=True)(path)
    assert plain.tempo is None and plain.gain_db is None
    assert np.array_equal(oa.pcm16_to_float(plain.pcm.numpy()), (tone * 32767).astype('<i2').astype(np.float32) / 32768)
    np.random.seed(5)
    tempo = np.random.uniform(0.85, 1.15)
    gain = np.random.uniform(-6, 8)
    np.random.seed(5)
    aug = ToTensor(augment=True, defer=True)(path)
    assert aug.tempo == tempo and aug.gain_db == gain and torch.equal(aug.pcm, plain.pcm)
    batch = RawAudioBatch.from_clips([plain, aug])
    assert batch.offsets == [0, 3 * sr, 6 * sr] and batch.tempos == [1.0, tempo] and batch.gains_db == [0.0, gain]
    assert 'augment=True' in repr(ToTensor(augment=True))


def test_metrics_log_format_and_best_checkpoints(tmp_path):
    """train.py's epoch-end files: the reference's metrics-log record (train.py:359-374) and the five best-CER
    checkpoints (train.py:223-229, lowest CER kept)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ds2_train_cli', os.path.join(ROOT, 'train.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    log = tmp_path / 'metrics-log'
    train_h, val_h = {'ctcloss': [3.0], 'wer': [90.0], 'cer': [50.0]}, {'ctcloss': [4.0], 'wer': [(95.0, 96.5)], 'cer': [60]}
    cli.write_metrics_log(str(log), 1, train_h, val_h)
    assert log.read_text() == 'Epoch [1] | Train ctcloss 3.000wer 90.000cer 50.000| Val ctcloss 4.000wer 95.000/96.500cer 60\n'
    best = cli.BestCheckpoints(str(tmp_path), n_saved=2)
    for cer in (30.0, 10.0, 20.0, 40.0, 5.0):
        best(cer, {'cer': cer})
    kept = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('model_best-ckpt_'))
    assert kept == ['model_best-ckpt_2.pth', 'model_best-ckpt_5.pth']          # CER 10 and 5
    args = cli.build_parser().parse_args(['cfg.json', '--train-manifest', 'a', '--val-manifest', 'b', '--no-sorta-grad'])
    assert args.no_sorta_grad and not args.no_shuffle


class _Paths(torch.utils.data.Dataset):
    def __init__(self, paths, transform):
        self.paths, self.transform = paths, transform

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        return self.transform(self.paths[i]).numel()


def test_waveform_loader_in_worker_processes(tmp_path):
    """The deferred loader (what the training path builds) is pure host work and runs in forked DataLoader workers; the
    per-utterance form decodes on the GPU and must REFUSE to run in a worker with a clear message instead of dying on a
    re-initialised device, and ``get_data_loaders`` falls back to ``num_workers=0`` for it."""
    from codes.transforms import Compose, PCMClip, ToTensor, waveform_scale
    from codes.utils import training_utils as tu
    paths = []
    for i, n in enumerate((1600, 2400, 3200, 4000)):
        paths.append(str(tmp_path / ('w%d.wav' % i)))
        _write_wav(paths[-1], 0.1 * np.ones(n, np.float32))
    got = list(torch.utils.data.DataLoader(_Paths(paths, ToTensor(defer=True)), batch_size=1, num_workers=2))
    assert [int(g) for g in got] == [1600, 2400, 3200, 4000]
    assert isinstance(ToTensor(defer=True)(paths[0]), PCMClip)
    with pytest.raises(RuntimeError, match='cannot run in a DataLoader worker'):
        list(torch.utils.data.DataLoader(_Paths(paths, ToTensor(defer=False)), batch_size=1, num_workers=2))
    assert tu._is_deferred(Compose([ToTensor(defer=True)])) and not tu._is_deferred(Compose([ToTensor(defer=False)]))
    assert waveform_scale(Compose([ToTensor(defer=True, scale='int32')])) == 65536.0
    assert waveform_scale(Compose([ToTensor(defer=True)])) == 1.0 / 32768.0
    with pytest.raises(ValueError):
        ToTensor(scale=-1.0)


def test_data_parallel_env_is_one_function_for_train_and_bench():
    """RCCL channel cap / hardware-queue count: setdefault semantics (an operator's value wins), and both entry points
    call the same function instead of carrying their own copies of the numbers."""
    from codes.utils.dist_utils import DEFAULTS, assert_no_fallbacks, data_parallel_env
    env = {'NCCL_MAX_NCHANNELS': '16'}
    got = data_parallel_env(env)
    assert got == {'GPU_MAX_HW_QUEUES': DEFAULTS['GPU_MAX_HW_QUEUES'], 'NCCL_MAX_NCHANNELS': '16'} and env == got
    assert_no_fallbacks(0)
    with pytest.raises(RuntimeError, match='fell back'):
        assert_no_fallbacks(2, 'a test')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ('train.py', 'bench.py'):
        src = open(os.path.join(root, name)).read()
        assert 'data_parallel_env()' in src, name
        assert "setdefault('NCCL_MAX_NCHANNELS'" not in src and "setdefault('GPU_MAX_HW_QUEUES'" not in src, name


@pytest.mark.parametrize('world', [1, 2, 4, 8])
def test_bench_bin_plan_is_the_reference_partition(world):
    """bench.py's synthetic corpus under N ranks: rank r times bins r, r + N, ... of the plan (codes/sampler.py:119-125's rule);
    the N bins of one step are ADJACENT in the length-sorted corpus (similar lengths: the step's max over ranks is not set by a
    straggler), every rank sees the same number of bins, the ranks' bins tile the corpus exactly once, and any window of
    consecutive steps has about the corpus's mean clip length (the short / long interleave)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan = bench.bin_plan(10, bench.NUM_BINS * world, world=world)
    assert len(plan) == bench.NUM_BINS * world
    seeds = [p[0] for p in plan]
    assert len(set(seeds)) == len(seeds)                               # every bin of the corpus exactly once
    per_rank = [plan[r::world] for r in range(world)]
    assert all(len(b) == bench.NUM_BINS for b in per_rank)
    means = np.array([[float(np.mean(e[1])) for e in b] for b in per_rank])      # [rank][step] mean clip seconds
    spread = (means.max(0) - means.min(0)) / means.mean(0)
    # one step's bins have similar lengths: N adjacent bins span N x 0.07 s of the corpus -- 6 % of the mean clip for the
    # median step at N = 8, 44 % for the very shortest group (1.0 .. 1.6 s clips), whose step is the cheapest anyway
    assert float(spread.max()) < 0.5 and float(np.median(spread)) < 0.08, spread
    frames = np.array([[bench.frames_of_plan(e) for e in b] for b in per_rank]).sum(0)
    win = np.convolve(frames, np.ones(4) / 4.0, mode='valid')
    assert float(win.max() / win.min()) < 1.6                           # any four consecutive steps: about the mean length
    for e in plan:
        assert np.all(np.diff(e[1]) >= 0) and 1.0 <= e[1][0] and e[1][-1] <= 15.0


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench, root


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` the way the driver runs `--gpus 1` (no launcher, no WORLD_SIZE): bench.py composes the
    torch.distributed.run command itself (one rank per GPU, rendezvous on 127.0.0.1: the reference is started the same way,
    train.py:118-124), runs it as a CHILD (no exec: the box refuses an exec from a process that could have touched the
    GPU), hands the child's one stdout line through and returns its exit code."""
    bench, root = _load_bench()
    cmd = bench.self_launch_command(8, ['--gpus', '8', '--steps', '20', '--warmup', '5'], port=29999)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29999'
    script = cmd.index(os.path.join(root, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']

    seen = {}

    class Done:
        stdout = b'{"metric": "stub", "n_gpus": 4}\n'
        returncode = 7

    def runner(cmd, env=None, stdout=None):
        seen['cmd'], seen['env'] = cmd, env
        return Done()

    monkeypatch.delenv('WORLD_SIZE', raising=False)
    rc = bench.self_launch(4, ['--gpus', '4'], runner=runner)
    assert rc == 7                                                       # the child's code is the parent's
    assert capsys.readouterr().out == '{"metric": "stub", "n_gpus": 4}\n'   # rank 0's line, nothing else, on stdout
    assert seen['cmd'][seen['cmd'].index('--nproc-per-node') + 1] == '4'
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_bench_gpus_n_end_to_end_with_a_stub_launcher(tmp_path):
    """The whole door, in a real process: `python bench.py --gpus 2` with no WORLD_SIZE must reach torch.distributed.run
    (here a stub `torch/distributed/run.py`-shaped module earlier on the path is not possible without shadowing torch, so the
    child is the real launcher and the ranks fail where they should: at `bench.py needs an MI355X`, NOT at an assertion about
    WORLD_SIZE) -- and the parent leaves with a non-zero code of the child's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['MASTER_PORT'] = '29873'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline', '--no-extras'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode()
    assert r.returncode != 0
    assert 'starting 2 ranks' in err
    assert 'AssertionError' not in err
    assert 'needs an MI355X' in err or 'HIP' in err or 'No HIP GPUs' in err or 'invalid device' in err, err[-2000:]
    assert r.stdout.decode().strip() == ''                                # no JSON line from a failed run


def test_quoted_kernel_averages_come_from_the_csvs():
    """profiles/README.md and DESIGN.md section 6 quote per-kernel averages of the round's profiled runs: the blocks are written
    by tools/profiles_readme.py FROM profiles/r06_bench_*_kernel_stats.csv and must equal what the CSVs give (round 5's prose had
    drifted from its files)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'profiles_readme.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _fake_node(tmp_path, gpu_numa=(0, 0, 0, 0, 1, 1, 1, 1), cpus_per_node=16):
    """A sysfs tree of a two-socket node: KFD nodes 0-1 are the CPUs, 2.. the GPUs (one PCI bus each)."""
    root = tmp_path / 'sysfs'
    for n in range(2):
        d = root / 'sys/class/kfd/kfd/topology/nodes' / str(n)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count %d\nsimd_count 0\nlocation_id 0\ndomain 0\n' % cpus_per_node)
        nd = root / ('sys/devices/system/node/node%d' % n)
        nd.mkdir(parents=True)
        (nd / 'cpulist').write_text('%d-%d\n' % (n * cpus_per_node, (n + 1) * cpus_per_node - 1))
    for g, numa in enumerate(gpu_numa):
        d = root / 'sys/class/kfd/kfd/topology/nodes' / str(2 + g)
        d.mkdir(parents=True)
        bus = 0x10 + 0x10 * g
        (d / 'properties').write_text('cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n' % (bus << 8))
        pd = root / ('sys/bus/pci/devices/0000:%02x:00.0' % bus)
        pd.mkdir(parents=True)
        (pd / 'numa_node').write_text('%d\n' % numa)
    return str(root)


def test_numa_cpu_plan_on_a_two_socket_node(tmp_path):
    """codes/utils/dist_utils.py: every rank lands on its GPU's NUMA node, the ranks of one node get disjoint equal shares,
    visible-device lists are honoured, and anything unknown leaves the affinity alone."""
    from codes.utils import dist_utils as du
    root = _fake_node(tmp_path)
    assert du.gpu_numa_nodes(root, {}) == [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [du.numa_cpu_plan(r, 8, sysfs=root, environ={}, current=set(range(32))) for r in range(8)]
    assert plans[0] == {0, 1, 2, 3} and plans[3] == {12, 13, 14, 15} and plans[4] == {16, 17, 18, 19} and plans[7] == {28, 29, 30, 31}
    assert all(not (plans[a] & plans[b]) for a in range(8) for b in range(a))
    # two ranks only: each gets half of ITS node (both GPUs are on node 0 here)
    assert du.numa_cpu_plan(1, 2, sysfs=root, environ={}, current=set(range(32))) == set(range(8, 16))
    # HIP_VISIBLE_DEVICES remaps local rank 0 to the fifth GPU (node 1)
    env = {'HIP_VISIBLE_DEVICES': '4,5'}
    assert du.gpu_numa_nodes(root, env) == [1, 1]
    assert du.numa_cpu_plan(0, 2, sysfs=root, environ=env, current=set(range(32))) == set(range(16, 24))
    assert du.gpu_numa_nodes(root, {'HIP_VISIBLE_DEVICES': 'GPU-abcdef'}) == []
    # HIP_ and CUDA_VISIBLE_DEVICES are two names of ONE filter on ROCm, not a composition: a launcher that sets both to the
    # same list still means GPUs 4 and 5 (composing them raised IndexError or picked another GPU's node); HIP's wins
    assert du.gpu_numa_nodes(root, {'HIP_VISIBLE_DEVICES': '4,5', 'CUDA_VISIBLE_DEVICES': '4,5'}) == [1, 1]
    assert du.gpu_numa_nodes(root, {'HIP_VISIBLE_DEVICES': '1,0', 'CUDA_VISIBLE_DEVICES': '7,6'}) == [0, 0]
    assert du.gpu_numa_nodes(root, {'ROCR_VISIBLE_DEVICES': '2,3,4,5', 'CUDA_VISIBLE_DEVICES': '2,3'}) == [1, 1]
    # SMT: logical CPUs c and c + 8 of node 0 are the two threads of one core -- a rank gets WHOLE cores, never the sibling
    # threads of another rank's cores
    smt = _fake_node(tmp_path / 'smt')
    for c in range(16):
        d = os.path.join(smt, 'sys/devices/system/cpu/cpu%d/topology' % c)
        os.makedirs(d)
        with open(os.path.join(d, 'thread_siblings_list'), 'w') as f:
            f.write('%d,%d\n' % (c % 8, c % 8 + 8))
    sp = [du.numa_cpu_plan(r, 8, sysfs=smt, environ={}, current=set(range(32))) for r in range(4)]
    assert sp == [{0, 8, 1, 9}, {2, 10, 3, 11}, {4, 12, 5, 13}, {6, 14, 7, 15}]
    # a container's mask that is already inside one node, an unknown NUMA node, no topology: hands off
    assert du.numa_cpu_plan(0, 8, sysfs=root, environ={}, current={0, 1, 2, 3}) is None
    assert du.numa_cpu_plan(0, 8, sysfs=_fake_node(tmp_path / 'b', gpu_numa=(-1,) * 8), environ={}, current=set(range(32))) is None
    assert du.numa_cpu_plan(0, 8, sysfs=str(tmp_path / 'nothing'), environ={}, current=set(range(32))) is None
    assert du.top_layer_spare_cus({}) == 40 and du.top_layer_spare_cus({'NCCL_MAX_NCHANNELS': '16'}) == 24
    # data_parallel_env on a foreign environ never touches this process's affinity
    before = os.sched_getaffinity(0)
    out = du.data_parallel_env({'LOCAL_RANK': '0'})
    assert out == {'GPU_MAX_HW_QUEUES': '3', 'NCCL_MAX_NCHANNELS': '32'} and os.sched_getaffinity(0) == before
