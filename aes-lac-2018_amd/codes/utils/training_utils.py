"""Factories driven by the JSON config (reference ``codes/utils/training_utils.py``).

Config schema (unchanged): ``model{name, langs, freeze_layers, map_fc, params}``,
``training{num_epochs, batch_size, max_norm, augment, finetune}``, ``optimizer{name, params, per_layer_lr}``,
``scheduler{name, params}``.  Broken branches of the reference are implemented to their evident intent
(SURVEY.md section 4): ``langs[0]`` is the fine-tune target language, the new FC layer's *weight* is
normally initialised.
"""
import json
import logging
import os

import torch

from .. import transforms
from ..data import AudioDataLoader, AudioDataset
from ..model import DeepSpeech, _BatchNormParams, _LinearParams
from ..sampler import BucketingSampler, DistributedBucketingSampler

LOG = logging.getLogger('aes-lac-2018')
NUM_CLASSES = {'pt_BR': 43, 'en': 29}


def get_default_transforms(data_dir, config, gpu_frontend=True):
    """Waveform loader (+ per-utterance spectrogram when ``gpu_frontend`` is False) and one ToLabel per language
    (training_utils.py:18-34).  With ``gpu_frontend`` the spectrogram runs batched on the device after collate."""
    augment = bool(config.training.get('augment', False))       # tempo + gain on the training set only
    # gpu_frontend: workers hand on int16 clips + the drawn (tempo, gain); decode, WSOLA, gain and the spectrogram all run
    # on the device after collate.  Otherwise the reference's per-utterance contract (each transform returns a tensor).
    tail = [] if gpu_frontend else [transforms.ToSpectrogram(librosa_compat=True)]
    scale = audio_scale(config)
    train_t = transforms.Compose([transforms.ToTensor(augment=augment, defer=gpu_frontend, scale=scale)] + tail)
    val_t = transforms.Compose([transforms.ToTensor(augment=False, defer=gpu_frontend, scale=scale)] + tail)
    target_t = [transforms.ToLabel(os.path.join(data_dir, 'labels.{}.json'.format(lang)), lang=lang,
                                   remove_accents=(lang != 'pt_BR')) for lang in config.model.langs]
    return train_t, val_t, target_t


def audio_scale(config):
    """``training.audio_scale`` of the JSON config (an addition to the reference's schema; saved with the checkpoint's
    ``args`` so ``test.py`` decodes the way the model was trained): 'unit' (default), 'int32' or a number -- the amplitude
    contract of the waveform loader (``transforms.ToTensor``)."""
    training = config.get('training', {}) if hasattr(config, 'get') else {}
    return (training or {}).get('audio_scale', None)


def get_model(model_dict):
    if isinstance(model_dict.langs, (tuple, set, list)) and len(model_dict.langs) > 1:
        raise NotImplementedError('multi-task models are out of scope (SURVEY.md section 2 row 1)')
    params = dict(model_dict.get('params', {}) or {})
    # The drop-in covers ONE frontend geometry: 320-sample windows (20 ms at 16 kHz, 161 frequency bins) -- what every shipped
    # config and the reference's released checkpoints use.  The reference derives the model's input width for any window
    # (codes/model.py:124,148-151); the HIP conv, BatchNorm-layout and STFT kernels are specialised for 161 bins, so a config
    # that asks for another one is refused HERE, when it is loaded, with the reason -- not later inside the constructor.
    if int(params.get('window_size', 320)) != 320:
        raise ValueError('model.params.window_size = %r: this MI355X path implements window_size = 320 only (161 frequency '
                         'bins; conv / BatchNorm / STFT kernels are specialised for it) -- see README.md "What the drop-in '
                         'does not cover"' % (params['window_size'],))
    params.setdefault('num_classes', NUM_CLASSES[model_dict.langs[0]])
    model_dict['params'] = params
    return DeepSpeech(**params)


def _freeze_layers(model, freeze_layers):
    if freeze_layers is None:
        return model
    if isinstance(freeze_layers, str):
        freeze_layers = [freeze_layers]
    count = 0
    for name in freeze_layers:
        target = model if name == 'all' else getattr(model, name)
        if isinstance(target, torch.nn.Module):
            for m in target.modules():
                if isinstance(m, _BatchNormParams):
                    # the reference puts the BatchNorm of a frozen LAYER (not of 'all', whose parameters() is neither a
                    # Tensor nor a Module there: :66-67,72-74) in eval mode (:52-54,73) -- until the trainer's
                    # ``model.train()`` at the top of the first step undoes it (codes/engine.py:51; Trainer.update)
                    if name != 'all':
                        m.eval()
            params = target.parameters()
        else:
            params = [target]
        for p in params:
            count += p.numel()
            p.requires_grad = False
    LOG.info('\tFreezed {} parameters'.format(count))
    return model


def _resolve(path, data_dir):
    """A relative path of the config that does not exist from the working directory is looked up under ``--data-dir``
    (the scripts/*.json of the reference name ``map_en-pt_BR.json`` bare; the file lives in ``data/``)."""
    if os.path.isabs(path) or os.path.exists(path) or not data_dir:
        return path
    return os.path.join(data_dir, path)


def finetune_model(model, obj, data_dir=None):
    """Freeze layers and, when the alphabet changes, swap the last FC (training_utils.py:87-122)."""
    freeze_layers = obj.get('freeze_layers', None)
    lang = obj['langs'][0] if 'langs' in obj else obj['lang']
    num_classes = NUM_CLASSES[lang]
    map_fc = obj.get('map_fc', None)
    model = _freeze_layers(model, freeze_layers)
    head = model.fc[0].module
    old = head[1]
    if old.out_features != num_classes or (freeze_layers and freeze_layers[0] == 'all'):
        LOG.info('\tChanging the last FC layer')
        new = _LinearParams(old.in_features, num_classes).to(old.weight.device)
        with torch.no_grad():
            torch.nn.init.normal_(new.weight, 0, 0.01)
            if map_fc is not None:
                LOG.info('\t Mapping FC weights')
                pairs = json.load(open(_resolve(map_fc, data_dir)))
                old_idx, new_idx = zip(*pairs)
                new.weight.index_copy_(0, torch.tensor(new_idx, device=new.weight.device),
                                       old.weight.detach().index_select(0, torch.tensor(old_idx,
                                                                                        device=old.weight.device)))
        head[1] = new
        model._num_classes = num_classes
        model._flat_p = None                 # parameters changed: re-pack lazily
    return model


def get_optimizer(params, obj):
    return getattr(torch.optim, obj.get('name', 'SGD'))(params, **obj.params)


def get_scheduler(optimizer, obj):
    return getattr(torch.optim.lr_scheduler, obj.get('name', 'ExponentialLR'))(optimizer, **obj.params)


def get_per_params_lr(model, obj):
    """Per-layer learning-rate groups (training_utils.py:133-159): [[name, lr], ..., ['base']]."""
    per_layer_lr = obj.get('per_layer_lr', None)
    if per_layer_lr is None:
        return model.parameters()
    groups, taken, has_base = [], set(), False
    for conf in per_layer_lr:
        if conf[0] == 'base':
            has_base = True
            continue
        ps = list(getattr(model, conf[0]).parameters())
        g = {'params': ps}
        if len(conf) > 1:
            g['lr'] = conf[1]
        groups.append(g)
        taken.update(id(p) for p in ps)
    if has_base:
        groups.append({'params': [p for p in model.parameters() if id(p) not in taken]})
    return groups


def _is_deferred(transform):
    """True if no stage of ``transform`` is a waveform loader that decodes on the device per utterance."""
    stages = getattr(transform, 'transforms', [transform])
    return all(getattr(t, 'defer', True) for t in stages if isinstance(t, transforms.ToTensor))


def get_data_loaders(train_transforms, val_transforms, target_transforms, args, raw_audio=True):
    if not isinstance(target_transforms, (list, tuple)):
        target_transforms = [target_transforms]
    if len(target_transforms) != 1:
        raise NotImplementedError('multi-task data loading is out of scope')
    train_set = AudioDataset(args.data_dir, args.train_manifest[0], train_transforms, target_transforms[0])
    val_set = AudioDataset(args.data_dir, args.val_manifest[0], val_transforms, target_transforms[0])
    bsz = args.config.training.batch_size
    if args.distributed:
        sampler = DistributedBucketingSampler(train_set, batch_size=bsz)
    else:
        sampler = BucketingSampler(train_set, batch_size=bsz)
    pin = bool(raw_audio) and torch.cuda.is_available()     # page-locked int16 batches: asynchronous uploads one bin ahead
    workers = args.num_workers
    if workers > 0 and not all(_is_deferred(t) for t in (train_transforms, val_transforms)):
        # a per-utterance (non-deferred) waveform transform runs device kernels: it cannot live in a forked worker
        LOG.warning('the waveform transforms decode on the GPU per utterance (defer=False): loading with num_workers=0')
        workers = 0
    train_loader = AudioDataLoader(train_set, num_workers=workers, batch_sampler=sampler, raw_audio=raw_audio,
                                   pin_memory=pin)
    val_loader = AudioDataLoader(val_set, batch_size=bsz, num_workers=workers, raw_audio=raw_audio,
                                 pin_memory=pin)
    if raw_audio and torch.cuda.is_available():
        from ..data import DevicePrefetcher
        train_loader, val_loader = DevicePrefetcher(train_loader), DevicePrefetcher(val_loader)
    return train_loader, val_loader
