"""Checkpoint I/O (reference ``codes/utils/model_utils.py:19-124``, payload of ``train.py:331-340``).

Payload keys: ``args, state_dict, optimizer, scheduler, epoch, iteration, metrics, val_metrics``; the
state dict uses the reference's key names, so its checkpoints load here and vice versa.  Two older formats
are read as well: checkpoints whose ``args.config`` still has the ``network`` / ``transforms`` layout
(``model_utils.py:26-54``) and the ``version == '0.0.1'`` format of the released pre-trained English models
(``model_utils.py:80-124``, ``README.md:260-262``).
"""
import torch

from . import training_utils as tu
from .io_utils import AttrDict

# keys an early exporter saved although the first recurrent layer has no BatchNorm (model_utils.py:104-112)
_LEGACY_STRAY_KEYS = tuple('rnns.0.batch_norm.module.' + leaf
                           for leaf in ('weight', 'bias', 'running_mean', 'running_var'))


def num_of_parameters(model, trainable=False):
    return sum(p.numel() for p in model.parameters() if p.requires_grad or not trainable)


def get_state_dict(model):
    model = model.module if hasattr(model, 'module') else model
    return model.state_dict()


def make_checkpoint(args, model, optimizer, scheduler, epoch, iteration, metrics=None, val_metrics=None):
    return {'args': dict(args), 'state_dict': get_state_dict(model), 'optimizer': optimizer.state_dict(),
            'scheduler': scheduler.state_dict() if scheduler is not None else None, 'epoch': epoch,
            'iteration': iteration, 'metrics': metrics or {}, 'val_metrics': val_metrics or {}}


def upgrade_config(args):
    """``args.config`` in the pre-``model{}`` schema -> the current schema (model_utils.py:26-54)."""
    old = args.config
    net, trn = old.network, old.training
    label_file = old.transforms.label[0].params.labels              # e.g. "{data_dir}/labels.en.json"
    lang = label_file.split('.')[-2]
    params = AttrDict(net.params)
    params.num_classes = tu.NUM_CLASSES[lang]
    args.config = AttrDict({
        'model': {'name': net.name, 'map_fc': net.get('map_fc', None), 'freeze_layers': net.get('freeze_layers', None),
                  'langs': [lang], 'params': params},
        'training': {'num_epochs': trn.num_epochs, 'batch_size': args.batch_size, 'max_norm': trn.max_norm,
                     'augment': old.transforms.train[0].params.augment, 'finetune': args.finetune},
        'optimizer': {'name': 'SGD', 'params': {'lr': trn.learning_rate, 'momentum': trn.momentum, 'nesterov': True},
                      'per_layer_lr': trn.get('per_layer_lr', None)},
        'scheduler': {'name': 'ExponentialLR', 'params': {'gamma': trn.learning_anneal}},
    })
    return args


def load_legacy_model(ckpt):
    """``version == '0.0.1'`` checkpoints (model_utils.py:80-124): geometry in seconds, labels inline.

    Always returns ``(model, transforms, target_transforms)`` like the reference does for this format."""
    from .. import transforms as T
    from ..model import DeepSpeech
    if ckpt.get('version') != '0.0.1':
        raise ValueError('not a version-0.0.1 checkpoint')
    audio = ckpt['audio_conf']
    rate = audio['sample_rate']
    frame_length, hop = int(rate * audio['window_size']), int(rate * audio['window_stride'])
    labels = ckpt['labels']
    model = DeepSpeech(rnn_type=ckpt['rnn_type'], num_classes=len(labels), rnn_hidden_size=ckpt['hidden_size'],
                       num_rnn_layers=ckpt['hidden_layers'], window_size=frame_length,
                       bidirectional=ckpt['bidirectional'], context=ckpt.get('context', 20))
    weights = {k: v for k, v in ckpt['state_dict'].items() if k not in _LEGACY_STRAY_KEYS}
    model.load_state_dict(weights)
    front = T.Compose([T.ToTensor(augment=False, sample_rate=rate),
                       T.ToSpectrogram(frame_length=frame_length, hop=hop, librosa_compat=True)])
    return model, front, T.ToLabel(labels)


def load_model(model_path, num_classes=29, return_transforms=False, data_dir=None, return_ckpt=False):
    ckpt = torch.load(model_path, map_location='cpu', weights_only=False)
    if ckpt.get('version') == '0.0.1':
        return load_legacy_model(ckpt)
    args = AttrDict(ckpt['args'])
    if 'network' in args.config:
        args = upgrade_config(args)
    model = tu.get_model(args.config.model)
    model.load_state_dict(ckpt['state_dict'])
    out = [model]
    if return_transforms:
        out += list(tu.get_default_transforms(data_dir or args.data_dir, args.config))
    if return_ckpt:
        out.append(ckpt)
    return out[0] if len(out) == 1 else tuple(out)
