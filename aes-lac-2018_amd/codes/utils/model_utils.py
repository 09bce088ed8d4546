"""Checkpoint I/O (reference ``codes/utils/model_utils.py:19-77``, payload of ``train.py:331-340``).

Payload keys: ``args, state_dict, optimizer, scheduler, epoch, iteration, metrics, val_metrics``; the
state dict uses the reference's key names, so its checkpoints load here and vice versa.
"""
import torch

from . import training_utils as tu
from .io_utils import AttrDict


def num_of_parameters(model, trainable=False):
    return sum(p.numel() for p in model.parameters() if p.requires_grad or not trainable)


def get_state_dict(model):
    model = model.module if hasattr(model, 'module') else model
    return model.state_dict()


def make_checkpoint(args, model, optimizer, scheduler, epoch, iteration, metrics=None, val_metrics=None):
    return {'args': dict(args), 'state_dict': get_state_dict(model), 'optimizer': optimizer.state_dict(),
            'scheduler': scheduler.state_dict() if scheduler is not None else None, 'epoch': epoch,
            'iteration': iteration, 'metrics': metrics or {}, 'val_metrics': val_metrics or {}}


def load_model(model_path, num_classes=29, return_transforms=False, data_dir=None, return_ckpt=False):
    ckpt = torch.load(model_path, map_location='cpu', weights_only=False)
    args = AttrDict(ckpt['args'])
    model = tu.get_model(args.config.model)
    model.load_state_dict(ckpt['state_dict'])
    out = [model]
    if return_transforms:
        out += list(tu.get_default_transforms(data_dir or args.data_dir, args.config))
    if return_ckpt:
        out.append(ckpt)
    return out[0] if len(out) == 1 else tuple(out)
