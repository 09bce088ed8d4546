#!/usr/bin/env python
"""Training CLI with the reference's flags and JSON config schema (reference ``train.py:56-112``), driving the
MI355X hot path.  ignite, visdom and tensorboardX are not used: the event wiring of ``train.py:212-401`` is a plain
loop here.  One process per GPU; ``--local`` = single GPU, otherwise torch.distributed (nccl = RCCL).

    python train.py scripts/librispeech-from_scratch.json --train-manifest train.csv --val-manifest val.csv --local
"""
import argparse
import json
import logging
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'aes-lac-2018_amd'))

from codes.ctc import CTCLoss as warp_CTCLoss  # noqa: E402
from codes.decoder import GreedyDecoder  # noqa: E402
from codes.engine import create_evaluator, create_trainer  # noqa: E402
from codes.transforms import BatchSpectrogram  # noqa: E402
from codes.utils import model_utils as mu  # noqa: E402
from codes.utils import training_utils as tu  # noqa: E402
from codes.utils.io_utils import AttrDict, expand_values  # noqa: E402

LOG = logging.getLogger('aes-lac-2018')


def build_parser():
    p = argparse.ArgumentParser(description='DeepSpeech-ish model training')
    p.add_argument('config_file', help='Path to config JSON file')
    p.add_argument('--data-dir', metavar='DIR', default=os.getenv('PT_DATA_DIR', 'data/'))
    p.add_argument('--zipped', action='store_true')
    p.add_argument('--train-manifest', nargs='+', metavar='DIR', required=True)
    p.add_argument('--val-manifest', nargs='+', metavar='DIR', required=True)
    p.add_argument('--num-workers', default=4, type=int)
    p.add_argument('--silent', dest='silent', action='store_true')
    p.add_argument('--checkpoint', dest='checkpoint', action='store_true')
    p.add_argument('--checkpoint-per-batch', default=0, type=int)
    p.add_argument('--visdom', dest='visdom', action='store_true')
    p.add_argument('--tensorboard', dest='tensorboard', action='store_true')
    p.add_argument('--log-params', dest='log_params', action='store_true')
    p.add_argument('--id', default='AES LAC 2018 training')
    p.add_argument('--save-folder', default=os.getenv('PT_OUTPUT_DIR', 'results/'))
    p.add_argument('--continue-from', default='')
    p.add_argument('--no-shuffle', action='store_true')
    p.add_argument('--no-sorta-grad', action='store_true')
    p.add_argument('--local', action='store_true')
    p.add_argument('--init-method', default='env://', type=str)
    p.add_argument('--dist-backend', default='nccl', type=str)
    p.add_argument('--local-rank', '--local_rank', type=int, default=int(os.getenv('LOCAL_RANK', '0')))
    p.add_argument('-v', '--verbose', action='count')
    return p


def main(argv=None):
    if not torch.cuda.is_available():
        raise RuntimeError('Training script requires GPU. :(')
    torch.manual_seed(42)
    torch.cuda.manual_seed_all(42)
    random.seed(42)
    np.random.seed(42)

    args = AttrDict(vars(build_parser().parse_args(argv)))
    if args.zipped or args.visdom or args.tensorboard:
        raise NotImplementedError('--zipped / --visdom / --tensorboard are outside the hot path (SURVEY.md section 2)')
    args.distributed = not args.local
    with open(args.config_file, 'r', encoding='utf8') as f:
        args.config = AttrDict(json.load(f))
        args.config = expand_values(args.config, **args)
    os.makedirs(os.path.join(args.save_folder, args.config.model.name), exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(message)s',
                        handlers=[logging.StreamHandler(),
                                  logging.FileHandler(os.path.join(args.save_folder, args.config.model.name + '.log'))])

    device = torch.device('cuda' if args.local else 'cuda:{}'.format(args.local_rank))
    main_proc = True
    if args.distributed:
        # (see bench.py) three hardware queues serve the main / weight-gradient / all-reduce streams better than the
        # default four; the HIP runtime reads this at its first call, so it goes before set_device
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '3')
        torch.cuda.set_device(device)
        # the recurrence kernels need ~204 of the 256 CUs co-resident: keep RCCL's kernels at <= 32 workgroups
        os.environ.setdefault('NCCL_MAX_NCHANNELS', '32')
        torch.distributed.init_process_group(backend=args.dist_backend, init_method=args.init_method)
        main_proc = torch.distributed.get_rank() == 0

    ckpt = None
    if args.continue_from:
        LOG.info('Loading model from {}'.format(args.continue_from))
        model, ckpt = mu.load_model(args.continue_from, return_ckpt=True)
    else:
        model = tu.get_model(args.config.model)
    finetune = bool(args.config.training.get('finetune', False))
    if finetune:
        model = tu.finetune_model(model, args.config.model)
    model.to(device)

    optimizer = tu.get_optimizer(tu.get_per_params_lr(model, args.config.optimizer), args.config.optimizer)
    scheduler = tu.get_scheduler(optimizer, args.config.scheduler)
    start_epoch, start_iteration = 0, 0
    if ckpt is not None and not finetune:                         # resume (train.py:140-157)
        optimizer.load_state_dict(ckpt['optimizer'])
        if ckpt.get('scheduler'):
            scheduler.load_state_dict(ckpt['scheduler'])
        start_epoch, start_iteration = ckpt['epoch'], ckpt['iteration']

    train_t, val_t, target_t = tu.get_default_transforms(args.data_dir, args.config)
    train_loader, val_loader = tu.get_data_loaders(train_t, val_t, target_t, args)
    criterion = [warp_CTCLoss()]
    decoder = GreedyDecoder(target_t[0].label_encoder)
    frontend = BatchSpectrogram(device=device)
    skip_n = start_iteration % max(1, len(train_loader))
    trainer = create_trainer(model, optimizer, criterion, device, skip_n=skip_n, frontend=frontend,
                             **args.config.training)
    evaluator = create_evaluator(model, None, device, decoder=decoder)

    def eval_loader(loader):
        def gen():
            for wavs, targets, _, sizes in loader:
                inputs, pct = frontend(wavs)
                yield inputs, targets, pct, sizes
        return evaluator.run(gen())

    iteration = start_iteration
    for epoch in range(start_epoch, args.config.training.num_epochs):
        t_epoch = time.time()
        for i, batch in enumerate(train_loader):
            t0 = time.time()
            loss = trainer.update(batch)
            iteration += 1
            if main_proc and not args.silent and loss != 'Skipped':
                LOG.info('Epoch: [{}][{}/{}]\tTime {:.3f}\tData {:.3f}\tLoss {:.4f}'.format(
                    epoch + 1, i + 1, len(train_loader), time.time() - t0, trainer.data_time, loss))
            if main_proc and args.checkpoint_per_batch and iteration % args.checkpoint_per_batch == 0:
                torch.save(mu.make_checkpoint(args, model, optimizer, scheduler, epoch, iteration),
                           os.path.join(args.save_folder, args.config.model.name,
                                        'model_batch-ckpt_{}.pth'.format(iteration)))
        val = eval_loader(val_loader)
        if main_proc:
            LOG.info('Training Summary Epoch: [{}]\tTime taken (s): {:.0f}'.format(epoch + 1, time.time() - t_epoch))
            LOG.info('Validation Summary Epoch: [{}]\tAverage ctcloss {:.3f}\tAverage wer {:.3f}\tAverage cer {:.3f}'
                     .format(epoch + 1, val['ctcloss'], val['wer'], val['cer']))
        scheduler.step()
        if main_proc and args.checkpoint:
            torch.save(mu.make_checkpoint(args, model, optimizer, scheduler, epoch + 1, iteration, val_metrics=val),
                       os.path.join(args.save_folder, args.config.model.name, 'model_ckpt_{}.pth'.format(epoch + 1)))
        if not args.no_shuffle:
            train_loader.batch_sampler.shuffle(epoch)


if __name__ == '__main__':
    main()
