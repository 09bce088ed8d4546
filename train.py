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
from codes.transforms import BatchSpectrogram, waveform_scale  # noqa: E402
from codes.utils import model_utils as mu  # noqa: E402
from codes.utils import training_utils as tu  # noqa: E402
from codes.utils.dist_utils import data_parallel_env  # noqa: E402
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


def write_metrics_log(path, epoch, train_history, val_history):
    """One record per epoch in the reference's format (train.py:359-374): ``Epoch [N] | Train k v k v | Val k v ...``
    (the reference writes no line break between epochs; one is added here)."""
    with open(path, 'a') as f:
        f.write('Epoch [{}] '.format(epoch))
        for name, history in zip(['Train', 'Val'], [train_history, val_history]):
            f.write('| {} '.format(name))
            for k, v in history.items():
                f.write('{} '.format(k))
                if isinstance(v[-1], float):
                    f.write('{:.3f}'.format(v[-1]))
                elif isinstance(v[-1], (tuple, list)):
                    for i, t_k in enumerate(v[-1]):
                        f.write('{:.3f}{}'.format(t_k, '/' if i < len(v[-1]) - 1 else ''))
                else:
                    f.write('{}'.format(v[-1]))
        f.write('\n')


class BestCheckpoints(object):
    """The reference's ``best_ckpt_handler`` (train.py:223-229): keep the ``n_saved`` best ``model_best-ckpt_<N>.pth`` files
    by validation CER.  ignite keeps the HIGHEST scores and the reference passes the raw CER as the score; the evident
    intent -- keep the lowest CER -- is what this does.  N counts the handler's calls (= epochs run by this process),
    as ignite's file counter does."""

    def __init__(self, folder, n_saved=5):
        self.folder, self.n_saved, self.calls, self.saved = folder, n_saved, 0, []

    def __call__(self, cer, payload):
        self.calls += 1
        if len(self.saved) >= self.n_saved and cer >= max(s for s, _ in self.saved):
            return None
        path = os.path.join(self.folder, 'model_best-ckpt_{}.pth'.format(self.calls))
        torch.save(payload, path)
        self.saved.append((cer, path))
        self.saved.sort(key=lambda sp: sp[0])
        while len(self.saved) > self.n_saved:
            _, worst = self.saved.pop()
            if os.path.exists(worst):
                os.remove(worst)
        return path


def main(argv=None):
    args = AttrDict(vars(build_parser().parse_args(argv)))
    args.distributed = not args.local
    if args.distributed:
        # BEFORE the first torch.cuda call (is_available() already initialises the HIP runtime, which reads its flags
        # once): hardware-queue count and RCCL channel cap -- codes/utils/dist_utils.py, shared with bench.py
        dp_env = data_parallel_env()
    if not torch.cuda.is_available():
        raise RuntimeError('Training script requires GPU. :(')
    torch.manual_seed(42)
    torch.cuda.manual_seed_all(42)
    random.seed(42)
    np.random.seed(42)

    if args.zipped or args.visdom or args.tensorboard:
        raise NotImplementedError('--zipped / --visdom / --tensorboard are outside the hot path (SURVEY.md section 2)')
    with open(args.config_file, 'r', encoding='utf8') as f:
        args.config = AttrDict(json.load(f))
        args.config = expand_values(args.config, **args)
    out_dir = os.path.join(args.save_folder, args.config.model.name)
    os.makedirs(out_dir, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(message)s',
                        handlers=[logging.StreamHandler(),
                                  logging.FileHandler(os.path.join(args.save_folder, args.config.model.name + '.log'))])

    if args.distributed:
        LOG.info('data-parallel environment: {}'.format(dp_env))
    device = torch.device('cuda' if args.local else 'cuda:{}'.format(args.local_rank))
    main_proc = True
    if args.distributed:
        torch.cuda.set_device(device)
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
        model = tu.finetune_model(model, args.config.model, data_dir=args.data_dir)
    model.to(device)

    optimizer = tu.get_optimizer(tu.get_per_params_lr(model, args.config.optimizer), args.config.optimizer)
    scheduler = tu.get_scheduler(optimizer, args.config.scheduler)
    start_epoch, start_iteration = 0, 0
    train_history, val_history = {}, {}
    if ckpt is not None and not finetune:                         # resume (train.py:140-157)
        optimizer.load_state_dict(ckpt['optimizer'])
        if ckpt.get('scheduler'):
            scheduler.load_state_dict(ckpt['scheduler'])
        start_epoch, start_iteration = ckpt['epoch'], ckpt['iteration']
        train_history, val_history = dict(ckpt.get('metrics') or {}), dict(ckpt.get('val_metrics') or {})
        LOG.info('Start epoch: {}. Start iteration {}'.format(start_epoch, start_iteration))

    train_t, val_t, target_t = tu.get_default_transforms(args.data_dir, args.config)
    train_loader, val_loader = tu.get_data_loaders(train_t, val_t, target_t, args)
    criterion = [warp_CTCLoss()]
    decoder = GreedyDecoder(target_t[0].label_encoder)
    frontend = BatchSpectrogram(device=device, scale=waveform_scale(train_t))
    for ld in (train_loader, val_loader):            # decode + augmentation + STFT of the NEXT bin run on the prefetch stream
        if hasattr(ld, 'frontend'):
            ld.frontend = frontend
    steps_per_epoch = max(1, len(train_loader))
    skip_n = int(start_iteration % steps_per_epoch)               # train.py:198-200
    trainer = create_trainer(model, optimizer, criterion, device, skip_n=skip_n, frontend=frontend,
                             **args.config.training)
    evaluator = create_evaluator(model, None, device, decoder=decoder)
    best = BestCheckpoints(out_dir)

    def eval_loader(loader):
        def gen():
            for wavs, targets, pct, sizes in loader:
                if not isinstance(wavs, torch.Tensor):               # (a prefetcher with the frontend attached yields tensors)
                    wavs, pct = frontend(wavs)
                yield wavs, targets, pct, sizes
        return evaluator.run(gen())

    def payload(epoch, iteration):
        return mu.make_checkpoint(args, model, optimizer, scheduler, epoch, iteration, metrics=train_history,
                                  val_metrics=val_history)

    # SortaGrad: the first epoch runs the length-sorted bins in order; a resumed run (epoch >= 1) or --no-sorta-grad
    # starts from shuffled bins (train.py:376-379)
    if (not args.no_shuffle and start_epoch != 0) or args.no_sorta_grad:
        LOG.info('Shuffling batches for the following epochs')
        train_loader.batch_sampler.shuffle(start_epoch)

    # the reference resets ignite's counter to start_epoch * len(loader) on resume (train.py:391-394) and ignite then
    # counts every batch, the skip_n 'Skipped' ones included -- so after them it is back at the checkpoint's iteration
    iteration = start_epoch * steps_per_epoch
    for epoch in range(start_epoch, args.config.training.num_epochs):
        t_epoch = time.time()
        def log_step(rec):
            # the readback of a step is deferred until the next one has been enqueued (Trainer.update(defer=True)):
            # its line is printed one iteration late, same content
            i, t0, data_time, loss = rec
            loss = loss.result() if hasattr(loss, 'result') else loss
            if main_proc and not args.silent and loss != 'Skipped':
                LOG.info('Epoch: [{}][{}/{}]\tTime {:.3f}\tData {:.3f}\tLoss {:.4f}'.format(
                    epoch + 1, i + 1, len(train_loader), time.time() - t0, data_time, loss))

        prev = None
        for i, batch in enumerate(train_loader):
            t0 = time.time()
            loss = trainer.update(batch, defer=True)
            iteration += 1
            if prev is not None:
                log_step(prev)
            prev = (i, t0, trainer.data_time, loss)
            if main_proc and args.checkpoint_per_batch and iteration % args.checkpoint_per_batch == 0:
                trainer.flush()      # resolve this step's deferred readback first: a step whose recurrence timed out
                                     # raises HERE, before its weights can be written to a checkpoint
                torch.save(payload(epoch, iteration), os.path.join(out_dir, 'model_batch-ckpt_{}.pth'.format(iteration)))
        trainer.flush()
        if prev is not None:
            log_step(prev)
        LOG.info('Training epoch [{}] took {:.0f} s'.format(epoch + 1, time.time() - t_epoch))
        # epoch end, in the reference's handler order (train.py:272-374): train metrics, validation metrics, LR anneal,
        # epoch checkpoint, best-CER checkpoint, metrics-log
        train_m = eval_loader(train_loader)
        val_m = eval_loader(val_loader)
        for hist, m in ((train_history, train_m), (val_history, val_m)):
            for k, v in m.items():
                hist.setdefault(k, []).append(v)
        if main_proc:
            fmt = '\t'.join('Average {} {:.3f}'.format(k, v) for k, v in train_m.items())
            LOG.info('Training Summary Epoch: [{}]\t{}'.format(epoch + 1, fmt))
            LOG.info('Validation Summary Epoch: [{}]\tAverage ctcloss {:.3f}\tAverage wer {:.3f}\tAverage cer {:.3f}'
                     .format(epoch + 1, val_m['ctcloss'], val_m['wer'], val_m['cer']))
        old_lr = optimizer.param_groups[0]['lr']
        scheduler.step()
        LOG.info('Annealing learning rate from {:.5g} to {:5g}.'.format(old_lr, optimizer.param_groups[0]['lr']))
        if main_proc and args.checkpoint:
            torch.save(payload(epoch + 1, iteration), os.path.join(out_dir, 'model_ckpt_{}.pth'.format(epoch + 1)))
        if main_proc:
            best(val_m['cer'], payload(epoch + 1, iteration))
            write_metrics_log(os.path.join(out_dir, 'metrics-log'), epoch + 1, train_history, val_history)
        if not args.no_shuffle:
            LOG.info('Shuffling batches...')
            train_loader.batch_sampler.shuffle(epoch + 1)


if __name__ == '__main__':
    main()
