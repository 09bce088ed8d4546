#!/usr/bin/env python
"""One rank of the 2-rank data-parallel step test (started by tests/test_ddp_gpu.py as a fresh process).

    python tests/ddp_worker.py RANK WORLD PORT OUT.npz [cpu-store]

Both ranks share cuda:0 (the box has one GPU); the process group is ``gloo`` on device tensors.  The parent sets
DS2_GRU_MODE: ``step`` (the launch-per-step kernels) or ``persistent`` (the product's kernels: at this model's width a
launch is 8-24 workgroups, so both ranks' launches fit on the one GPU together).  Rank r != 0 starts from
DIFFERENT weights: the trainer's construction-time broadcast must overwrite them (DDP semantics).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'aes-lac-2018_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('DS2_TEST_HANG_S', '200')), exit=True)   # a hung rank says where
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from codes.engine import Trainer
        from codes.model import DeepSpeech
        from codes.sampler import DistributedBucketingSampler
        from oracle.model import OracleDeepSpeech, seeded_state_dict
        from tests import ddp_common as dc
        model = DeepSpeech(**dc.MODEL_KW)
        model.load_state_dict(seeded_state_dict(OracleDeepSpeech(**dc.MODEL_KW), 7 if rank == 0 else 70 + rank))
        model.to('cuda')
        opt = torch.optim.SGD(model.parameters(), lr=dc.LR, momentum=dc.MOMENTUM, nesterov=True)
        trainer = Trainer(model, opt, device='cuda', max_norm=dc.MAX_NORM)
        assert trainer.distributed and trainer.world == world and trainer._fused
        sampler = DistributedBucketingSampler(list(range(dc.NUM_UTTS)), batch_size=dc.BATCH)
        losses, norms = [], []
        for ids in sampler:
            inputs, targets, pct, sizes = dc.batch_of(ids)
            losses.append(trainer.update((torch.from_numpy(inputs), torch.from_numpy(targets), torch.from_numpy(pct),
                                          torch.from_numpy(sizes))))
            norms.append(trainer.last_grad_norm)
        torch.cuda.synchronize()
        res = {'losses': np.asarray(losses), 'norms': np.asarray(norms), 'overlap': np.int32(trainer.overlap)}
        for i, p in enumerate(model.parameters()):
            res['p%03d' % i] = p.detach().cpu().numpy()
        for k, v in model.state_dict().items():
            if 'running' in k:
                res['buf_' + k] = v.cpu().numpy()
        np.savez(out, **res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
