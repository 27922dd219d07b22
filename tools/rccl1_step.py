#!/usr/bin/env python
"""One process, backend 'nccl' (RCCL), world size 1: a few data-parallel adapt steps at 192x640, B = 5 with the bucketed
gradient exchange on the asynchronous tail stream -- for `rocprofv3 --kernel-trace` (tools/measure_round.sh ->
profiles/rNN_rccl1_timeline.txt: which queue the RCCL kernels run on and what they overlap).  Measures no scaling."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
    sys.path.insert(0, str(p))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 2000))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('CLSLAM_ASYNC_TAIL', '1')
os.environ.setdefault('CLSLAM_GRAD_BUCKETS', '3')
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from clslam_hip import synth  # noqa: E402
from predictor_util import make_predictor  # noqa: E402

H, W, B = 192, 640, 5
p = make_predictor(H, W, B)
p.enable_data_parallel(B, 0)
batch = {k: v.cuda() for k, v in synth.make_batch(B, H, W, seed=4).items()}
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    out, losses = p.adapt(None, batch, steps=1)
    float(losses['loss'])
p.synchronize()
torch.cuda.synchronize()
print('replicas in sync:', p.replicas_in_sync(), ' loss', float(losses['loss']))
dist.destroy_process_group()
