"""SURVEY.md 8(f) rank 3 -- the asynchronous predict/adapt mode (clslam_hip.async_mode): an inference replica fed by a
periodic broadcast of the trainers' weight arena.  gloo, world size 2 (rank 0 inference, rank 1 trainer), kernels
through the CPU emulator.  Checks: the replica never uses weights older than the stated bound, every installed
arena is BITWISE the trainer's snapshot of that frame, predictions made with installed weights equal what a single
process holding those weights predicts, and between installs the replica's weights do not move."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
H, W, B, FRAMES, EVERY = 64, 64, 2, 3, 1


def _worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    torch.set_num_threads(2)
    import torch.distributed as dist
    from clslam_hip import synth
    from clslam_hip.async_mode import AsyncAdaptation
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_predictor(H, W, B if rank == 1 else 1)
    am = AsyncAdaptation(p, sync_every=EVERY)
    assert am.role == ('inference' if rank == 0 else 'trainer')
    log = []
    for f in range(FRAMES):
        online = synth.make_batch(1, H, W, seed=60 + f)
        if rank == 1:
            replay = synth.make_batch(B - 1, H, W, seed=80 + f)
            training = {k: torch.cat([online[k], replay[k]]) for k in online}
            p.set_tie_break_noise(synth.make_noise(B, H, W, seed=f))
            out, losses = am.step(f, online, training)
            rec = {'frame': f, 'loss': float(losses['loss'])}
            if (f + 1) % EVERY == 0:
                p.engine.wait_training()
                rec['snapshot'] = p.engine.w.clone()
        else:
            p.set_tie_break_noise(synth.make_noise(1, H, W, seed=100 + f))
            am.keep_used_weights = True
            out, _ = am.step(f, online)
            rec = {'frame': f, 'weights_frame': am.used_weights_frame, 'installs': am.installs, 'w_used': am.used_weights,
                   'depth': out['depth', 0].clone(), 'T': out['cam_T_cam', 0, 1].clone()}
        log.append(rec)
    am.flush()
    final = {'w': p.engine.w.clone(), 'weights_frame': am.weights_frame, 'installs': am.installs, 'lag_bound': am.lag_bound_frames}
    torch.save({'log': log, 'final': final}, Path(out_dir) / f'rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_inference_replica_follows_the_trainer(tmp_path):
    port = 29500 + (os.getpid() % 2000) + 7
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    inf = torch.load(tmp_path / 'rank0.pt')
    trn = torch.load(tmp_path / 'rank1.pt')
    snaps = {r['frame']: r['snapshot'] for r in trn['log'] if 'snapshot' in r}
    assert sorted(snaps) == [0, 1, 2]
    assert all(torch.isfinite(torch.tensor(r['loss'])) for r in trn['log'])
    assert not torch.equal(snaps[0], snaps[2])                       # the trainer really moves
    # after flush() the replica holds the trainer's LAST snapshot, bit for bit
    assert inf['final']['weights_frame'] == 2 and inf['final']['installs'] == 3
    assert torch.equal(inf['final']['w'], snaps[2])
    sys.path.insert(0, str(ROOT / 'tests'))
    for rec in inf['log']:
        f, wf = rec['frame'], rec['weights_frame']
        assert f - wf <= inf['final']['lag_bound']                   # bounded staleness (wf = -1: the initial weights)
        if wf >= 0:
            assert wf in snaps and wf < f and torch.equal(rec['w_used'], snaps[wf])      # exactly a snapshot, never a mix
    # the weights in use only change at an install
    used = [r['weights_frame'] for r in inf['log']]
    assert used == sorted(used) and used[0] == -1
    # a single process holding snapshot wf predicts the same depth / pose for that frame (same kernels: bitwise)
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    rec = next(r for r in reversed(inf['log']) if r['weights_frame'] >= 0)
    p = make_predictor(H, W, 1)
    p.engine.install_weights(snaps[rec['weights_frame']])
    p.set_tie_break_noise(synth.make_noise(1, H, W, seed=100 + rec['frame']))
    out, _ = p.adapt(synth.make_batch(1, H, W, seed=60 + rec['frame']), None)
    assert torch.equal(out['depth', 0], rec['depth']) and torch.equal(out['cam_T_cam', 0, 1], rec['T'])


def _abort_worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    torch.set_num_threads(2)
    import torch.distributed as dist
    from clslam_hip import synth
    from clslam_hip.async_mode import AsyncAdaptation
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_predictor(H, W, 1)
    am = AsyncAdaptation(p, sync_every=1, transfer_timeout_s=300.0)
    what = 'finished'
    try:
        for f in range(2):
            online = synth.make_batch(1, H, W, seed=60 + f)
            if rank == 1 and f == 1:            # the trainer's second step fails like dpp.py:1115-1118
                online['relative_distance', 0] = torch.full_like(online['relative_distance', 0], float('nan'))
            am.step(f, online, online if rank == 1 else None)
        am.flush()
    except RuntimeError as e:
        what = str(e)
    (Path(out_dir) / f'rank{rank}.txt').write_text(what)
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_failed_trainer_releases_the_inference_replica(tmp_path):
    """A trainer whose step raises (NaN loss) posts an abort marker instead of its snapshot: the inference replica, which
    would otherwise wait for that broadcast forever, raises too."""
    port = 29500 + (os.getpid() % 2000) + 11
    mp.start_processes(_abort_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    assert 'NaN loss' in (tmp_path / 'rank1.txt').read_text()
    assert 'training replica failed' in (tmp_path / 'rank0.txt').read_text()


def _two_trainer_abort_worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='2')
    torch.set_num_threads(1)
    import time
    import torch.distributed as dist
    from clslam_hip import synth
    from clslam_hip.async_mode import AsyncAdaptation
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    trainers = dist.new_group([1, 2])               # (every rank creates every group)
    p = make_predictor(H, W, 1)
    am = AsyncAdaptation(p, sync_every=1, trainer_group=trainers, trainer_global_batch=2, trainer_shard_offset=max(rank - 1, 0),
                         transfer_timeout_s=300.0)
    what, t0 = 'finished', time.monotonic()
    try:
        for f in range(2):
            online = synth.make_batch(1, H, W, seed=60 + f)
            training = None
            if rank >= 1:
                training = {k: v[rank - 1:rank].clone() for k, v in synth.make_batch(2, H, W, seed=80 + f).items()}
                if rank == 2 and f == 1:            # ONE of the two trainers gets a malformed minibatch
                    del training['rgb', 1, 0]
            am.step(f, online, training)
        am.flush()
    except Exception as e:          # noqa: BLE001
        what = f'{type(e).__name__}: {e}'
    (Path(out_dir) / f'rank{rank}.txt').write_text(f'{time.monotonic() - t0:.1f} s | {what}')
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_one_failed_trainer_of_two_is_agreed_on(tmp_path):
    """two training replicas, one fails before its step's exchange.  The failing trainer completes the
    step's collectives with a status word (DepthPosePrediction._dp_abort_step), its peer raises DataParallelPeerFailure at the
    same step, both post the abort marker and the inference replica is released -- nobody waits for a communicator timeout."""
    port = 29500 + (os.getpid() % 2000) + 17
    mp.start_processes(_two_trainer_abort_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True, start_method='spawn')
    texts = [(tmp_path / f'rank{r}.txt').read_text() for r in range(3)]
    assert 'training replica failed' in texts[0], texts
    assert 'DataParallelPeerFailure' in texts[1], texts
    assert 'KeyError' in texts[2], texts
    assert all(float(t.split(' s |')[0]) < 240 for t in texts), texts
