"""Data-parallel replay minibatch: B triplets sharded over R ranks == single rank, over two steps of
one adapt() call (the second one on the frozen-feature reuse path),
incl. the sample-0 smoothness behaviour living on rank 0.  Runs on CPU: gloo, world_size 2, the
kernels through the emulator build (the GPU path differs only by backend 'nccl' = RCCL)."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
H, W, B = 64, 64, 3


def _worker(rank, world, port, counts, out_dir, steps, streamk, backend='emu'):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    if not streamk:
        os.environ['CLSLAM_NO_STREAMK'] = '1'
    torch.set_num_threads(2)
    import torch.distributed as dist
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    if backend == 'hip':             # one process per GPU, RCCL over xGMI
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(rank)
        use_backend('hip')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        use_backend('emu')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    off = sum(counts[:rank])
    p = make_predictor(H, W, max(counts[rank], 1))
    p.enable_data_parallel(B, off)
    full = synth.make_batch(B, H, W, seed=4)
    noise = synth.make_noise(B, H, W, seed=8)
    p.set_tie_break_noise({s: v[off:off + counts[rank]] for s, v in noise.items()})
    batch = {k: v[off:off + counts[rank]].clone() for k, v in full.items()}
    out, losses = p.adapt(None, batch, steps=steps)  # second step: frozen-feature reuse under data parallelism
    everything = p.gather_outputs(out)               # uneven shards (2 + 1)
    in_sync = p.replicas_in_sync()
    # forward-only calls are local (slam.py:178 / predict() on the rank that holds the online frame): no
    # collective, single-process weights.  Only rank 0 calls; a hidden all_reduce would hang or mis-pair here.
    solo = None
    if rank == 0:
        p.set_tie_break_noise({s: v[:1] for s, v in noise.items()})
        _, solo_l = p.adapt({k: v[:1].clone() for k, v in full.items()}, None)
        solo = {k: v.clone() for k, v in solo_l.items()}
    if rank == 1:                                    # a single flipped mantissa bit on one rank must be noticed
        p.engine.w.view(torch.int32)[12345] ^= 1
    diverged_seen = not p.replicas_in_sync()
    if rank == 1:
        p.engine.w.view(torch.int32)[12345] ^= 1
    cpu = lambda t: t.detach().cpu() if isinstance(t, torch.Tensor) else t   # noqa: E731
    torch.save({'in_sync': in_sync, 'diverged_seen': diverged_seen, 'full_depth': cpu(everything['depth', 0]), 'full_T': cpu(everything['cam_T_cam', 0, -1]),
                'g': cpu(p.engine.g), 'w': cpu(p.engine.w), 'loss': {k: cpu(v).clone() for k, v in losses.items()},
                'T': cpu(out['cam_T_cam', 0, 1]), 'solo': None if solo is None else {k: cpu(v) for k, v in solo.items()}},
               Path(out_dir) / f'rank{rank}.pt')
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('steps,streamk', [(2, False), (1, True)])
def test_two_ranks_equal_single_rank(tmp_path, monkeypatch, steps, streamk):
    """(2, False): two optimizer steps with the batch-invariant tiled convs only -- a sample's activations do not depend
    on the shard it sits in, so the two-rank run equals the single-rank run to summation order even after an update.
    (1, True): the stream-K convs (conv_sk.hip) cut their reduction where the launch's unit count says, i.e. per shard
    size: results differ in the last bits between shardings, so only the first step is compared tightly."""
    if not streamk:
        monkeypatch.setenv('CLSLAM_NO_STREAMK', '1')
    sys.path.insert(0, str(ROOT / 'tests'))
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    p = make_predictor(H, W, B)
    p.set_tie_break_noise(synth.make_noise(B, H, W, seed=8))
    full = synth.make_batch(B, H, W, seed=4)
    out, losses = p.adapt(None, {k: v.clone() for k, v in full.items()}, steps=steps)
    counts = [2, 1]
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, counts, str(tmp_path), steps, streamk), nprocs=2, join=True, start_method='spawn')
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    # identical all-reduced gradients and weights on both ranks
    assert torch.equal(r0['g'], r1['g']) and torch.equal(r0['w'], r1['w'])
    # equal to the single-rank run up to summation order
    g = p.engine.g
    # (second step: the weights already differ by the summation order of step 1, so does this gradient)
    # (stream-K: last-bit differences of the forward move a bilinear / min kink for a pixel or two -- DESIGN.md section 2)
    assert float((r0['g'] - g).abs().max() / g.abs().max()) < (1e-3 if streamk else 1e-4)
    dw = (r0['w'] - p.engine.w).abs()
    if streamk:      # Adam's first update is lr * sign(g): a near-zero gradient entry may land on the other side (2 lr)
        assert float(dw.max()) < 2.5e-4 and float((dw > 0.3e-4).float().mean()) < 2e-3
    else:
        assert float(dw.max()) < 0.3e-4          # well below one lr-sized flip
    for k, v in losses.items():
        assert abs(float(r0['loss'][k]) - float(v)) <= 2e-5 * max(abs(float(v)), 1e-3), k   # losses of the last step
    assert torch.allclose(r0['T'], out['cam_T_cam', 0, 1][:2], atol=1e-6)
    # rank 0's forward-only call on the online sample == a single process doing the same (after the same 2 steps)
    p.set_tie_break_noise({s: v[:1] for s, v in synth.make_noise(B, H, W, seed=8).items()})
    _, solo = p.adapt({k: v[:1].clone() for k, v in full.items()}, None)
    for k, v in solo.items():
        assert abs(float(r0['solo'][k]) - float(v)) <= (1e-3 if streamk else 1e-4) * max(abs(float(v)), 1e-3), k
    # the explicit all-gather helper: the single-process full-batch dict, identical on both ranks
    for r in (r0, r1):
        assert r['in_sync'] and r['diverged_seen']
        assert r['full_depth'].shape == out['depth', 0].shape
        assert torch.equal(r['full_depth'], r0['full_depth'])
        assert torch.allclose(r['full_depth'], out['depth', 0], rtol=1e-5, atol=0)
        assert torch.allclose(r['full_T'], out['cam_T_cam', 0, -1], atol=1e-6)


@pytest.mark.timeout(900)
def test_a_rank_without_samples_takes_part_with_zeros(tmp_path):
    """While the replay buffer is still filling up the global minibatch is smaller than the number of ranks (slam.py:150-160:
    `training_data` is the online sample alone at first): a rank whose shard is EMPTY still posts every collective (zeros) and
    applies the same optimizer step.  Shards 3 + 0 over two steps: both ranks end bit-identical, and rank 0 -- which holds the
    whole minibatch -- is bitwise the single process (x + 0 == x)."""
    sys.path.insert(0, str(ROOT / 'tests'))
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    steps = 2
    p = make_predictor(H, W, B)
    p.set_tie_break_noise(synth.make_noise(B, H, W, seed=8))
    full = synth.make_batch(B, H, W, seed=4)
    out, losses = p.adapt(None, {k: v.clone() for k, v in full.items()}, steps=steps)
    port = 29500 + (os.getpid() % 2000) + 7
    mp.start_processes(_worker, args=(2, port, [3, 0], str(tmp_path), steps, True), nprocs=2, join=True, start_method='spawn')
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(r0['g'], r1['g']) and torch.equal(r0['w'], r1['w']) and r0['in_sync'] and r1['in_sync']
    assert torch.equal(r0['w'], p.engine.w) and torch.equal(r0['g'], p.engine.g)
    for k, v in losses.items():
        assert float(r0['loss'][k]) == float(v) == float(r1['loss'][k]), k
    assert r1['T'].shape == (0, 4, 4) and torch.equal(r1['full_depth'], out['depth', 0]) and torch.equal(r0['full_depth'], out['depth', 0])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_nccl_equal_single_rank(tmp_path):
    """The same check on hardware whenever two GPUs are visible: one process per GPU, backend 'nccl' (= RCCL over
    xGMI), the asynchronous tail (all-reduce + Adam on their own stream) and replicas_in_sync() on device tensors."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two MI355X (the build has only been given single-GPU boxes)')
    sys.path.insert(0, str(ROOT / 'tests'))
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('hip')
    steps = 2
    p = make_predictor(H, W, B)
    p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=8).items()})
    full = synth.make_batch(B, H, W, seed=4)
    out, losses = p.adapt(None, {k: v.clone() for k, v in full.items()}, steps=steps)
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, [2, 1], str(tmp_path), steps, True, 'hip'), nprocs=2, join=True, start_method='spawn')
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(r0['g'], r1['g']) and torch.equal(r0['w'], r1['w'])       # identical replicas after RCCL + Adam
    assert r0['in_sync'] and r1['in_sync'] and r0['diverged_seen'] and r1['diverged_seen']
    g = p.engine.g.cpu()
    assert float((r0['g'] - g).abs().max() / g.abs().max()) < 5e-2               # step 2 on a piecewise-smooth loss
    assert float((r0['w'] - p.engine.w.cpu()).abs().max()) < 4.5e-4              # at most a couple of lr-sized flips
    assert torch.allclose(r0['full_depth'], out['depth', 0].cpu(), rtol=2e-2, atol=0)


def _stale_worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    torch.set_num_threads(2)
    import torch.distributed as dist
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_predictor(H, W, 1)
    p.enable_data_parallel(3, rank)          # the ranks hold 1 + 1 samples: a global batch of 3 is stale
    batch = {k: v[rank:rank + 1].clone() for k, v in synth.make_batch(2, H, W, seed=4).items()}
    w0 = p.engine.w.clone()
    msg = ''
    try:
        p.adapt(None, batch)
    except RuntimeError as e:
        msg = str(e)
    torch.save({'msg': msg}, Path(out_dir) / f'stale{rank}.pt')
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_stale_global_batch_is_detected_on_every_rank(tmp_path):
    """enable_data_parallel(global_batch, offset) has to be repeated whenever the minibatch grows; a stale value gives
    every rank wrong 1/B loss weights.  The ranks' sample counts ride on the loss all-reduce and every rank raises."""
    port = 29500 + os.getpid() % 2000
    mp.spawn(_stale_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert 'global' in torch.load(tmp_path / f'stale{r}.pt')['msg']


def _peer_failure_worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    torch.set_num_threads(2)
    import time
    import torch.distributed as dist
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_predictor(H, W, 1)
    p.enable_data_parallel(2, rank)
    full = synth.make_batch(2, H, W, seed=4)
    batch = {k: v[rank:rank + 1].clone() for k, v in full.items()}
    p.adapt(None, dict(batch))                      # a good step first
    w1, steps1 = p.engine.w.clone(), p.engine.adam_step_count
    bad = dict(batch)
    if rank == 1:
        del bad['rgb_aug', -1, 0]                   # this rank's minibatch is malformed: it fails before any collective
    t0 = time.monotonic()
    kind, msg, agreed = '', '', False
    try:
        p.adapt(None, bad)
    except Exception as e:          # noqa: BLE001
        kind, msg, agreed = type(e).__name__, str(e), bool(getattr(e, 'dp_agreed', False))
    waited = time.monotonic() - t0
    same = bool(torch.equal(p.engine.w, w1)) and p.engine.adam_step_count == steps1
    out, _ = p.adapt(None, dict(batch))             # the group is still in step: the next good step pairs up again
    torch.save({'kind': kind, 'msg': msg, 'agreed': agreed, 'waited': waited, 'same': same, 'w': p.engine.w.clone()},
               Path(out_dir) / f'peer{rank}.pt')
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_a_failed_rank_is_agreed_on_by_every_rank(tmp_path):
    """one rank of a data-parallel group fails before its step's exchange (a malformed minibatch).  It
    completes the step's collectives with the status word set; the healthy rank raises DataParallelPeerFailure at once instead
    of sitting in the all-reduce until the communicator times out; no rank applies the step; the next step pairs up again and
    leaves bit-identical replicas."""
    port = 29500 + (os.getpid() % 2000) + 23
    mp.spawn(_peer_failure_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'peer{r}.pt') for r in range(2))
    assert r0['kind'] == 'DataParallelPeerFailure' and 'peer rank' in r0['msg']
    assert r1['kind'] == 'KeyError' and r1['agreed']
    assert r0['same'] and r1['same']
    assert r0['waited'] < 60 and r1['waited'] < 60
    assert torch.equal(r0['w'], r1['w'])


def _peer_failure_with_empty_shard_worker(rank, world, port, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CLSLAM_EMU_THREADS='4')
    torch.set_num_threads(2)
    import time
    import torch.distributed as dist
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('emu')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = make_predictor(H, W, 1)
    n = 1 if rank < 2 else 0                        # shards 1 + 1 + 0: the third rank holds no sample (world > minibatch)
    p.enable_data_parallel(2, min(rank, 2))
    full = synth.make_batch(2, H, W, seed=4)
    batch = {k: v[rank:rank + n].clone() for k, v in full.items()}
    p.adapt(None, dict(batch))                      # a good step first
    w1, steps1 = p.engine.w.clone(), p.engine.adam_step_count
    bad = dict(batch)
    if rank == 1:
        del bad['rgb_aug', -1, 0]
    t0 = time.monotonic()
    kind, agreed = '', False
    try:
        p.adapt(None, bad)
    except Exception as e:          # noqa: BLE001
        kind, agreed = type(e).__name__, bool(getattr(e, 'dp_agreed', False))
    waited = time.monotonic() - t0
    same = bool(torch.equal(p.engine.w, w1)) and p.engine.adam_step_count == steps1
    p.adapt(None, dict(batch))                      # every collective of the failed step was matched: the next one pairs up
    torch.save({'kind': kind, 'agreed': agreed, 'waited': waited, 'same': same, 'w': p.engine.w.clone()}, Path(out_dir) / f'peer3_{rank}.pt')
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_a_failed_rank_beside_an_empty_shard_leaves_no_collective_unmatched(tmp_path):
    """the rank WITHOUT samples used to raise right behind the loss exchange, before posting the step's gradient
    all-reduce(s) -- the ranks with samples (and the failing rank's _dp_abort_step) post them first and check afterwards, so one
    collective stayed unmatched until the communicator timed out.  Three ranks (1 + 1 + 0 samples), the middle one fails: all
    three raise at once, nobody applies the step, and the following good step runs on all of them."""
    port = 29500 + (os.getpid() % 2000) + 41
    mp.spawn(_peer_failure_with_empty_shard_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    r = [torch.load(tmp_path / f'peer3_{i}.pt') for i in range(3)]
    assert r[0]['kind'] == 'DataParallelPeerFailure' and r[2]['kind'] == 'DataParallelPeerFailure'
    assert r[1]['kind'] == 'KeyError' and r[1]['agreed']
    assert all(x['same'] for x in r) and all(x['waited'] < 60 for x in r)
    assert torch.equal(r[0]['w'], r[1]['w']) and torch.equal(r[0]['w'], r[2]['w'])
