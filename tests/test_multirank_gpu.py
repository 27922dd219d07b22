"""The multi-rank code on a real MI355X with TWO PROCESSES SHARING THE ONE GPU of the test box (backend gloo on device
tensors; RCCL itself needs two devices and keeps its own test, test_data_parallel.py::test_two_ranks_nccl_equal_single_rank).
What runs here for the first time on hardware: real HIP streams under the asynchronous tail (gradient reduction ->
all-reduce -> Adam on their own stream while the next step's frozen encoders run), `replicas_in_sync` / `gather_outputs` /
`install_weights` on device memory, the data-parallel shards of BASELINE configs 4 and 5 at their real sizes
(3 + 2 triplets at 192x640; 2 + 1 at 384x1280 with the loop-closure encoder forward every frame on the rank that holds
the online frame), and the CoVIO asynchronous predict/adapt mode (clslam_hip.async_mode, SURVEY.md 8f rank 3;
reference README.md:62,171-172).  Every check compares against ONE process computing the same thing on the same GPU."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _setup(rank, world, port, backend='gloo'):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    from emu_util import use_backend
    torch.cuda.set_device(0)                       # both ranks on the one GPU
    use_backend('hip')
    if backend == 'nccl':          # RCCL: one rank per device (a single rank here: the box has one GPU)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    return dist


def _lcd_weights():
    sys.path.insert(0, str(ROOT / 'tests'))
    from test_lcd_encoder import _weights
    return _weights()[1]


def _dp_worker(rank, world, port, H, W, counts, steps, frames, lcd, async_tail, out_dir, buckets=None, tag='', backend='gloo'):
    os.environ['CLSLAM_ASYNC_TAIL'] = '1' if async_tail else '0'
    if buckets is not None:
        os.environ['CLSLAM_GRAD_BUCKETS'] = str(buckets)
    dist = _setup(rank, world, port, backend)
    from clslam_hip import synth
    from predictor_util import make_predictor
    B = sum(counts)
    off, n = sum(counts[:rank]), counts[rank]
    p = make_predictor(H, W, n)
    p.enable_data_parallel(B, off)
    assert p.engine.async_tail == async_tail
    enc = None
    if lcd and rank == 0:
        from loop_closure_detection import FeatureEncoder
        enc = FeatureEncoder(p.device, weights=_lcd_weights())
    feats, rec = [], None
    for f in range(frames):
        full = synth.make_batch(B, H, W, seed=4 + f)
        noise = synth.make_noise(B, H, W, seed=8 + f)
        p.set_tie_break_noise({s: v[off:off + n].cuda() for s, v in noise.items()})
        batch = {k: v[off:off + n].clone().pin_memory() for k, v in full.items()}     # host minibatch: uploads inside adapt()
        if enc is not None:                                                           # slam.py:180,223: rgb(+1, 0) of the online frame
            feats.append(enc(full['rgb', 1, 0][:1].cuda()).cpu())
        out, losses = p.adapt(None, batch, steps=steps)
        if f == 0:
            p.engine.wait_training()
            first = {'g': p.engine.g.clone().cpu(), 'loss': {k: v.detach().cpu().clone() for k, v in losses.items()},
                     'depth': out['depth', 0].cpu()}
        rec = (out, losses)
    out, losses = rec
    everything = p.gather_outputs(out)
    in_sync = p.replicas_in_sync()
    if rank == 1:                                   # a single flipped mantissa bit on one rank must be noticed
        p.engine.w.view(torch.int32)[12345] ^= 1
    diverged_seen = not p.replicas_in_sync()
    if rank == 1:
        p.engine.w.view(torch.int32)[12345] ^= 1
    p.synchronize()
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu()      # noqa: E731
    torch.save({'in_sync': in_sync, 'diverged_seen': diverged_seen, 'full_depth': cpu(everything['depth', 0]),
                'full_T': cpu(everything['cam_T_cam', 0, -1]), 'g': cpu(p.engine.g), 'w': cpu(p.engine.w), 'm': cpu(p.engine.m),
                'loss': {k: cpu(v).clone() for k, v in losses.items()}, 'T': cpu(out['cam_T_cam', 0, 1]), 'feats': feats, 'first': first},
               Path(out_dir) / f'rank{rank}_{int(async_tail)}{tag}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('H,W,counts,lcd', [(192, 640, [3, 2], False), (384, 1280, [2, 1], True)],
                         ids=['config4-shards-3+2@192x640', 'config5-shards-2+1@384x1280+lcd'])
def test_two_ranks_on_one_gpu_equal_single_rank(tmp_path, H, W, counts, lcd):
    sys.path.insert(0, str(ROOT / 'tests'))
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('hip')
    B, steps, frames = sum(counts), 1, 2
    for tail in (True, False):
        port = 29500 + (os.getpid() % 2000) + (3 if tail else 5)
        mp.start_processes(_dp_worker, args=(2, port, H, W, counts, steps, frames, lcd, tail, str(tmp_path)), nprocs=2, join=True,
                           start_method='spawn')
    r = {(k, t): torch.load(tmp_path / f'rank{k}_{t}.pt') for k in (0, 1) for t in (0, 1)}
    # single process, same GPU, same two frames
    p = make_predictor(H, W, B)
    for f in range(frames):
        p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=8 + f).items()})
        out, losses = p.adapt(None, {k: v.clone() for k, v in synth.make_batch(B, H, W, seed=4 + f).items()}, steps=steps)
        if f == 0:
            g_first, loss_first, depth_first = p.engine.g.cpu(), {k: float(v) for k, v in losses.items()}, out['depth', 0].cpu()
    torch.cuda.synchronize()
    w = p.engine.w.cpu()
    for t in (0, 1):
        r0, r1 = r[0, t], r[1, t]
        # identical replicas after the all-reduce + Adam, checksum check works on device tensors
        assert torch.equal(r0['g'], r1['g']) and torch.equal(r0['w'], r1['w']) and torch.equal(r0['m'], r1['m'])
        assert r0['in_sync'] and r1['in_sync'] and r0['diverged_seen'] and r1['diverged_seen']
        assert torch.equal(r0['full_depth'], r1['full_depth']) and r0['full_depth'].shape == out['depth', 0].shape
        # first frame (same weights everywhere): the all-reduced gradient is the single-process gradient up to summation
        # order, the stream-K cuts (they follow the shard size) and a handful of kink flips of the piecewise-smooth loss
        # (tests/test_full_size.py holds the same shard-sum rule inside one process)
        e_g = float((r0['first']['g'] - g_first).abs().max() / g_first.abs().max())
        assert e_g < 2e-2, e_g
        for k, v in loss_first.items():
            assert abs(float(r0['first']['loss'][k]) - v) <= 1e-4 * max(abs(v), 1e-3), k
        assert torch.allclose(r0['first']['depth'], depth_first[:counts[0]], rtol=1e-4, atol=0)
        # second frame: Adam's first update is lr * sign(g), so near-zero gradient entries land on either side and the
        # weights -- and with them the second step -- differ by lr-sized flips: bounded, not compared tightly
        assert float((r0['w'] - w).abs().max()) < 4.5e-4
        rel = ((r0['full_depth'] - out['depth', 0].cpu()).abs() / out['depth', 0].cpu().abs()).mean()
        assert float(rel) < 1e-2, float(rel)
        assert all(torch.isfinite(v).all() for v in r0['loss'].values())
    # the asynchronous tail is invisible: bitwise the serial order (same shards, same kernels, same all-reduce)
    for k in (0, 1):
        for name in ('g', 'w', 'm', 'full_depth', 'T'):
            assert torch.equal(r[k, 0][name], r[k, 1][name]), (k, name)
    if lcd:
        from loop_closure_detection import FeatureEncoder
        enc = FeatureEncoder(p.device, weights=_lcd_weights())
        for f in range(frames):
            ref = enc(synth.make_batch(B, H, W, seed=4 + f)['rgb', 1, 0][:1].cuda()).cpu()
            assert ref.shape == (1, 576) and torch.equal(r[0, 1]['feats'][f], ref)


@pytest.mark.timeout(1500)
def test_bucketed_gradient_exchange_is_bitwise_the_single_all_reduce(tmp_path):
    """SURVEY.md section 5 (K19, "bucketed / overlapped"): three all-reduces in backward-completion order, the first two
    underneath the rest of the backward on the tail stream (Engine.grad_buckets), against ONE all-reduce of the whole arena
    after the backward.  Two ranks (a + b is the same number in either order), 3 + 2 triplets at 192x640, three frames of
    adapt(steps=2) -- the first backward of a workspace exchanges without overlap, the later ones overlapped: gradients,
    weights, moments and outputs are bitwise equal, with the asynchronous tail and without it."""
    sys.path.insert(0, str(ROOT / 'tests'))
    H, W, counts = 192, 640, [3, 2]
    runs = {}
    for i, (buckets, tail) in enumerate([(1, True), (3, True), (2, True), (3, False)]):
        port = 29500 + (os.getpid() % 2000) + 21 + 2 * i
        tag = f'_b{buckets}'
        mp.start_processes(_dp_worker, args=(2, port, H, W, counts, 2, 3, False, tail, str(tmp_path), buckets, tag), nprocs=2,
                           join=True, start_method='spawn')
        runs[buckets, tail] = [torch.load(tmp_path / f'rank{k}_{int(tail)}{tag}.pt') for k in (0, 1)]
    ref = runs[1, True]
    for key, got in runs.items():
        for k in (0, 1):
            assert torch.equal(got[0]['g'], got[1]['g']) and got[k]['in_sync']
            for name in ('g', 'w', 'm', 'full_depth', 'T'):
                assert torch.equal(ref[k][name], got[k][name]), (key, k, name)
            assert torch.equal(ref[k]['first']['g'], got[k]['first']['g'])


@pytest.mark.timeout(1500)
def test_one_rank_rccl_is_bitwise_the_single_process_step(tmp_path):
    """The RCCL code path on the one GPU of the box (review r4 item 6): backend 'nccl', world size 1 -- enable_data_parallel(B, 0),
    the loss exchange with its consistency tag, the three bucketed ncclAllReduce calls issued on the TAIL stream beside the
    persistent stream-K / Winograd kernels of the backward, `record_stream`-free bucket slices, the asynchronous tail on and
    off; three frames of adapt(steps=2) at 192x640, B = 5.  A one-rank all-reduce is the identity, so gradients, weights, Adam
    moments and outputs must be BITWISE those of the plain single-process step.  It measures no scaling (one rank); it is the
    first time the collective library itself meets this stream schedule (rounds 1-4 ordered it through gloo's host-blocking
    calls only).  profiles/r05_rccl1_timeline.txt shows which stream the RCCL kernels run on."""
    sys.path.insert(0, str(ROOT / 'tests'))
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    H, W, B, steps, frames = 192, 640, 5, 2, 3
    for i, tail in enumerate((True, False)):
        port = 29500 + (os.getpid() % 2000) + 41 + 2 * i
        mp.start_processes(_dp_worker, args=(1, port, H, W, [B], steps, frames, False, tail, str(tmp_path), 3, '_rccl1', 'nccl'),
                           nprocs=1, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank0_{t}_rccl1.pt') for t in (1, 0)]
    use_backend('hip')
    p = make_predictor(H, W, B)
    for f in range(frames):
        p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(B, H, W, seed=8 + f).items()})
        out, losses = p.adapt(None, {k: v.clone().pin_memory() for k, v in synth.make_batch(B, H, W, seed=4 + f).items()}, steps=steps)
    p.synchronize()
    for r in got:
        assert r['in_sync']
        for name, ref in (('g', p.engine.g), ('w', p.engine.w), ('m', p.engine.m), ('full_depth', out['depth', 0]), ('T', out['cam_T_cam', 0, 1])):
            assert torch.equal(r[name], ref.detach().cpu()), name
        for k, v in losses.items():
            assert torch.equal(r['loss'][k].reshape(-1), v.detach().cpu().reshape(-1)), k


# ---- CoVIO asynchronous predict / adapt mode ------------------------------------------------------------------------
AH, AW, AB, FRAMES, EVERY = 192, 640, 2, 4, 2


def _async_worker(rank, world, port, out_dir):
    dist = _setup(rank, world, port)
    from clslam_hip import synth
    from clslam_hip.async_mode import AsyncAdaptation
    from predictor_util import make_predictor
    p = make_predictor(AH, AW, AB if rank == 1 else 1)
    am = AsyncAdaptation(p, sync_every=EVERY)
    log = []
    for f in range(FRAMES):
        online = synth.make_batch(1, AH, AW, seed=60 + f)
        if rank == 1:
            replay = synth.make_batch(AB - 1, AH, AW, seed=80 + f)
            training = {k: torch.cat([online[k], replay[k]]) for k in online}
            p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(AB, AH, AW, seed=f).items()})
            out, losses = am.step(f, online, training)
            rec = {'frame': f, 'loss': float(losses['loss'])}
            if (f + 1) % EVERY == 0:
                p.engine.wait_training()
                rec['snapshot'] = p.engine.w.clone().cpu()
        else:
            p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(1, AH, AW, seed=100 + f).items()})
            am.keep_used_weights = True
            if f == EVERY:                 # (test only) make the first install deterministic: wait for the transfer posted after
                am._install(block=True)    # frame EVERY-1 instead of picking it up whenever it happens to have completed
            out, _ = am.step(f, online)
            rec = {'frame': f, 'weights_frame': am.used_weights_frame, 'installs': am.installs, 'w_used': am.used_weights.cpu(),
                   'depth': out['depth', 0].cpu(), 'T': out['cam_T_cam', 0, 1].cpu()}
        log.append(rec)
    am.flush()
    torch.cuda.synchronize()
    final = {'w': p.engine.w.cpu(), 'weights_frame': am.weights_frame, 'installs': am.installs, 'lag_bound': am.lag_bound_frames}
    torch.save({'log': log, 'final': final}, Path(out_dir) / f'rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_async_predict_adapt_mode_on_the_gpu(tmp_path):
    """Inference replica + training replica as two processes on the GPU: the weight arena travels by an asynchronous
    broadcast of a DEVICE staging buffer, is installed into device memory at a frame boundary, and every prediction is
    bitwise what a single process holding that snapshot predicts."""
    sys.path.insert(0, str(ROOT / 'tests'))
    port = 29500 + (os.getpid() % 2000) + 9
    mp.start_processes(_async_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    inf, trn = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    snaps = {r['frame']: r['snapshot'] for r in trn['log'] if 'snapshot' in r}
    assert sorted(snaps) == [1, 3] and not torch.equal(snaps[1], snaps[3])
    assert all(torch.isfinite(torch.tensor(r['loss'])) for r in trn['log'])
    assert inf['final']['weights_frame'] == 3 and inf['final']['installs'] == 2
    assert torch.equal(inf['final']['w'], snaps[3])                      # after flush(): the trainer's last snapshot, bit for bit
    used = [r['weights_frame'] for r in inf['log']]
    assert used == sorted(used) and used[0] == -1
    for rec in inf['log']:
        f, wf = rec['frame'], rec['weights_frame']
        assert f - wf <= inf['final']['lag_bound']
        if wf >= 0:
            assert wf in snaps and wf < f and torch.equal(rec['w_used'], snaps[wf])     # exactly a snapshot, never a mix
    from clslam_hip import synth
    from emu_util import use_backend
    from predictor_util import make_predictor
    use_backend('hip')
    p = make_predictor(AH, AW, 1)
    checked = 0
    for rec in inf['log']:
        if rec['weights_frame'] < 0:
            continue
        p.engine.install_weights(snaps[rec['weights_frame']].cuda())
        p.set_tie_break_noise({s: v.cuda() for s, v in synth.make_noise(1, AH, AW, seed=100 + rec['frame']).items()})
        out, _ = p.adapt(synth.make_batch(1, AH, AW, seed=60 + rec['frame']), None)
        assert torch.equal(out['depth', 0].cpu(), rec['depth']) and torch.equal(out['cam_T_cam', 0, 1].cpu(), rec['T'])
        checked += 1
    assert checked >= 1
