"""BASELINE configs 4 and 5 at their REAL workload on the one MI355X of the test box (missing #1).

  config 4: 192x640, 1 online + K=32 replay triplets (B = 33) sharded over 8 ranks as 5,4,4,4,4,4,4,4
  config 5: 384x1280, K=8 (B = 9) sharded as 2,1,1,1,1,1,1,1, + the loop-closure encoder forward on the online frame
  (slam/slam.py:99,174-178,223,300-309; config/config_adapt.yaml:33).

Three things hold each of them:
  (a) ONE process: the full-batch step is deterministic (bitwise), its forward agrees with the ORACLE on the same B = 33 /
      B = 9 minibatch at the 1e-4 bar (depth, disparities, poses, every loss scalar), and the step is the SUM of its eight
      shards run one after the other the way the ranks run them (global 1/B sample weights, all B smoothness terms on the
      shard that holds sample 0) -- losses to 2e-5, gradients to the measured shard-sum range.
  (b) EIGHT processes sharing the GPU (backend gloo on device tensors; RCCL needs eight devices) with INJECTED tie-break
      noise: every rank holds bit-identical gradients, weights and Adam moments after the all-reduce + optimizer step, and
      the first frame's all-reduced gradient / losses / rank-0 outputs are the single process's.
What stays unmeasured is RCCL itself over xGMI (no multi-GPU box)."""
import hashlib
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from clslam_hip import synth
from emu_util import use_backend
from helpers import make_oracle, rel_err
from predictor_util import make_predictor

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu

CONFIGS = {
    'config4': dict(H=192, W=640, counts=[5, 4, 4, 4, 4, 4, 4, 4], lcd=False, seed=70),
    'config5': dict(H=384, W=1280, counts=[2, 1, 1, 1, 1, 1, 1, 1], lcd=True, seed=74),
}
# shard-sum rule, gradients (relative L2 / max over the arena): measured on the MI355X 3 + 2 at B = 5: 3.7e-3 / 4.8e-3; the
# eight-shard sums of B = 33 / B = 9 are printed by the test and bounded at ~2x what was measured there
SHARD_SUM_TOL = {'config4': (1.5e-2, 2e-2), 'config5': (1.5e-2, 2e-2)}


class _NoDist:
    """torch.distributed stand-in inside ONE process: the shards run one after the other, the test adds them up."""
    @staticmethod
    def all_reduce(t, group=None):
        return None

    @staticmethod
    def get_world_size(group=None):
        return 1


def _lcd_weights():
    from clslam_hip import lcd
    return lcd.synthetic_state_dict()


@pytest.mark.parametrize('name', list(CONFIGS))
def test_full_batch_vs_oracle_and_sum_of_the_eight_shards(name, capsys):
    use_backend('hip')
    c = CONFIGS[name]
    H, W, counts = c['H'], c['W'], c['counts']
    B = sum(counts)
    batch = synth.make_batch(B, H, W, seed=c['seed'])
    noise = synth.make_noise(B, H, W, seed=c['seed'] + 1)

    def run(lo, hi, dp):
        p = make_predictor(H, W, hi - lo)
        if dp:
            p._dp = dict(group=None, global_batch=B, offset=lo, dist=_NoDist)
            p._check_dp_tag = lambda tag: None      # (the stand-in sums nothing: the sample counts never add up to B)
        p.set_tie_break_noise({s: n[lo:hi].contiguous() for s, n in noise.items()})
        out, losses = p.adapt(None, {k: v[lo:hi].clone() for k, v in batch.items()}, steps=1)
        keep = {k: out[k].clone() for k in [('disp', s) for s in range(4)] + [('depth', 0), ('cam_T_cam', 0, -1), ('cam_T_cam', 0, 1)]}
        return p.engine.g.clone(), {k: v.clone() for k, v in losses.items()}, keep, p.engine.w.clone()

    g_full, l_full, o_full, w_full = run(0, B, False)
    g_again, l_again, o_again, w_again = run(0, B, False)
    assert torch.equal(g_full, g_again) and torch.equal(w_full, w_again)
    assert all(torch.equal(o_full[k], o_again[k]) for k in o_full) and all(torch.equal(l_full[k], l_again[k]) for k in l_full)

    # (1) the full minibatch against the oracle (forward quantities; the oracle's B = 33 / B = 9 forward takes seconds)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        o = make_oracle(H, W, B)
        o.set_eval()
        with torch.no_grad():
            oo, ol = o.process_batch(batch, noise)
    finally:
        torch.set_num_threads(threads)
    errs = {str(k): rel_err(o_full[k].cpu(), oo[k]) for k in o_full}
    errs.update({k: abs(float(l_full[k]) - float(ol[k])) / max(abs(float(ol[k])), 1e-3) for k in ol})
    assert max(errs.values()) < 1e-4, {k: v for k, v in errs.items() if v >= 1e-4}

    # (2) the eight shards, run as the ranks run them, add up to the full-batch step
    g_sum, l_sum, parts = torch.zeros_like(g_full), {}, []
    lo = 0
    for n in counts:
        g, l, keep, _ = run(lo, lo + n, True)
        g_sum += g
        for k, v in l.items():
            l_sum[k] = l_sum.get(k, 0.0) + float(v)
        parts.append(keep['disp', 0])
        lo += n
    assert rel_err(torch.cat(parts).cpu(), o_full['disp', 0].cpu()) < 1e-5
    diff = (g_sum - g_full).double()
    l2 = float(diff.norm() / g_full.double().norm())
    mx = float(diff.abs().max() / g_full.abs().max())
    with capsys.disabled():
        print(f'\n[{name}: {H}x{W} B={B} as {counts}] full batch vs oracle: worst relative error {max(errs.values()):.2e} '
              f'(depth {errs[str(("depth", 0))]:.1e}, loss {errs["loss"]:.1e}); shard-sum rule over the eight shards: '
              f'gradient relative L2 {l2:.2e}, max {mx:.2e}')
    tol_l2, tol_mx = SHARD_SUM_TOL[name]
    assert l2 < tol_l2 and mx < tol_mx, (l2, mx)
    for k, v in l_full.items():
        assert abs(l_sum[k] - float(v)) < 2e-5 * max(abs(float(v)), 1e-4), (k, l_sum[k], float(v))

    if c['lcd']:       # config 5: the loop-closure descriptor of the online frame (slam.py:223), HIP vs the oracle's MobileNetV3
        from loop_closure_detection import FeatureEncoder
        from oracle.mobilenet import MobileNetV3SmallFeatures, feature_encoder
        sd = _lcd_weights()
        enc = FeatureEncoder(torch.device('cuda:0'), weights=sd)
        m = MobileNetV3SmallFeatures()
        m.load_state_dict({**m.state_dict(), **sd})
        img = batch['rgb', 1, 0][:1]
        assert rel_err(enc(img.cuda()).cpu(), feature_encoder(m, img)) < 1e-4


# ---- eight processes on the one GPU --------------------------------------------------------------------------------
def _sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def _worker(rank, world, port, names, data_dir, out_dir):
    for p in (ROOT / 'cl-slam_amd', ROOT, ROOT / 'tests'):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(0)                       # all ranks on the one GPU
    use_backend('hip')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    result = {}
    for name in names:
        c = CONFIGS[name]
        H, W, counts = c['H'], c['W'], c['counts']
        B, off, n = sum(counts), sum(counts[:rank]), counts[rank]
        shard = torch.load(Path(data_dir) / f'{name}_rank{rank}.pt')
        p = make_predictor(H, W, n)
        p.enable_data_parallel(B, off)
        p.set_tie_break_noise({s: v.cuda() for s, v in shard['noise'].items()})
        enc = None
        if c['lcd'] and rank == 0:
            from clslam_hip import lcd
            from loop_closure_detection import FeatureEncoder
            enc = FeatureEncoder(p.device, weights=lcd.synthetic_state_dict())
        rec = {}
        for frame in range(2):
            batch = {k: v.clone().pin_memory() for k, v in shard['batch'].items()}      # host minibatch: uploads inside adapt()
            out, losses = p.adapt(None, batch, steps=1)
            if enc is not None:
                rec[f'lcd{frame}'] = enc(shard['batch']['rgb', 1, 0][:1].cuda()).cpu()
            if frame == 0:
                rec['g0_sha'] = _sha(p.engine.g)
                rec['loss0'] = {k: float(v) for k, v in losses.items()}
                if rank == 0:
                    rec['g0'] = p.engine.g.cpu().clone()
                    rec['depth0'] = out['depth', 0].cpu().clone()
                    rec['T0'] = out['cam_T_cam', 0, 1].cpu().clone()
        rec.update(g=_sha(p.engine.g), w=_sha(p.engine.w), m=_sha(p.engine.m), v=_sha(p.engine.v), in_sync=p.replicas_in_sync(),
                   loss1={k: float(v) for k, v in losses.items()})
        if rank == 0:
            rec['w1'] = p.engine.w.cpu().clone()
        result[name] = rec
        del p
        torch.cuda.empty_cache()
        dist.barrier()
    torch.save(result, Path(out_dir) / f'rank{rank}.pt')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(2400)
def test_eight_ranks_on_one_gpu_with_the_real_shardings(tmp_path, capsys):
    use_backend('hip')
    names = list(CONFIGS)
    data_dir = tmp_path / 'data'
    data_dir.mkdir()
    full = {}
    for name in names:
        c = CONFIGS[name]
        B = sum(c['counts'])
        batch = synth.make_batch(B, c['H'], c['W'], seed=c['seed'] + 2)
        noise = synth.make_noise(B, c['H'], c['W'], seed=c['seed'] + 3)
        full[name] = (batch, noise)
        lo = 0
        for rank, n in enumerate(c['counts']):
            torch.save({'batch': {k: v[lo:lo + n].clone() for k, v in batch.items()},
                        'noise': {s: v[lo:lo + n].clone() for s, v in noise.items()}}, data_dir / f'{name}_rank{rank}.pt')
            lo += n
    port = 29500 + (os.getpid() % 2000) + 11
    mp.start_processes(_worker, args=(8, port, names, str(data_dir), str(tmp_path)), nprocs=8, join=True, start_method='spawn')
    ranks = [torch.load(tmp_path / f'rank{r}.pt') for r in range(8)]
    for name in names:
        c = CONFIGS[name]
        H, W, counts = c['H'], c['W'], c['counts']
        B = sum(counts)
        r0 = ranks[0][name]
        # bit-identical replicas on all eight ranks: the all-reduced gradient of both frames, weights and both Adam moments
        for key in ('g0_sha', 'g', 'w', 'm', 'v'):
            assert len({r[name][key] for r in ranks}) == 1, (name, key)
        assert all(r[name]['in_sync'] for r in ranks)
        for k, v in r0['loss0'].items():            # the all-reduced loss scalars are the same numbers on every rank
            assert all(r[name]['loss0'][k] == v for r in ranks), k
        # ... and they are the single process's step on the same minibatch and noise
        batch, noise = full[name]
        p = make_predictor(H, W, B)
        p.set_tie_break_noise({s: v.cuda() for s, v in noise.items()})
        out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
        g1 = p.engine.g.cpu()
        diff = (r0['g0'] - g1).double()
        l2, mx = float(diff.norm() / g1.double().norm()), float(diff.abs().max() / g1.abs().max())
        with capsys.disabled():
            print(f'\n[{name}: 8 ranks {counts} on one GPU] first-frame all-reduced gradient vs one process: relative L2 {l2:.2e}, '
                  f'max {mx:.2e}; loss {r0["loss0"]["loss"]:.6f} vs {float(losses["loss"]):.6f}')
        tol_l2, tol_mx = SHARD_SUM_TOL[name]
        assert l2 < tol_l2 and mx < tol_mx, (name, l2, mx)
        for k, v in losses.items():
            assert abs(r0['loss0'][k] - float(v)) <= 1e-4 * max(abs(float(v)), 1e-3), (name, k)
        assert torch.allclose(r0['depth0'], out['depth', 0][:counts[0]].cpu(), rtol=1e-4, atol=0)
        assert torch.allclose(r0['T0'], out['cam_T_cam', 0, 1][:counts[0]].cpu(), rtol=1e-4, atol=1e-7)
        # second frame ran on the updated weights: finite, and the replicas' weights are one lr-sized update away from one process's
        out2, losses2 = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
        assert float((r0['w1'] - p.engine.w.cpu()).abs().max()) < 4.5e-4
        assert all(abs(v) < float('inf') for v in r0['loss1'].values())
        if c['lcd']:
            from clslam_hip import lcd
            from loop_closure_detection import FeatureEncoder
            enc = FeatureEncoder(p.device, weights=lcd.synthetic_state_dict())
            ref = enc(batch['rgb', 1, 0][:1].cuda()).cpu()
            assert torch.equal(r0['lcd0'], ref) and torch.equal(r0['lcd1'], ref)
        del p
        torch.cuda.empty_cache()
