"""BASELINE.json's full sizes (192x640, 384x1280) on the MI355X.  The oracle finishes a B=1 step at these
sizes in seconds, so B=1 is held to it directly; the B=5 adapt step of the benchmark is held to
size-independent properties of the path: the loss and every gradient are sums of per-sample terms
(SURVEY.md 8e), so the step over the full minibatch must equal the sum of the steps over its shards, and two
identical runs must agree bit for bit (every reduction in the path has a fixed order)."""
import pytest
import torch

from clslam_hip import synth
from emu_util import use_backend
from helpers import make_oracle, rel_err
from predictor_util import make_predictor

pytestmark = pytest.mark.gpu


class _NoDist:
    """Stand-in for torch.distributed inside ONE process: shards are run one after the other and summed by the
    test itself."""
    @staticmethod
    def all_reduce(t, group=None):
        return None


def _shard(batch, lo, hi):
    return {k: v[lo:hi].clone() for k, v in batch.items()}


@pytest.mark.parametrize('H,W', [(192, 640), (384, 1280)])
def test_b1_step_matches_oracle_at_full_size(H, W):
    use_backend('hip')
    p = make_predictor(H, W, 1)
    o = make_oracle(H, W, 1)
    batch = synth.make_batch(1, H, W, seed=3)
    noise = synth.make_noise(1, H, W, seed=4)
    p.set_tie_break_noise(noise)
    out, losses = p.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=1)
    oo, ol = o.adapt(batch, steps=1, noise_per_step=[noise])
    oo = {k: v.detach() for k, v in oo.items()}
    ol = {k: v.detach() for k, v in ol.items()}
    # 1e-4 relative (north_star) on the step-0 quantities
    assert rel_err(out['depth', 0].cpu(), oo['depth', 0]) < 1e-4
    for s in range(4):
        assert rel_err(out['disp', s].cpu(), oo['disp', s]) < 1e-4
    for f in (-1, 1):
        assert rel_err(out['cam_T_cam', 0, f].cpu(), oo['cam_T_cam', 0, f]) < 1e-4
    assert abs(float(losses['loss']) - float(ol['loss'])) < 1e-4 * abs(float(ol['loss']))
    # after one Adam step (lr 1e-4, first update = lr * sign(g)): weights within 2 lr of the oracle's, at most a
    # fraction of a percent of them on the other side of a sign flip (conditioning)
    p.engine.sync_modules()
    lr = 1e-4
    flipped = total = 0
    for name in ('depth_decoder', 'pose_decoder'):
        sd_o = o.models[name].state_dict()
        for k, v in torch.nn.Module.state_dict(p.models[name]).items():
            d = (v.cpu() - sd_o[k]).abs()
            assert float(d.max()) <= 2.05 * lr, (name, k, float(d.max()))
            flipped += int((d > 0.5 * lr).sum()); total += d.numel()
    assert flipped <= 0.01 * total, (flipped, total)


def test_b5_step_is_the_sum_of_its_shards_and_deterministic(capsys):
    """192x640, 1 online + 4 replay triplets (the benchmark's step).  Shards [0:3] and [3:5] run as
    data-parallel ranks would (global sample weights, all smoothness terms on the shard holding sample 0),
    one after the other on this GPU; gradients and losses add up to the full-batch step."""
    use_backend('hip')
    H, W, B = 192, 640, 5
    batch = synth.make_batch(B, H, W, seed=0)
    noise = synth.make_noise(B, H, W, seed=1)

    def run(lo, hi, dp):
        p = make_predictor(H, W, hi - lo)
        if dp:
            p._dp = dict(group=None, global_batch=B, offset=lo, dist=_NoDist)
            p._check_dp_tag = lambda tag: None      # (the stand-in sums nothing: the sample counts never add up to B)
        p.set_tie_break_noise({s: n[lo:hi].contiguous() for s, n in noise.items()})
        out, losses = p.adapt(None, _shard(batch, lo, hi), steps=1)
        return p.engine.g.clone(), {k: v.clone() for k, v in losses.items()}, out['disp', 0].clone(), p.engine.w.clone()

    g_full, l_full, d_full, w_full = run(0, B, False)
    g_again, l_again, d_again, w_again = run(0, B, False)
    assert torch.equal(g_full, g_again) and torch.equal(w_full, w_again) and torch.equal(d_full, d_again)
    g_a, l_a, d_a, _ = run(0, 3, True)
    g_b, l_b, d_b, _ = run(3, B, True)
    # the forward of a sample does not depend on its batch mates (only the tile configuration, and with it the
    # fp32 summation order, changes with the batch size)
    assert rel_err(torch.cat([d_a, d_b]).cpu(), d_full.cpu()) < 1e-5
    # Gradients: the tile configuration (hence the fp32 summation order) of a few layers depends on the batch
    # size, the forwards differ by ~1e-6, and the loss gradient is very ill-conditioned with respect to that:
    # in the ORACLE (torch CPU) a 2e-8 change of the pose moves the decoder gradients of this input by 0.3-1 %
    # (conditioning).  With bitwise-equal forwards (CPU emulator, 64x128) the rule holds to 7e-8.
    diff = (g_a + g_b - g_full).double()
    l2 = float(diff.norm() / g_full.double().norm())
    mx = float(diff.abs().max() / g_full.abs().max())
    with capsys.disabled():
        print(f'[hip 192x640 B=5] shard-sum rule (3 + 2): relative L2 {l2:.2e}, max {mx:.2e}')
    # measured on the MI355X over the round's builds: L2 3.7e-3 ... 5.2e-3, max 4.8e-3 ... 5.6e-3 (a handful of kink flips between
    # the B = 5 and the 3 + 2 forwards, tests/test_backward_parity.py counts them)
    assert l2 < 1.2e-2 and mx < 1.5e-2, (l2, mx)
    for k in ('loss', 'velocity_loss', 'reprojection_loss/scale_0', 'smooth_loss/scale_0', 'reg_loss/scale_3'):
        assert abs(float(l_a[k]) + float(l_b[k]) - float(l_full[k])) < 2e-5 * max(abs(float(l_full[k])), 1e-4), k


def test_b1_step_matches_reference_golden_at_full_size():
    """192x640: the HIP path against vectors produced by the REAL reference at this size (tests/golden/make_golden.py,
    adapt_full_b1.npz) -- not only against the oracle."""
    import math
    from clslam_hip.engine import TrainableLayout
    from helpers import load_golden
    from test_oracle_golden import _check_full_size
    use_backend('hip')
    g = load_golden('adapt_full_b1')
    H, W, B, seed_b, seed_n = (int(v) for v in g['params'])
    p = make_predictor(H, W, B)
    p.set_tie_break_noise(synth.make_noise(B, H, W, seed=seed_n))
    out, losses = p.adapt(None, synth.make_batch(B, H, W, seed=seed_b), steps=1)
    eng = p.engine
    eng.wait_training()
    grads = {name: TrainableLayout.to_reference(eng.g[off:off + math.prod(shape)], shape).cpu()
             for name, off, shape in eng.layout.entries}
    # step-0 outputs and losses at the 1e-4 bar; gradient norms carry the selection flips (tests/test_backward_parity.py).
    # tol_warp: the sampling positions are held to 2.5e-4 px in x / 8e-5 px in y against the float64 oracle
    # (tests/test_warp_positions.py -- torch's own fp32 is at 1.9e-4 / 6e-5), times an image gradient of up to 1 per px.
    _check_full_size(g, out, losses, grads, 1e-4, 1e-4, 3e-2, tol_warp=5e-4)
