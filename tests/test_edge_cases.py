"""Edge cases of the C ABI and the Python boundary: empty batches are no-ops, invalid arguments come back as
error codes + clslam_last_error() (ClslamError on the Python side, never a crash), shapes the kernels cannot
tile are refused up front, and the predictor validates like the reference's constructor (dpp.py:87-120)."""
import pytest
import torch

from clslam_hip import ops
from clslam_hip._lib import ClslamError
from emu_util import BACKENDS, use_backend


@pytest.mark.parametrize('backend', BACKENDS)
def test_empty_batches_are_no_ops(backend):
    dev = use_backend(backend)
    z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
    out = torch.full((0, 8, 8, 16), 1.0, device=dev)
    ops.conv2d(z(0, 8, 8, 16), z(16, 9, 16), out, ksize=3)
    ops.maxpool3x3s2(z(0, 8, 8, 64), z(0, 4, 4, 64))
    ops.dispconv_fwd(z(0, 8, 8, 16), z(9, 16), z(1), z(0, 8, 8))
    ops.adam_step(z(0), z(0), z(0), z(0), 1e-4, 1)
    ops.global_avgpool(z(0, 4, 4, 16), z(0, 16))
    from clslam_hip.ingest import ImagePyramid, color_jitter
    assert ImagePyramid(32, 64)(torch.zeros(0, 40, 70, 3, dtype=torch.uint8, device=dev))[3].shape == (0, 3, 4, 8)
    assert color_jitter(torch.zeros(0, 8, 8, 3, dtype=torch.uint8, device=dev), [0, 3], [1.1, 1, 1, 0.1]).shape == (0, 8, 8, 3)


@pytest.mark.parametrize('backend', BACKENDS)
def test_invalid_arguments_raise_with_message(backend):
    dev = use_backend(backend)
    z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
    with pytest.raises(ClslamError, match='multiples of 16'):
        ops.conv2d(z(1, 8, 8, 12), z(16, 9, 12), z(1, 8, 8, 16), ksize=3)            # channel count the MFMA tiles cannot take
    with pytest.raises(ClslamError, match='ksize'):
        ops.conv2d(z(1, 8, 8, 16), z(16, 25, 16), z(1, 8, 8, 16), ksize=5, pad=2)
    with pytest.raises(ClslamError, match='output size'):
        ops.conv2d(z(1, 4, 4, 16), z(16, 9, 16), z(1, 9, 9, 16), ksize=3, upsample_a=True, pad=1)   # 8x8 in, 9x9 out
    with pytest.raises(ClslamError):
        ops.conv2d(z(1, 8, 8, 16).double(), z(16, 9, 16), z(1, 8, 8, 16), ksize=3)   # dtype checked before the call
    with pytest.raises(ClslamError):
        ops.conv2d(z(1, 8, 8, 32)[..., ::2], z(16, 9, 16), z(1, 8, 8, 16), ksize=3)  # non-contiguous
    with pytest.raises(ClslamError, match='hue'):
        from clslam_hip.ingest import color_jitter
        color_jitter(torch.zeros(1, 4, 4, 3, dtype=torch.uint8, device=dev), [3], [1, 1, 1, 0.7])
    # the error state does not stick: a good call afterwards works
    out = torch.full((1, 8, 8, 16), float('nan'), device=dev)
    ops.conv2d(z(1, 8, 8, 16), z(16, 9, 16), out, ksize=3)
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize('backend', BACKENDS)
def test_predictor_validation_matches_reference_constructor(backend):
    """dpp.py:87-120 and the engine's own shape contract."""
    use_backend(backend)
    from predictor_util import make_predictor
    with pytest.raises(ValueError):
        make_predictor(60, 128, 1)                       # not a multiple of 32: five stride-2 stages
    with pytest.raises((ValueError, NotImplementedError)):
        make_predictor(64, 128, 1, multiple_gpus=True)   # nn.DataParallel mode is not provided (one process per GPU)
    p = make_predictor(64, 128, 2)
    from clslam_hip import synth
    bad = synth.make_batch(3, 64, 128, seed=0)           # batch of 3 under a configured batch_size of 2 (dpp.py:1031-1032)
    with pytest.raises(RuntimeError, match='must match'):
        p.adapt(None, bad, steps=1)
    with pytest.raises(ClslamError):
        p.predict({k: (v[..., :96] if v.dim() == 4 else v) for k, v in synth.make_batch(1, 64, 128, seed=0).items()})   # wrong width
    # predict_pose validates like run_encoder/forward (the stem kernel takes H, W from its input)
    good = synth.make_batch(1, 64, 128, seed=0)['rgb', 0, 0]
    with pytest.raises(ClslamError, match='predict_pose'):
        p.predict_pose(good[0], good[0, :, :, :96])
    with pytest.raises(ClslamError, match='predict_pose'):
        p.predict_pose(good, torch.cat([good, good]))
    T, cov = p.predict_pose(good[0], good[0])
    assert T.shape == (4, 4)
    # disp_to_depth (utils.py:134-135): max_depth without min_depth is refused
    with pytest.raises(ValueError, match='min_depth is None'):
        make_predictor(64, 128, 1, min_depth=None, max_depth=80.0)


@pytest.mark.parametrize('backend', BACKENDS)
def test_single_weight_broadcasts_over_the_batch(backend):
    """a configured batch_size of 1 with a larger actual batch: the reference's (1,) weight vector broadcasts
    (dpp.py:1031-1032, :1073) -- every sample is weighted 1, the loss is the SUM over samples"""
    use_backend(backend)
    from clslam_hip import synth
    from predictor_util import make_predictor
    p = make_predictor(64, 128, 1)
    two = synth.make_batch(2, 64, 128, seed=5)
    noise = synth.make_noise(2, 64, 128, seed=5)
    p.set_tie_break_noise(noise)
    _, l2 = p.adapt({k: v.clone() for k, v in two.items()}, None)
    parts = []
    for i in range(2):
        p.set_tie_break_noise({s: v[i:i + 1] for s, v in noise.items()})
        _, li = p.adapt({k: v[i:i + 1].clone() for k, v in two.items()}, None)
        parts.append(float(li['reprojection_loss/scale_0']))
    assert abs(float(l2['reprojection_loss/scale_0']) - sum(parts)) < 1e-5 * sum(parts)


def test_launch_on_steers_launches_and_restores_on_error():
    """ops.launch_on: launches inside the block go to the given stream's raw handle, nested blocks restore the outer one,
    an exception leaves no override behind (the engine's side-stream blocks rely on it instead of torch.cuda.stream())."""
    from clslam_hip import ops

    class FakeStream:
        def __init__(self, h):
            self.cuda_stream = h
    t = torch.zeros(1)
    assert ops._FORCED_STREAM is None and ops._stream(t) == 0
    with ops.launch_on(FakeStream(11)):
        assert ops._stream(t) == 11
        with ops.launch_on(FakeStream(22)):
            assert ops._stream(t) == 22
        assert ops._stream(t) == 11
        with ops.launch_on(None):                 # no stream: a no-op block
            assert ops._stream(t) == 11
    assert ops._FORCED_STREAM is None
    with pytest.raises(ValueError):
        with ops.launch_on(FakeStream(33)):
            raise ValueError('boom')
    assert ops._FORCED_STREAM is None and ops._stream(t) == 0


@pytest.mark.parametrize('backend', BACKENDS)
def test_tie_break_noise_follows_torch_manual_seed(backend):
    """dpp.py:1055-1056 draws from torch's global generator: a torch.manual_seed() at any time re-seeds it.  The in-kernel
    Philox stream is keyed by the device generator's seed on every forward (not frozen at construction), restarts its
    draw counter on a re-seed, and data-parallel ranks (engine.noise_stream = shard offset) draw different fields from
    the same seed.  Two predictors seeded alike run bitwise the same adaptation step."""
    use_backend(backend)
    from clslam_hip import synth
    from predictor_util import make_predictor
    H, W, B = 64, 128, 1
    p = make_predictor(H, W, B)
    e = p.engine
    torch.manual_seed(5)
    a1, a2 = e._next_noise_draw(), e._next_noise_draw()
    assert a1[0] == a2[0] and a2[1] == a1[1] + 1
    if backend == 'hip':                                   # the draw counter is the device generator's Philox offset
        torch.manual_seed(5)
        assert e._next_noise_draw() == a1                  # re-seeding AFTER construction reproduces the stream
        state = torch.cuda.get_rng_state()
        nxt = e._next_noise_draw()
        torch.cuda.set_rng_state(state)
        assert e._next_noise_draw() == nxt
    torch.manual_seed(6)
    c1 = e._next_noise_draw()
    assert c1[0] != a1[0] and c1[1] == a1[1]
    e.noise_stream = 3
    torch.manual_seed(5)
    d1 = e._next_noise_draw()
    assert d1[0] == a1[0] and d1[1] != a1[1] and (d1[1] & 0xFFFFFFFFFF) == a1[1]
    e.noise_stream = 0
    batch = synth.make_batch(B, H, W, seed=12)
    res = []
    for seed in (7, 7, 8):
        q = make_predictor(H, W, B)
        torch.manual_seed(seed)
        _, losses = q.adapt(None, {k: v.clone() for k, v in batch.items()}, steps=2)
        res.append((q.engine.w.clone(), float(losses['loss'])))
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
