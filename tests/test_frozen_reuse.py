"""adapt(steps=S) (dpp.py:309-313; config_adapt.yaml runs S=5) pushes the same minibatch S times through encoders
that are frozen and in eval mode (dpp.py:308): steps 2..S keep the encoder features and the identity-reprojection
maps of step 1 instead of recomputing them.  That must be invisible -- bitwise the same outputs, losses, weights and
Adam moments as recomputing everything -- and must never leak across adapt() calls or a weight reload."""
import pytest
import torch

from clslam_hip import synth
from emu_util import BACKENDS, use_backend
from predictor_util import make_predictor

H, W = 64, 128


def _size(backend):
    return (H, W) if backend == 'hip' else (64, 64)   # the emulated kernels are slow: keep the CPU suite short


def _run(B: int, reuse: bool, inject_noise: bool, H: int, W: int, plan=(3, 2)):
    p = make_predictor(H, W, B)
    p.engine.reuse_frozen_features = reuse
    if inject_noise:
        p.set_tie_break_noise(synth.make_noise(B, H, W, seed=9))
    res = []
    for call, steps in enumerate(plan):             # two calls with different minibatches
        batch = synth.make_batch(B, H, W, seed=70 + call)
        out, losses = p.adapt(None, batch, steps=steps)
        res += [out['disp', 0].clone(), out['cam_T_cam', 0, 1].clone(), out['rgb', -1, 2].clone(), losses['loss'].clone()]
    p.engine.wait_training()
    return res + [p.engine.w.clone(), p.engine.m.clone(), p.engine.v.clone()]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('B', [2, 3, 4])            # B=2 forces the hipGraph path on the GPU, B=3 the eager one, B=4 the eager one
                                                    # with the opt-in early coarse-scale loss schedule (engine.early_loss)
def test_frozen_feature_reuse_is_bitwise_invisible(backend, B, monkeypatch):
    if B == 2:
        monkeypatch.setenv('CLSLAM_HIPGRAPH', '1')   # opt-in path: its encoder-free second graph is covered here
    if B == 4:
        if backend != 'hip':
            pytest.skip('the early schedule needs the side streams')
        monkeypatch.setenv('CLSLAM_EARLY_LOSS', '1')
    if backend != 'hip' and B == 3:
        pytest.skip('eager path already covered by B=2 on the emulator')
    use_backend(backend)
    h, w = _size(backend)
    plan = (3, 2) if backend == 'hip' else (2, 1)
    n = B if backend == 'hip' else 1                 # emulator: one sample keeps the CPU suite short
    ref = _run(n, False, True, h, w, plan)
    got = _run(n, True, True, h, w, plan)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize('backend', BACKENDS)
def test_reuse_actually_skips_the_encoders(backend, monkeypatch):
    use_backend(backend)
    H, W = _size(backend)
    n = 3 if backend == 'hip' else 1
    p = make_predictor(H, W, n)
    calls = []
    orig = p.engine._encoder
    monkeypatch.setattr(p.engine, '_encoder', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(p.engine, 'graph_preferred', lambda B: False)
    p.adapt(None, synth.make_batch(n, H, W, seed=3), steps=3 if backend == 'hip' else 2)
    assert len(calls) == 2                           # depth + pose encoder, first step only
    p.adapt(None, synth.make_batch(n, H, W, seed=4), steps=1)
    assert len(calls) == 4                           # a new call never reuses
    p.predict(synth.make_batch(n, H, W, seed=5))
    assert len(calls) == 6


@pytest.mark.parametrize('backend', BACKENDS)
def test_reload_voids_held_features(backend):
    use_backend(backend)
    H, W = _size(backend)
    p = make_predictor(H, W, 1)
    batch = synth.make_batch(1, H, W, seed=3)
    p.adapt(None, batch, steps=1)
    ws = p.engine.workspace(1)
    assert ws.frozen_valid
    p.engine.sync_modules()
    with torch.no_grad():                            # edit an encoder weight from outside -> re-pack
        next(torch.nn.Module.parameters(p.models['depth_encoder'])).mul_(1.01)
    p.engine.pack_if_needed()
    assert not ws.frozen_valid


@pytest.mark.gpu
def test_hipgraph_replay_matches_eager_launches(monkeypatch):
    """The opt-in hipGraph path (CLSLAM_HIPGRAPH=1) replays the very launches of the eager path: bitwise equal."""
    use_backend('hip')
    monkeypatch.setenv('CLSLAM_HIPGRAPH', '0')
    ref = _run(2, True, True, H, W)
    monkeypatch.setenv('CLSLAM_HIPGRAPH', '1')
    got = _run(2, True, True, H, W)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i
