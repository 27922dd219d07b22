"""Loop-closure feature encoder (MobileNetV3-small forward) on the HIP kernels vs the oracle's torch
restatement (parity with torchvision itself is UNPINNED, see oracle/mobilenet.py)."""
import pytest
import torch

from clslam_hip import synth
from emu_util import BACKENDS, use_backend
from helpers import rel_err


def _weights(seed=0):
    from oracle.mobilenet import MobileNetV3SmallFeatures
    m = MobileNetV3SmallFeatures()
    sd = m.state_dict()
    new = {}
    for k, v in sd.items():
        u = torch.from_numpy(synth.hash_uniform(max(v.numel(), 1), synth._key_seed('lcd/' + k, seed))[:v.numel()]).reshape(v.shape)
        if k.endswith('num_batches_tracked'):
            new[k] = v.clone()
        elif v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            new[k] = (u * 2 - 1) * (6.0 / fan_in) ** 0.5
        elif k.endswith('running_var'):
            new[k] = 0.6 + 0.8 * u
        elif k.endswith('running_mean'):
            new[k] = 0.2 * (u - 0.5)
        elif '.1.weight' in k:
            new[k] = 0.8 + 0.4 * u
        else:
            new[k] = 0.2 * (u - 0.5)
    m.load_state_dict(new)
    return m, new


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('B,H,W', [(1, 64, 128), (2, 48, 80)])
def test_feature_encoder_matches_oracle(backend, B, H, W):
    dev = use_backend(backend)
    from loop_closure_detection import FeatureEncoder
    from oracle.mobilenet import feature_encoder
    model, sd = _weights()
    img = synth.make_batch(B, H, W, seed=6)['rgb', 1, 0]
    ref = feature_encoder(model, img)
    enc = FeatureEncoder(dev, weights=sd)
    assert enc.num_features == 576
    got = enc(img)
    assert got.shape == (B, 576)
    assert rel_err(got.cpu(), ref) < 1e-4, rel_err(got.cpu(), ref)
    # 3-D input as slam.py passes it, and state-dict validation
    assert rel_err(enc(img[0]).cpu(), ref[:1]) < 1e-4
    from clslam_hip._lib import ClslamError
    with pytest.raises(ClslamError):
        FeatureEncoder(dev, weights={k: v for k, v in sd.items() if 'features.3.' not in k})


@pytest.mark.gpu
def test_feature_encoder_latency_on_gpu(capsys):
    """One LCD descriptor per frame (slam.py:180,223): report and bound the latency at both resolutions of
    BASELINE.json (B=1; launch-latency-bound: 52 kernel launches, 0.28 GFLOP at 192x640)."""
    import time
    dev = use_backend('hip')
    from loop_closure_detection import FeatureEncoder
    _, sd = _weights()
    enc = FeatureEncoder(dev, weights=sd)
    for H, W in ((192, 640), (384, 1280)):
        img = synth.make_batch(1, H, W, seed=6)['rgb', 1, 0].to(dev)
        for _ in range(3):
            enc(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            enc(img)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        with capsys.disabled():
            print(f'\n[lcd] MobileNetV3-small features {H}x{W} B=1: {ms:.3f} ms per call')
        assert ms < 10.0
