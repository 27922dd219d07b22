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


def test_lcd_pin_script_is_honest():
    """tests/golden/make_lcd_golden.py pins oracle/mobilenet.py against the real torchvision the first time a build
    container has it; until then it must say PARITY UNPINNED (exit 3) -- and once tests/golden/lcd_features.npz exists
    the oracle is held to it."""
    import subprocess
    import sys
    from pathlib import Path
    import numpy as np
    gold = Path(__file__).parent / 'golden' / 'lcd_features.npz'
    try:
        import torchvision  # noqa: F401
        have_tv = True
    except Exception:  # noqa: BLE001
        have_tv = False
    if not have_tv:
        r = subprocess.run([sys.executable, str(gold.parent / 'make_lcd_golden.py')], capture_output=True, text=True, timeout=300)
        assert r.returncode == 3 and 'PARITY UNPINNED' in r.stdout, r.stdout + r.stderr
    if gold.exists():                                    # a container with torchvision + weights has pinned it
        import os
        wfile = os.environ.get('CLSLAM_MOBILENETV3_WEIGHTS')
        if not wfile or not Path(wfile).exists():
            pytest.skip('golden features exist but the ImageNet checkpoint is not on this machine')
        from oracle.mobilenet import MobileNetV3SmallFeatures, feature_encoder
        real = torch.load(wfile, map_location='cpu')
        m = MobileNetV3SmallFeatures()
        m.load_state_dict({k: v for k, v in real.items() if k.startswith('features.')})
        g = np.load(gold)
        for i in range(2):
            n, H, W, seed = (int(v) for v in g[f'params_{i}'])
            img = synth.make_batch(n, H, W, seed=seed)['rgb', 1, 0]
            assert rel_err(feature_encoder(m.eval(), img), torch.from_numpy(g[f'features_{i}'])) < 1e-6
