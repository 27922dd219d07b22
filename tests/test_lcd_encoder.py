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


# torchvision 0.11 `mobilenet_v3_small().features` -- every convolution / squeeze-excitation tensor with its shape, WRITTEN OUT AS
# DATA (from the published architecture; torchvision itself is not installable here).  A BatchNorm2d(eps=1e-3,
# momentum=0.01) with weight / bias / running_mean / running_var / num_batches_tracked follows every `...N.0.weight` convolution as
# `...N.1.*`.  What is pinned: the 927,008 feature parameters (= torchvision's 2,542,856 minus the unused classifier: 576*1024 +
# 1024 + 1024*1000 + 1000), the key list a real checkpoint must load into, BN eps, the make_divisible(exp // 4, 8) squeeze widths
# (8, 24, 64, 64, 32, 40, 72, 144, 144).  What stays UNPINNED: the arithmetic against real torchvision with ImageNet weights.
TORCHVISION_FEATURE_TENSORS = [
    ('features.0.0.weight', (16, 3, 3, 3)), ('features.1.block.0.0.weight', (16, 1, 3, 3)),
    ('features.1.block.1.fc1.weight', (8, 16, 1, 1)), ('features.1.block.1.fc1.bias', (8,)),
    ('features.1.block.1.fc2.weight', (16, 8, 1, 1)), ('features.1.block.1.fc2.bias', (16,)),
    ('features.1.block.2.0.weight', (16, 16, 1, 1)), ('features.2.block.0.0.weight', (72, 16, 1, 1)),
    ('features.2.block.1.0.weight', (72, 1, 3, 3)), ('features.2.block.2.0.weight', (24, 72, 1, 1)),
    ('features.3.block.0.0.weight', (88, 24, 1, 1)), ('features.3.block.1.0.weight', (88, 1, 3, 3)),
    ('features.3.block.2.0.weight', (24, 88, 1, 1)), ('features.4.block.0.0.weight', (96, 24, 1, 1)),
    ('features.4.block.1.0.weight', (96, 1, 5, 5)), ('features.4.block.2.fc1.weight', (24, 96, 1, 1)),
    ('features.4.block.2.fc1.bias', (24,)), ('features.4.block.2.fc2.weight', (96, 24, 1, 1)),
    ('features.4.block.2.fc2.bias', (96,)), ('features.4.block.3.0.weight', (40, 96, 1, 1)), ('features.5.block.0.0.weight',
    (240, 40, 1, 1)), ('features.5.block.1.0.weight', (240, 1, 5, 5)), ('features.5.block.2.fc1.weight', (64, 240, 1, 1)),
    ('features.5.block.2.fc1.bias', (64,)), ('features.5.block.2.fc2.weight', (240, 64, 1, 1)),
    ('features.5.block.2.fc2.bias', (240,)), ('features.5.block.3.0.weight', (40, 240, 1, 1)),
    ('features.6.block.0.0.weight', (240, 40, 1, 1)), ('features.6.block.1.0.weight', (240, 1, 5, 5)),
    ('features.6.block.2.fc1.weight', (64, 240, 1, 1)), ('features.6.block.2.fc1.bias', (64,)),
    ('features.6.block.2.fc2.weight', (240, 64, 1, 1)), ('features.6.block.2.fc2.bias', (240,)),
    ('features.6.block.3.0.weight', (40, 240, 1, 1)), ('features.7.block.0.0.weight', (120, 40, 1, 1)),
    ('features.7.block.1.0.weight', (120, 1, 5, 5)), ('features.7.block.2.fc1.weight', (32, 120, 1, 1)),
    ('features.7.block.2.fc1.bias', (32,)), ('features.7.block.2.fc2.weight', (120, 32, 1, 1)),
    ('features.7.block.2.fc2.bias', (120,)), ('features.7.block.3.0.weight', (48, 120, 1, 1)),
    ('features.8.block.0.0.weight', (144, 48, 1, 1)), ('features.8.block.1.0.weight', (144, 1, 5, 5)),
    ('features.8.block.2.fc1.weight', (40, 144, 1, 1)), ('features.8.block.2.fc1.bias', (40,)),
    ('features.8.block.2.fc2.weight', (144, 40, 1, 1)), ('features.8.block.2.fc2.bias', (144,)),
    ('features.8.block.3.0.weight', (48, 144, 1, 1)), ('features.9.block.0.0.weight', (288, 48, 1, 1)),
    ('features.9.block.1.0.weight', (288, 1, 5, 5)), ('features.9.block.2.fc1.weight', (72, 288, 1, 1)),
    ('features.9.block.2.fc1.bias', (72,)), ('features.9.block.2.fc2.weight', (288, 72, 1, 1)),
    ('features.9.block.2.fc2.bias', (288,)), ('features.9.block.3.0.weight', (96, 288, 1, 1)),
    ('features.10.block.0.0.weight', (576, 96, 1, 1)), ('features.10.block.1.0.weight', (576, 1, 5, 5)),
    ('features.10.block.2.fc1.weight', (144, 576, 1, 1)), ('features.10.block.2.fc1.bias', (144,)),
    ('features.10.block.2.fc2.weight', (576, 144, 1, 1)), ('features.10.block.2.fc2.bias', (576,)),
    ('features.10.block.3.0.weight', (96, 576, 1, 1)), ('features.11.block.0.0.weight', (576, 96, 1, 1)),
    ('features.11.block.1.0.weight', (576, 1, 5, 5)), ('features.11.block.2.fc1.weight', (144, 576, 1, 1)),
    ('features.11.block.2.fc1.bias', (144,)), ('features.11.block.2.fc2.weight', (576, 144, 1, 1)),
    ('features.11.block.2.fc2.bias', (576,)), ('features.11.block.3.0.weight', (96, 576, 1, 1)), ('features.12.0.weight',
    (576, 96, 1, 1))
]


def test_lcd_encoder_shape_facts_of_torchvision_are_pinned():
    from oracle.mobilenet import BN, MobileNetV3SmallFeatures, make_divisible
    m = MobileNetV3SmallFeatures()
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 927_008 == 2_542_856 - (576 * 1024 + 1024 + 1024 * 1000 + 1000)
    want = []
    for k, shape in TORCHVISION_FEATURE_TENSORS:
        want.append((k, shape))
        if k.endswith('.0.weight'):          # Conv2dNormActivation: the BatchNorm that follows
            c = shape[0]
            want += [(k[:-len('0.weight')] + '1.' + n, (c,)) for n in ('weight', 'bias', 'running_mean', 'running_var')]
            want.append((k[:-len('0.weight')] + '1.num_batches_tracked', ()))
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == want
    assert BN(8).eps == 1e-3 and BN(8).momentum == 0.01
    se = [shape[0] for k, shape in TORCHVISION_FEATURE_TENSORS if k.endswith('fc1.weight')]
    assert se == [8, 24, 64, 64, 32, 40, 72, 144, 144] == [make_divisible(e // 4, 8) for e in (16, 96, 240, 240, 120, 144, 288, 576, 576)]
    # the product's encoder validates a state dict against exactly this key set
    from clslam_hip import lcd
    assert set(lcd.synthetic_state_dict()) == {k for k in sd if not k.endswith('num_batches_tracked')}


def test_lcd_golden_generator_runs_whenever_torchvision_is_importable(tmp_path):
    """tests/golden/make_lcd_golden.py pins the restatement against the REAL torchvision model the moment one is importable (a
    maintainer's machine, a later image); here it must at least say so instead of silently producing nothing."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / 'golden' / 'make_lcd_golden.py'
    have = importlib.util.find_spec('torchvision') is not None
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True)
    if have:       # architecture pin (same closed-form state dict through torchvision's network and the restatement)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    else:
        assert r.returncode == 3 and 'PARITY UNPINNED' in r.stdout
