"""The replay buffer's per-frame sample building on the GPU (slam/replay_buffer.py:263-291, slam/slam.py:98):
the float-tensor colour jitter against the oracle's restatement of torchvision 0.11.1's functional_tensor.py (PARITY UNPINNED
against torchvision itself: source absent), and the drop-in `_get` against a fixture produced by the REFERENCE's own
``ReplayBuffer._get`` (tests/golden/make_replay_get_golden.py)."""
import pickle
import random
from pathlib import Path

import numpy as np
import pytest
import torch

from clslam_hip import ingest
from emu_util import BACKENDS, use_backend

GOLDEN = Path(__file__).resolve().parent / 'golden' / 'replay_get.npz'


@pytest.mark.parametrize('backend', BACKENDS)
def test_float_tensor_jitter_matches_the_torchvision_restatement(backend):
    """every op alone, all 24 orders of the four ops, gray pixels / saturated pixels / hue wrap-around.  BITWISE whenever no op
    follows `contrast`; otherwise <= 2e-6: the one step that is not bitwise is the contrast mean (torch.mean's summation order is
    not reproduced), and a hue round trip behind it amplifies that last-bit difference (h = (g - b) / (max - min))."""
    import itertools
    from oracle import jitter_tensor as jt
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(3)
    imgs = torch.rand(6, 3, 20, 36, generator=g)
    imgs[1, :, :5] = imgs[1, :1, :5]                 # gray rows: maxc == minc
    imgs[2, 0] = 1.0                                 # saturated red: hue 0 / wrap
    imgs[3] = (imgs[3] * 255).round() / 255          # exact byte fractions like ToTensor produces
    imgs[4] = 0.0
    cases = [([op], [1.15, 0.85, 1.2, -0.1]) for op in range(4)] + [([op], [0.8, 1.2, 0.8, 0.1]) for op in range(4)]
    cases += [(list(perm), [0.93, 1.07, 1.13, 0.037]) for perm in itertools.permutations(range(4))]
    cases += [([], [1, 1, 1, 0]), ([3], [1, 1, 1, -0.5]), ([3], [1, 1, 1, 0.5]), ([1, 3], [1, 0.8, 1, 0.0])]
    worst = 0.0
    for order, factors in cases:
        ref = torch.cat([jt.color_jitter(imgs[i:i + 1], order, factors) for i in range(len(imgs))])   # per-image contrast mean
        got = ingest.color_jitter_tensor(imgs.to(dev), ingest.jitter_params([(order, factors)] * len(imgs), dev)).cpu()
        err = float((got - ref).abs().max())
        worst = max(worst, err)
        if 1 not in order or order[-1] == 1:
            assert torch.equal(got, ref) or err <= 6e-8, (order, err)     # (contrast last: one rounding of the blend away)
        assert err <= 2e-6, (order, factors, err)
    # different draws per image in ONE launch
    rng = random.Random(5)
    draws = [ingest.draw_color_jitter(rng=rng) for _ in range(len(imgs))]
    ref = torch.cat([jt.color_jitter(imgs[i:i + 1], *draws[i]) for i in range(len(imgs))])
    got = ingest.color_jitter_tensor(imgs.to(dev), ingest.jitter_params(draws, dev)).cpu()
    assert float((got - ref).abs().max()) <= 2e-6
    # the draws consume Python's random stream like datasets/utils.py:236-259 (four uniforms, then the shuffle)
    a, b = random.Random(9), random.Random(9)
    order, factors = ingest.draw_color_jitter(rng=a)
    assert factors == [b.uniform(0.8, 1.2), b.uniform(0.8, 1.2), b.uniform(0.8, 1.2), b.uniform(-.1, .1)]
    ops = [0, 1, 2, 3]
    b.shuffle(ops)
    assert order == ops and a.random() == b.random()


@pytest.mark.parametrize('backend', BACKENDS)
def test_replay_get_drop_in_matches_the_reference_get(backend, tmp_path):
    """ReplaySampleBuilder.get_one / get_many against the dict the REFERENCE's ReplayBuffer._get built from the same PNG files and
    the same `random` seed (fixture: raw frames, seed, every returned tensor): 'rgb' levels bit-exact (Pillow's LANCZOS chain),
    'rgb_aug' <= 2e-6, every other entry passed through, keys and shapes identical."""
    from PIL import Image
    dev = use_backend(backend)
    z = np.load(GOLDEN, allow_pickle=False)
    H, W, scales, frames = int(z['height']), int(z['width']), [int(s) for s in z['scales']], [int(f) for f in z['frames']]
    files = []
    for i in range(int(z['n_samples'])):
        sample = {('camera_matrix', 0): torch.from_numpy(z['camera_matrix']).clone(), ('index',): torch.tensor([i])}
        for f in frames:
            png = tmp_path / f's{i}_f{f}.png'
            Image.fromarray(z[f'raw_{i}_{f}']).save(png)
            sample['rgb', f] = png
        fn = tmp_path / f'kitti_{i:05}.pkl'
        with open(fn, 'wb') as fh:
            pickle.dump(sample, fh)
        files.append(fn)
    build = ingest.ReplaySampleBuilder(H, W, scales, frames, device=dev)
    random.seed(int(z['seed']))
    one_by_one = [build.get_one(fn) for fn in files]
    random.seed(int(z['seed']))
    batched = build.get_many(files)
    for i, (a, b) in enumerate(zip(one_by_one, batched)):
        want_keys = {tuple(k.split('|')[1:]) for k in z.files if k.startswith(f'out{i}|')}
        got_keys = {tuple(str(p) for p in k) for k in a}
        assert got_keys == want_keys == {tuple(str(p) for p in k) for k in b}
        for k in a:
            ref = torch.from_numpy(z['|'.join([f'out{i}'] + [str(p) for p in k])])
            for got in (a[k], b[k]):
                assert tuple(got.shape) == tuple(ref.shape), k
                if k[0] == 'rgb':
                    assert torch.equal(got.cpu(), ref), k
                elif k[0] == 'rgb_aug':
                    assert float((got.cpu() - ref).abs().max()) <= 2e-6, (k, float((got.cpu() - ref).abs().max()))
                else:
                    assert torch.equal(got.cpu(), ref), k
        assert a['rgb_aug', 0, 0].device.type == dev.type
    # slam.py:300-309 on a GPU-resident replay dict
    online = {k: v.cpu() for k, v in one_by_one[0].items()}
    cat = ingest.cat_dict(online, batched[1])
    assert cat['rgb', 0, 0].shape[0] == 2 and cat['rgb', 0, 0].device.type == dev.type


def test_jitter_pin_script_is_honest():
    """tests/golden/make_replay_jitter_golden.py pins oracle/jitter_tensor.py against the real torchvision the first time a build
    container has it; until then it must say PARITY UNPINNED (exit 3) -- and once tests/golden/replay_jitter.npz exists the
    restatement is held to torchvision's recorded outputs."""
    import subprocess
    import sys
    gold = GOLDEN.parent / 'replay_jitter.npz'
    try:
        import torchvision  # noqa: F401
        have_tv = True
    except Exception:  # noqa: BLE001
        have_tv = False
    if not have_tv:
        r = subprocess.run([sys.executable, str(GOLDEN.parent / 'make_replay_jitter_golden.py')], capture_output=True, text=True, timeout=300)
        assert r.returncode == 3 and 'PARITY UNPINNED' in r.stdout, r.stdout + r.stderr
    if gold.exists():
        from oracle import jitter_tensor as jt
        sys.path.insert(0, str(GOLDEN.parent))
        import make_replay_jitter_golden as mk
        z = np.load(gold)
        imgs = mk.images()
        for order, factors, ref in zip(z['orders'], z['factors'], z['outputs']):
            order = [int(o) for o in order if o >= 0]
            got = torch.cat([jt.color_jitter(imgs[i:i + 1], order, list(factors)) for i in range(len(imgs))])
            assert float((got - torch.from_numpy(ref)).abs().max()) <= 1e-6, (order, factors)


class _BufferShapedLikeTheReference:
    """The three members of slam/replay_buffer.py's ReplayBuffer that install() touches, with the reference's calling pattern:
    `get` chooses files, calls `self._get(filename)` once per file and concatenates the dicts key by key (replay_buffer.py:226-233)."""

    def __init__(self, files, host_get):
        self.files, self._host_get, self.calls = files, host_get, 0

    def _get(self, filename, include_batch=True):
        self.calls += 1
        return self._host_get(filename)

    def get(self, sample, image_features=None):
        data = self._get(self.files[0])
        for fn in self.files[1:]:
            nxt = self._get(fn)
            for key in data:
                data[key] = torch.cat([data[key], nxt[key]])
        return data


class _SlamShapedLikeTheReference:
    @staticmethod
    def _cat_dict(a, b):                 # slam/slam.py:300-309
        return {k: torch.cat([a[k], b[k]]) for k in a if k in b}


@pytest.mark.parametrize('backend', BACKENDS)
def test_install_switches_a_replay_buffer_and_a_slam_object_over(backend, tmp_path):
    """install() on objects with the reference's calling pattern (the reference itself runs in tests/test_reference_callers.py, build
    container only; this twin also runs on the MI355X): `get` builds the K samples in ONE get_many() -- bitwise what get_many()
    returns for those files with the same `random` state --, the object's own `_get` is never entered, `_cat_dict` joins a host
    sample with the device minibatch, and the classes are untouched (another instance still runs its own code)."""
    from PIL import Image
    dev = use_backend(backend)
    z = np.load(GOLDEN, allow_pickle=False)
    H, W, scales, frames = int(z['height']), int(z['width']), [int(s) for s in z['scales']], [int(f) for f in z['frames']]
    files = []
    for i in range(int(z['n_samples'])):
        sample = {('camera_matrix', 0): torch.from_numpy(z['camera_matrix']).clone(), ('index',): torch.tensor([i])}
        for f in frames:
            png = tmp_path / f's{i}_f{f}.png'
            Image.fromarray(z[f'raw_{i}_{f}']).save(png)
            sample['rgb', f] = png
        fn = tmp_path / f'kitti_{i:05}.pkl'
        with open(fn, 'wb') as fh:
            pickle.dump(sample, fh)
        files.append(fn)
    build = ingest.ReplaySampleBuilder(H, W, scales, frames, device=dev, decode_threads=4)
    random.seed(11)
    want = build.get_many(files)
    want = {k: torch.cat([d[k] for d in want]) for k in want[0]}
    buf, slam = _BufferShapedLikeTheReference(files, host_get=None), _SlamShapedLikeTheReference()
    assert build.install(buf, slam) is build
    random.seed(11)
    got = buf.get({'index': torch.tensor([0])})
    assert buf.calls == 0 and set(got) == set(want)
    for k in want:
        assert torch.equal(got[k].cpu(), want[k].cpu()) and got[k].device == want[k].device, k
    assert got['rgb_aug', 0, 0].device.type == dev.type and got['rgb_aug', 0, 0].shape[0] == len(files)
    online = {k: v[:1].cpu() for k, v in got.items()}
    joined = slam._cat_dict(online, got)
    assert joined['rgb', 0, 0].shape[0] == len(files) + 1 and joined['rgb', 0, 0].device.type == dev.type
    assert torch.equal(joined['rgb', 0, 0][1:].cpu(), got['rgb', 0, 0].cpu())
    # direct callers of _get get the builder's, too; the CLASS keeps its own code
    one = buf._get(files[0])
    assert one['rgb', 0, 0].device.type == dev.type and buf.calls == 0
    other = _BufferShapedLikeTheReference(files[:1], host_get=lambda fn: {'x': torch.zeros(1)})
    assert set(other.get(None)) == {'x'} and other.calls == 1
