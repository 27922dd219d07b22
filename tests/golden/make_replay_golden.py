#!/usr/bin/env python
"""Golden vectors for the replay buffer's diversity bookkeeping and the loop-closure search, produced by the
REFERENCE's own code (slam/replay_buffer.py:82-184 `ReplayBuffer.add`, loop_closure_detection/
loop_closure_detection.py:41-76 `LoopClosureDetection.add/search`) imported from /root/reference in the build
container.  faiss is absent there: tests/golden/numpy_faiss.py (exact inner-product search by definition) stands in
for it; FeatureEncoder is replaced by a table lookup (the encoder forward has its own parity test).

    python tests/golden/make_replay_golden.py        # rewrites tests/golden/replay_lcd.npz

The fixture holds the seeded inputs' parameters and the reference's DECISIONS (which samples are stored after
every add, what a search returns) -- data, no source."""
import sys
import tempfile
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(OUT))


def feature_stream(n: int, d: int, seed: int, dup_every: int = 7) -> np.ndarray:
    """non-negative features (pooled post-ReLU activations, slam.py:146-147) from a handful of 'places' + noise;
    every dup_every-th sample is a near copy of an earlier one (exercises the similarity threshold)"""
    rng = np.random.default_rng(seed)
    places = rng.random((5, d)).astype(np.float32) ** 2
    x = np.empty((n, d), np.float32)
    for i in range(n):
        x[i] = places[rng.integers(5)] + 0.35 * rng.random(d).astype(np.float32)
        if i and i % dup_every == 0:
            x[i] = x[rng.integers(i)] + 0.01 * rng.random(d).astype(np.float32)
    return x


def lcd_stream(n: int, d: int, seed: int, revisit_from: int, revisit_gap: int) -> np.ndarray:
    """a trajectory that comes back: frame i >= revisit_from looks like frame i - revisit_gap"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    for i in range(1, n):                                  # neighbours look alike
        x[i] = 0.9 * x[i - 1] + 0.45 * x[i]
    for i in range(revisit_from, n):
        x[i] = x[i - revisit_gap] + 0.03 * rng.standard_normal(d).astype(np.float32)
    return x


def main() -> None:
    import numpy_faiss
    for name in ('cv2', 'wandb', 'colour_demosaicing', 'g2o', 'torchvision', 'torchvision.transforms',
                 'torchvision.transforms.functional', 'torchvision.models', 'torchvision.models.feature_extraction',
                 'matplotlib', 'matplotlib.pyplot', 'skimage', 'skimage.transform'):
        sys.modules[name] = MagicMock()
    sys.modules['faiss'] = numpy_faiss
    sys.path.insert(0, str(REF))
    for m in [k for k in sys.modules if k == 'datasets' or k.startswith('datasets.')]:
        del sys.modules[m]
    # the reference's FeatureEncoder builds a torchvision network; neither add() (features passed in) nor the
    # search bookkeeping needs it
    enc = types.ModuleType('loop_closure_detection.encoder')

    class FeatureEncoder:
        num_features = 576
        table = {}

        def __init__(self, device):
            self.device = device

        def __call__(self, image):
            return torch.from_numpy(self.table[int(image.reshape(-1)[0])])[None]
    enc.FeatureEncoder = FeatureEncoder
    sys.modules['loop_closure_detection.encoder'] = enc
    from loop_closure_detection.config import LoopClosureDetection as LcdConfig
    from loop_closure_detection.loop_closure_detection import LoopClosureDetection
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_replay_buffer', REF / 'slam' / 'replay_buffer.py')
    rb = importlib.util.module_from_spec(spec)             # slam/__init__ would pull in g2o-based classes
    spec.loader.exec_module(rb)

    out = {}
    # ---- replay buffer: maximize_diversity ---------------------------------------------------------------
    for tag, (n, d, cap, thr, seed) in {'a': (60, 512, 6, 0.95, 1), 'b': (40, 64, 3, 0.9, 2), 'c': (25, 512, 100, 0.95, 3)}.items():
        feats = feature_stream(n, d, seed)
        with tempfile.TemporaryDirectory() as tmp:
            buf = rb.ReplayBuffer(Path(tmp), 'Kitti', None, 64, 128, [0, 1, 2, 3], [0, -1, 1], batch_size=2,
                                  maximize_diversity=True, max_buffer_size=cap, similarity_threshold=thr,
                                  similarity_sampling=False)
            stored = np.full((n, cap + 1), -1, np.int64)
            for i in range(n):
                sample = {'index': torch.tensor([i])}
                buf.add(sample, {'index': i, 'images': ['a', 'b', 'c']}, feats[i:i + 1].copy())
                ids = np.sort(numpy_faiss.vector_to_array(buf.faiss_index.id_map))
                stored[i, :len(ids)] = ids
                on_disk = sorted(int(f.stem.split('_')[1]) for f in buf.online_filenames)
                assert on_disk == ids.tolist()
        out[f'rb_{tag}_params'] = np.array([n, d, cap, seed], np.int64)
        out[f'rb_{tag}_threshold'] = np.array(thr)
        out[f'rb_{tag}_stored'] = stored
    # ---- loop closure search ----------------------------------------------------------------------------------
    n, d = 700, 576
    feats = lcd_stream(n, d, 4, revisit_from=420, revisit_gap=380)
    FeatureEncoder.table = {i: feats[i] for i in range(n)}
    for tag, (thr, gap, nm) in {'a': (0.99, 250, 1), 'b': (0.97, 250, 3), 'c': (0.5, 10, 5)}.items():
        lcd = LoopClosureDetection(LcdConfig(Path('x.yaml'), thr, gap, nm))
        queries, res_ids, res_d = [], [], []
        for i in range(n):
            lcd.add(i + 1, torch.full((3, 2, 2), float(i)))           # image ids = step numbers (slam.py:219)
            if (i + 1) % 5 == 0:                                      # keyframe_frequency (slam.py:220)
                ids, dist = lcd.search(i + 1)
                dist = np.atleast_1d(np.asarray(dist, np.float32))
                queries.append(i + 1)
                res_ids.append(list(ids) + [-1] * (nm - len(ids)))
                res_d.append(list(dist) + [np.nan] * (nm - len(dist)))
        out[f'lcd_{tag}_cfg'] = np.array([thr, gap, nm], np.float64)
        out[f'lcd_{tag}_queries'] = np.array(queries, np.int64)
        out[f'lcd_{tag}_ids'] = np.array(res_ids, np.int64)
        out[f'lcd_{tag}_dist'] = np.array(res_d, np.float32)
        print(tag, 'queries with a match:', int((np.array(res_ids)[:, 0] >= 0).sum()), 'of', len(queries))
    out['lcd_params'] = np.array([n, d, 4, 420, 380], np.int64)
    np.savez_compressed(OUT / 'replay_lcd.npz', **out)
    print('wrote', OUT / 'replay_lcd.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
