"""SURVEY.md 8(f) rank 4 -- the replay buffer's diversity bookkeeping and the loop-closure search on the MI355X:
  * oracle/replay_lcd.py (numpy restatement) reproduces the decisions of the REFERENCE's own code recorded in
    tests/golden/replay_lcd.npz (tests/golden/make_replay_golden.py);
  * clslam_hip.diversity.DiversityBuffer, the `faiss`-named shim and loop_closure_detection.LoopClosureDetection
    (HIP kernels; CPU emulator here, the GPU with -m gpu) reproduce the same decisions."""
import pickle
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from emu_util import BACKENDS, use_backend

GOLD = Path(__file__).parent / 'golden' / 'replay_lcd.npz'
sys.path.insert(0, str(GOLD.parent))
from make_replay_golden import feature_stream, lcd_stream  # noqa: E402  (seeded input generators only)


def _gold():
    return np.load(GOLD)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_replay_diversity_matches_reference(tag):
    from oracle.replay_lcd import ReplayDiversity
    g = _gold()
    n, d, cap, seed = (int(v) for v in g[f'rb_{tag}_params'])
    feats = feature_stream(n, d, seed)
    buf = ReplayDiversity(cap, float(g[f'rb_{tag}_threshold']))
    for i in range(n):
        buf.add(feats[i], i)
        want = g[f'rb_{tag}_stored'][i]
        assert sorted(buf.ids.tolist()) == want[want >= 0].tolist(), i


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_loop_closure_matches_reference(tag):
    from oracle.replay_lcd import LoopClosure
    g = _gold()
    n, d, seed, rfrom, rgap = (int(v) for v in g['lcd_params'])
    feats = lcd_stream(n, d, seed, rfrom, rgap)
    thr, gap, nm = g[f'lcd_{tag}_cfg']
    lcd = LoopClosure(float(thr), int(gap), int(nm))
    qi = 0
    for i in range(n):
        lcd.add(i + 1, feats[i])
        if (i + 1) % 5 == 0:
            ids, dist = lcd.search(i + 1)
            want = g[f'lcd_{tag}_ids'][qi]
            assert ids == want[want >= 0].tolist(), i
            assert np.allclose(dist, g[f'lcd_{tag}_dist'][qi][:len(dist)], atol=2e-6)
            qi += 1


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_diversity_buffer_matches_reference(backend, tag):
    dev = use_backend(backend)
    from clslam_hip.diversity import DiversityBuffer
    from oracle.replay_lcd import ReplayDiversity
    g = _gold()
    n, d, cap, seed = (int(v) for v in g[f'rb_{tag}_params'])
    thr = float(g[f'rb_{tag}_threshold'])
    feats = feature_stream(n, d, seed)
    buf, ref = DiversityBuffer(d, cap, thr, dev), ReplayDiversity(cap, thr)
    for i in range(n):
        f = torch.from_numpy(feats[i:i + 1]).to(dev) if i % 2 else feats[i:i + 1]   # device tensors and numpy rows
        added, removed, sim = buf.add(f, i)
        r_added, r_removed, r_sim = ref.add(feats[i], i)
        assert (added, removed) == (r_added, r_removed), i
        assert abs(sim - r_sim) < 2e-6
        want = g[f'rb_{tag}_stored'][i]
        assert sorted(buf.ids.tolist()) == want[want >= 0].tolist(), i
        assert buf.ntotal <= cap
    # similarity matrix in slot order == the reference's distance_matrix wherever both slots are occupied
    if ref.dist is not None:
        S, ids = buf.similarity_matrix()
        assert ids == [int(v) for v in ref.dist_ids]
        occ = np.array(ids) >= 0
        assert np.allclose(S[np.ix_(occ, occ)], ref.dist[np.ix_(occ, occ)], atol=2e-6)
    sims, sids = buf.similarities(feats[0])
    assert sorted(sids.tolist()) == sorted(buf.ids.tolist())
    lookup = {int(i): OIrow for i, OIrow in zip(ref.ids, ref.x)}
    q = feats[0] / np.linalg.norm(feats[0])
    assert np.allclose(sims, [float(lookup[int(i)] @ q) for i in sids], atol=2e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_faiss_shim_runs_the_reference_call_sequence(backend):
    """the call sequence of slam/replay_buffer.py:95-160 + :237-250 against the `faiss`-named module"""
    use_backend(backend)
    import faiss
    d, n = 64, 9
    feats = feature_stream(n, d, 5)
    index = faiss.IndexIDMap(faiss.index_factory(d, 'Flat', faiss.METRIC_INNER_PRODUCT))
    assert index.ntotal == 0
    for i in range(n):
        f = feats[i:i + 1].copy()
        faiss.normalize_L2(f)
        assert abs(float((f * f).sum()) - 1) < 1e-5
        if index.ntotal:
            sim = index.search(f, 1)[0][0][0]
            assert -1.0001 <= sim <= 1.0001
        index.add_with_ids(f, np.array([100 + i]))
    assert faiss.vector_to_array(index.id_map).tolist() == list(range(100, 100 + n))
    x = index.index.reconstruct_n(0, index.ntotal)
    D, I = index.search(x, index.ntotal)
    assert D.shape == (n, n) and sorted(I[0].tolist()) == list(range(100, 100 + n))
    assert np.allclose(D[:, 0], 1, atol=1e-5) and (I[:, 0] == np.arange(100, 100 + n)).all()
    assert index.remove_ids(np.array([103])) == 1
    assert faiss.vector_to_array(index.id_map).tolist() == [100, 101, 102, 104, 105, 106, 107, 108]
    assert np.allclose(index.index.reconstruct(3), x[4])
    with pytest.raises(RuntimeError):
        index.add(x[:1])
    with pytest.raises(NotImplementedError):
        faiss.index_factory(d, 'IVF16,Flat', faiss.METRIC_INNER_PRODUCT)
    # save_state / load_state pickle the index (replay_buffer.py:237-250)
    clone = pickle.loads(pickle.dumps({'faiss_index': index}))['faiss_index']
    assert faiss.vector_to_array(clone.id_map).max() == 108
    D2, I2 = clone.search(x[:2], 3)
    D1, I1 = index.search(x[:2], 3)
    assert np.array_equal(I1, I2) and np.allclose(D1, D2)
    flat = faiss.index_factory(d, 'Flat', faiss.METRIC_INNER_PRODUCT)     # loop_closure_detection.py:35-57
    flat.add(x[:5])
    assert flat.ntotal == 5 and flat.is_trained
    D, I = flat.search(np.expand_dims(flat.reconstruct(2), 0), 100)
    assert I.shape == (1, 100) and I[0, 0] == 2 and (I[0, 5:] == -1).all()


def _lcd_weights():
    from test_lcd_encoder import _weights
    return _weights()[1]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_loop_closure_detection_matches_reference(backend, tag, monkeypatch, tmp_path):
    """LoopClosureDetection (device-resident features + FlatIPIndex) takes the reference's decisions on the golden
    stream.  The encoder forward is replaced by the stream's features here (it has its own parity test); one real
    encoder call checks the plumbing."""
    dev = use_backend(backend)
    wfile = tmp_path / 'mbv3.pth'
    torch.save(_lcd_weights(), wfile)
    monkeypatch.setenv('CLSLAM_MOBILENETV3_WEIGHTS', str(wfile))
    from loop_closure_detection import Config, LoopClosureDetection
    g = _gold()
    n, d, seed, rfrom, rgap = (int(v) for v in g['lcd_params'])
    feats = lcd_stream(n, d, seed, rfrom, rgap)
    thr, gap, nm = g[f'lcd_{tag}_cfg']
    lcd = LoopClosureDetection(Config(Path('x.yaml'), float(thr), int(gap), int(nm)))
    assert lcd.device.type == dev.type and lcd.model.num_features == 576
    real_model = lcd.model
    table = torch.from_numpy(feats).to(dev)

    class Lookup:
        num_features = 576

        def __call__(self, image):
            return table[int(image.reshape(-1)[0])][None].clone()
    lcd.model = Lookup()
    qi = 0
    for i in range(n):
        lcd.add(i + 1, torch.full((3, 2, 2), float(i)))
        if (i + 1) % 5 == 0:
            ids, dist = lcd.search(i + 1)
            want = g[f'lcd_{tag}_ids'][qi]
            assert ids == want[want >= 0].tolist(), i
            assert np.allclose(dist, g[f'lcd_{tag}_dist'][qi][:len(dist)], atol=3e-6)
            qi += 1
    assert lcd.faiss_index.ntotal == n
    # plumbing with the real encoder: add two frames, cosine of a frame with itself is 1
    lcd.model = real_model
    from clslam_hip import synth
    img = synth.make_batch(2, 64, 128, seed=6)['rgb', 1, 0]
    lcd2 = LoopClosureDetection(Config(Path('x.yaml'), 0.5, 0, 1))
    lcd2.add(1, img[0])
    lcd2.add(2, img[1])
    assert abs(lcd2.predict(img[0], img[0]) - 1) < 1e-5
    ids, dist = lcd2.search(2)
    assert ids in ([1], []) and len(dist) == len(ids)
