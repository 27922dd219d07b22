"""FlatIPIndex against the numpy restatement of faiss 'Flat' inner-product search:
scores to fp32 round-off, identical ranking wherever scores are distinct, faiss's padding, ids, removal."""
import numpy as np
import pytest

from clslam_hip.flat_index import FLT_MAX, FlatIPIndex, normalize_L2
from emu_util import BACKENDS, use_backend
from oracle import flat_index as OI


def _unit(n, d, seed):
    x = np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)
    normalize_L2(x)
    return x


def _check(index, db, ids, q, k):
    D, I = index.search(q, k)
    De, Ie = OI.search(db, ids, q, k)
    assert D.shape == (len(q), k) and I.shape == (len(q), k) and D.dtype == np.float32 and I.dtype == np.int64
    assert np.array_equal(I == -1, Ie == -1)
    real = Ie != -1
    assert np.allclose(D[real], De[real], rtol=0, atol=2e-6)
    assert np.all(D[~real] == -FLT_MAX)
    # ranking: wherever the GPU picked another id, its true score ties the expected one at fp32 round-off
    diff = np.argwhere((I != Ie) & real)
    lookup = {int(v): j for j, v in enumerate(ids)}
    for qi, r in diff:
        s_got = float(db[lookup[int(I[qi, r])]].astype(np.float64) @ q[qi].astype(np.float64))
        assert abs(s_got - float(De[qi, r])) < 2e-6, (qi, r)
    for qi in range(len(q)):                                   # no id twice
        got = I[qi][I[qi] != -1]
        assert len(set(got.tolist())) == len(got)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('n,d,k,nq', [(0, 576, 5, 2), (1, 576, 3, 1), (37, 512, 1, 3), (100, 512, 100, 100),
                                      (300, 10, 100, 2), (4096, 64, 100, 1), (4097, 64, 100, 2), (9000, 576, 100, 1)])
def test_search_matches_restatement(backend, n, d, k, nq):
    if backend != 'hip' and n * d > 300000:
        pytest.skip('large case runs on the GPU only')
    use_backend(backend)
    db, q = _unit(n, d, 1), _unit(nq, d, 2)
    index = FlatIPIndex(d)
    for lo in range(0, n, 1000):                               # grows its storage as it goes
        index.add(db[lo:lo + 1000])
    assert index.ntotal == n
    _check(index, db, np.arange(n), q, k)


@pytest.mark.parametrize('backend', BACKENDS)
def test_reference_usage_patterns(backend):
    """The calls loop_closure_detection.py:44-76 and replay_buffer.py:96-152 make."""
    use_backend(backend)
    d = 64
    feats = _unit(120, d, 3)
    # loop closure: add one frame at a time, query a stored frame, drop itself and temporal neighbours
    lcd = FlatIPIndex(d)
    for f in feats:
        lcd.add(f[None])
    assert np.allclose(lcd.reconstruct(17), feats[17])
    D, I = lcd.search(lcd.reconstruct(17)[None], 100)
    assert I[0, 0] == 17 and abs(D[0, 0] - 1.0) < 1e-5 and np.all(np.diff(D[0]) <= 0)
    # replay buffer: ids, nearest-neighbour similarity, full similarity matrix, removal
    buf = FlatIPIndex(d)
    ids = np.arange(1000, 1120)
    buf.add_with_ids(feats, ids)
    assert np.array_equal(buf.ids, ids)
    _check(buf, feats, ids, feats[:5], 1)
    Dm, Im = buf.search(buf.reconstruct_n(0, buf.ntotal), buf.ntotal)
    assert np.array_equal(np.sort(Im, 1), np.tile(ids, (120, 1)))
    assert buf.remove_ids(np.array([1003, 1119])) == 2 and buf.ntotal == 118
    keep = ~np.isin(ids, [1003, 1119])
    assert np.array_equal(buf.ids, ids[keep])
    _check(buf, feats[keep], ids[keep], feats[[3, 50]], 10)
    # duplicates: equal scores come back in insertion order
    dup = FlatIPIndex(d)
    dup.add(np.repeat(feats[:1], 5, 0))
    D, I = dup.search(feats[:1], 8)
    assert I[0].tolist() == [0, 1, 2, 3, 4, -1, -1, -1]
