"""TEST INFRASTRUCTURE: a numpy stand-in for the subset of faiss the reference's replay buffer / loop-closure code
calls, built on oracle.flat_index (the definition of exact inner-product search).  The golden generators install
it as ``sys.modules['faiss']`` so that the REFERENCE's own bookkeeping code (slam/replay_buffer.py:82-184,
loop_closure_detection/loop_closure_detection.py:41-76) runs in the build container, where faiss is absent.
Never imported by the product."""
import numpy as np

from oracle import flat_index as OI

METRIC_INNER_PRODUCT = 0


def normalize_L2(x):
    x[:] = OI.normalize_L2(x)


class IndexFlat:
    is_trained = True

    def __init__(self, d):
        self.d = d
        self.x = np.zeros((0, d), np.float32)

    ntotal = property(lambda self: self.x.shape[0])

    def add(self, v):
        self.x = np.concatenate([self.x, np.asarray(v, np.float32).reshape(-1, self.d)])

    def search(self, q, k):
        return OI.search(self.x, np.arange(self.ntotal), np.asarray(q, np.float32).reshape(-1, self.d), k)

    def reconstruct(self, i):
        return self.x[i].copy()

    def reconstruct_n(self, i0, n):
        return self.x[i0:i0 + n].copy()


def index_factory(d, desc, metric):
    assert desc == 'Flat' and metric == METRIC_INNER_PRODUCT
    return IndexFlat(d)


class IndexIDMap:
    def __init__(self, index):
        self.index = index
        self.id_map = np.zeros(0, np.int64)

    ntotal = property(lambda self: self.index.ntotal)

    def add_with_ids(self, v, ids):
        self.index.add(v)
        self.id_map = np.concatenate([self.id_map, np.asarray(ids, np.int64).reshape(-1)])

    def search(self, q, k):
        return OI.search(self.index.x, self.id_map, np.asarray(q, np.float32).reshape(-1, self.index.d), k)

    def remove_ids(self, ids):
        keep = ~np.isin(self.id_map, np.asarray(ids, np.int64))
        self.index.x = self.index.x[keep]
        self.id_map = self.id_map[keep]


def vector_to_array(v):
    return np.array(v)
