"""``faiss``-named module over the MI355X exact inner-product index (clslam_hip.flat_index) -- the SUBSET of
faiss the reference's hot-path callers use, so that slam/replay_buffer.py (:7, :95-160, :212-215, :237-250) and
loop_closure_detection/loop_closure_detection.py (:4, :35-57) run unchanged where faiss-gpu
(requirements.txt:20) is not installed:

    faiss.METRIC_INNER_PRODUCT, faiss.index_factory(d, 'Flat', METRIC_INNER_PRODUCT)
    index.add / search / reconstruct / reconstruct_n / remove_ids / ntotal / d / is_trained / reset
    faiss.IndexIDMap(index): .index, .id_map, add_with_ids / search / remove_ids / ntotal
    faiss.vector_to_array(index.id_map), faiss.normalize_L2(x)
    pickling an index (ReplayBuffer.save_state / load_state)

Arrays cross this interface as numpy (that is faiss's interface); vectors and scores live on the GPU.  This
directory is only on sys.path when cl-slam_amd/ is; remove it to use a real faiss install instead.  Anything
outside the subset raises NotImplementedError instead of guessing.  Exact ties come back in insertion order
(faiss leaves them unspecified)."""
import numpy as np

from clslam_hip.flat_index import FLT_MAX, FlatIPIndex
from clslam_hip.flat_index import normalize_L2 as _normalize_L2

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
__version__ = '0+clslam_hip'


def normalize_L2(x) -> None:
    _normalize_L2(x)


def _as_ids(ids) -> np.ndarray:
    if hasattr(ids, '_ids'):                     # an IdVector
        ids = ids._ids
    return np.asarray(ids, dtype=np.int64).reshape(-1)


class IdVector:
    """stands in for the SWIG std::vector<idx_t> behind IndexIDMap.id_map (read with faiss.vector_to_array)"""

    def __init__(self, owner: 'IndexIDMap') -> None:
        self._owner = owner

    @property
    def _ids(self) -> np.ndarray:
        return self._owner._core.ids

    def size(self) -> int:
        return int(self._owner.ntotal)

    def at(self, i: int) -> int:
        return int(self._ids[i])


def vector_to_array(v) -> np.ndarray:
    if isinstance(v, IdVector):
        return v._ids.copy()
    return np.array(v)


class IndexFlatIP:
    """faiss.IndexFlatIP: ids are storage positions; remove_ids compacts and renumbers."""
    metric_type = METRIC_INNER_PRODUCT
    is_trained = True

    def __init__(self, d: int) -> None:
        self._core = FlatIPIndex(int(d))

    d = property(lambda self: self._core.d)
    ntotal = property(lambda self: self._core.ntotal)

    def add(self, x) -> None:
        n = np.asarray(x).reshape(-1, self.d).shape[0]
        self._core.add_with_ids(x, np.arange(self.ntotal, self.ntotal + n, dtype=np.int64))

    def search(self, x, k: int):
        return self._core.search(np.asarray(x, dtype=np.float32), k)

    def reconstruct(self, i: int) -> np.ndarray:
        return self._core.reconstruct(int(i))

    def reconstruct_n(self, i0: int, n: int) -> np.ndarray:
        return self._core.reconstruct_n(int(i0), int(n))

    def remove_ids(self, ids) -> int:
        removed = self._core.remove_ids(_as_ids(ids))
        self._core._ids = np.arange(self._core.ntotal, dtype=np.int64)
        return removed

    def reset(self) -> None:
        self._core.ntotal = 0
        self._core._ids = np.empty(0, dtype=np.int64)

    def __getstate__(self):
        return {'d': self.d, 'x': self._core.reconstruct_n(0, self.ntotal), 'ids': self._core.ids}

    def __setstate__(self, st) -> None:
        self._core = FlatIPIndex(int(st['d']))
        if len(st['ids']):
            self._core.add_with_ids(st['x'], st['ids'])


IndexFlat = IndexFlatIP


def index_factory(d: int, description: str, metric: int = METRIC_L2):
    if description != 'Flat' or metric != METRIC_INNER_PRODUCT:
        raise NotImplementedError(f"index_factory({d}, {description!r}, metric={metric}): only ('Flat', "
                                  'METRIC_INNER_PRODUCT) is provided (the reference uses nothing else)')
    return IndexFlatIP(d)


class IndexIDMap:
    """faiss.IndexIDMap over a flat index: arbitrary int64 ids, `index` = the wrapped flat index."""
    is_trained = True

    def __init__(self, index: IndexFlatIP) -> None:
        if not isinstance(index, IndexFlatIP):
            raise NotImplementedError('IndexIDMap wraps the flat inner-product index only')
        if index.ntotal:
            raise RuntimeError('index must be empty on input')            # faiss's own precondition
        self.index = index
        self._core = index._core
        self.id_map = IdVector(self)

    d = property(lambda self: self._core.d)
    ntotal = property(lambda self: self._core.ntotal)
    metric_type = METRIC_INNER_PRODUCT

    def add(self, x) -> None:
        raise RuntimeError('add does not make sense with IndexIDMap, use add_with_ids')

    def add_with_ids(self, x, ids) -> None:
        self._core.add_with_ids(np.asarray(x, dtype=np.float32), _as_ids(ids))

    def search(self, x, k: int):
        return self._core.search(np.asarray(x, dtype=np.float32), k)

    def remove_ids(self, ids) -> int:
        return self._core.remove_ids(_as_ids(ids))

    def __getstate__(self):
        return {'index': self.index, 'ids': self._core.ids}

    def __setstate__(self, st) -> None:
        self.index = st['index']
        self._core = self.index._core
        self._core._ids = np.asarray(st['ids'], dtype=np.int64).copy()
        self.id_map = IdVector(self)


__all__ = ['METRIC_INNER_PRODUCT', 'METRIC_L2', 'IndexFlat', 'IndexFlatIP', 'IndexIDMap', 'index_factory', 'normalize_L2',
           'vector_to_array', 'FLT_MAX']
