"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of
  * the diversity-maximising branch of ``ReplayBuffer.add`` (slam/replay_buffer.py:100-152) and
  * ``LoopClosureDetection.add/search`` (loop_closure_detection/loop_closure_detection.py:41-76)
with faiss's exact inner-product search written out (oracle.flat_index).  Pinned: tests/golden/replay_lcd.npz
holds the decisions of the REFERENCE's own code on the same seeded streams (tests/golden/make_replay_golden.py,
faiss replaced by tests/golden/numpy_faiss.py because faiss is not installed in the build container)."""
from typing import List, Optional, Tuple

import numpy as np

from . import flat_index as OI


class ReplayDiversity:
    """state of replay_buffer.py's faiss_index / distance_matrix / distance_matrix_indices"""

    def __init__(self, capacity: int, threshold: float) -> None:
        self.capacity, self.threshold = capacity, threshold
        self.x: Optional[np.ndarray] = None          # stored (normalised) features, faiss storage order
        self.ids = np.zeros(0, np.int64)             # id_map
        self.dist: Optional[np.ndarray] = None       # distance_matrix      (:119-127)
        self.dist_ids: Optional[np.ndarray] = None   # distance_matrix_indices

    def add(self, feature: np.ndarray, index: int) -> Tuple[bool, Optional[int], float]:
        f = OI.normalize_L2(np.asarray(feature, np.float32).reshape(1, -1))         # :102
        if self.x is None:
            self.x = np.zeros((0, f.shape[1]), np.float32)
        similarity = 0.0 if len(self.ids) == 0 else float(OI.search(self.x, self.ids, f, 1)[0][0, 0])   # :107-110
        if not similarity < self.threshold:                                         # :112
            return False, None, similarity
        self.x = np.concatenate([self.x, f])                                        # :113
        self.ids = np.concatenate([self.ids, [index]])
        removed = None
        n = len(self.ids)
        if n > self.capacity:                                                       # :118
            if self.dist is None:                                                   # :120-127
                D, I = OI.search(self.x, self.ids, self.x, n)
                for i in range(n):
                    D[i, :] = D[i, I[i].argsort()]
                self.dist, self.dist_ids = D, self.ids.copy()
            else:                                                                   # :129-139
                slot = int(np.argwhere(self.dist_ids < 0)[0, 0])
                a, b = OI.search(self.x, self.ids, f, n)
                self.dist_ids[slot] = index
                sorter = np.argsort(b[0])
                a = a[:, sorter[np.searchsorted(b[0], self.dist_ids, sorter=sorter)]][0]
                self.dist[slot, :] = self.dist[:, slot] = a
            victim = int(np.argmax(self.dist.sum(0) - self.dist.diagonal()))        # :141-143
            self.dist[:, victim] = self.dist[victim, :] = -1
            removed = int(self.dist_ids[victim])
            self.dist_ids[victim] = -1
            keep = self.ids != removed                                              # :148 remove_ids
            self.x, self.ids = self.x[keep], self.ids[keep]
        return True, removed, similarity


class LoopClosure:
    def __init__(self, threshold: float, id_threshold: int, num_matches: int) -> None:
        self.threshold, self.id_threshold, self.num_matches = threshold, id_threshold, num_matches
        self.x: Optional[np.ndarray] = None
        self.image_id_to_index, self.index_to_image_id = {}, {}

    def add(self, image_id: int, feature: np.ndarray) -> None:                      # :41-50
        f = OI.normalize_L2(np.asarray(feature, np.float32).reshape(1, -1))
        self.x = f if self.x is None else np.concatenate([self.x, f])
        self.image_id_to_index[image_id] = len(self.x) - 1
        self.index_to_image_id[len(self.x) - 1] = image_id

    def search(self, image_id: int) -> Tuple[List[int], np.ndarray]:                # :53-76
        index_id = self.image_id_to_index[image_id]
        D, I = OI.search(self.x, np.arange(len(self.x)), self.x[index_id:index_id + 1], 100)
        D, I = D[0], I[0]
        for keep in (lambda: I != -1, lambda: I != index_id, lambda: D > self.threshold,
                     lambda: np.abs(I - index_id) > self.id_threshold):
            m = keep()
            D, I = D[m], I[m]
        D, I = D[:self.num_matches], I[:self.num_matches]
        return sorted(self.index_to_image_id[int(i)] for i in I), D
