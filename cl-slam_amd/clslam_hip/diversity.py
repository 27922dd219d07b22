"""Diversity-maximising replay buffer bookkeeping on the MI355X.

What the reference's ``ReplayBuffer.add`` does with faiss + a numpy similarity matrix when
``maximize_diversity`` is on (slam/replay_buffer.py:104-152):

    similarity = nearest stored cosine similarity (0 for an empty buffer)                :107-110
    if similarity < similarity_threshold: store the sample                               :112-113
        if more than max_buffer_size samples are stored:                                 :118
            keep the full similarity matrix (built once :120-127, then only the row/column of the
            newcomer is rewritten in the slot freed by the previous eviction :128-139)
            evict argmax(column sums - self similarity) = the sample most similar to all :141-150

Here the stored vectors, the similarity matrix and the occupancy flags stay in HBM in *slot* order; one
candidate costs two launches (clslam_ip_scores, clslam_diversity_commit) and ONE 20-byte D2H -- the host must
learn which sample file to write / delete (replay_buffer.py:163-184), nothing else crosses.  The features are
the depth encoder's pooled output (slam/slam.py:143-147), which is already on the device.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from .flat_index import normalize_L2


class DiversityBuffer:
    def __init__(self, d: int, capacity: int, similarity_threshold: float, device=None) -> None:
        lib = _lib.get_lib()
        if capacity < 1:
            raise ValueError('capacity must be positive')
        self.d, self.capacity, self.threshold = int(d), int(capacity), float(similarity_threshold)
        self.device = torch.device(device) if device is not None else torch.device(lib.device_type)
        self.max_slots = self.capacity + 1          # the newcomer is stored before the eviction (replay_buffer.py:113-118)
        self._db = torch.zeros(self.max_slots, self.d, device=self.device)
        self._sim = torch.full((self.max_slots, self.max_slots), -1.0, device=self.device)
        self._occ = torch.zeros(self.max_slots, dtype=torch.uint8, device=self.device)
        self._scores = torch.empty(self.max_slots, device=self.device)
        self._result = torch.empty(5, dtype=torch.int32, device=self.device)
        self._simout = torch.empty(1, device=self.device)
        self._slot_ids = np.full(self.max_slots, -1, dtype=np.int64)   # distance_matrix_indices of the reference
        self.nslots = 0                               # slots ever used (high-water mark)

    # ------------------------------------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return int((self._slot_ids >= 0).sum())

    @property
    def ids(self) -> np.ndarray:
        """sample ids of the occupied slots, slot order"""
        return self._slot_ids[self._slot_ids >= 0].copy()

    def _query(self, feature) -> torch.Tensor:
        q = torch.as_tensor(feature, dtype=torch.float32).reshape(1, -1)
        if q.shape[1] != self.d:
            raise ValueError(f'expected a {self.d}-d feature, got {tuple(q.shape)}')
        q = q.to(self.device).contiguous().clone()
        normalize_L2(q)                               # replay_buffer.py:102
        return q

    def add(self, feature, sample_id: int) -> Tuple[bool, Optional[int], float]:
        """One candidate.  Returns (stored?, id of the evicted sample or None, nearest similarity)."""
        lib = _lib.get_lib()
        q = self._query(feature)
        stream = ops._stream(q)
        if self.nslots:
            lib.call('clslam_ip_scores', self._db.data_ptr(), q.data_ptr(), self._scores.data_ptr(), self.nslots, self.d, 1,
                     stream)
        lib.call('clslam_diversity_commit', self._db.data_ptr(), self._sim.data_ptr(), self.max_slots, self._occ.data_ptr(),
                 self.nslots, self.max_slots, self.d, self.capacity, self.threshold, q.data_ptr(), self._scores.data_ptr(),
                 self._result.data_ptr(), self._simout.data_ptr(), stream)
        res = self._result.cpu()                       # the one device->host copy of a candidate (20 bytes)
        accepted, slot, evict, _count = (int(v) for v in res[:4])
        similarity = float(res[4:5].view(torch.float32))
        if not accepted:
            return False, None, similarity
        self._slot_ids[slot] = int(sample_id)
        self.nslots = max(self.nslots, slot + 1)
        removed = None
        if evict >= 0:
            removed = int(self._slot_ids[evict])
            self._slot_ids[evict] = -1
        return True, removed, similarity

    def similarities(self, feature) -> Tuple[np.ndarray, np.ndarray]:
        """(similarity, id) of every stored sample to `feature`, slot order (the similarity-sampling weights of
        ReplayBuffer.get, replay_buffer.py:211-226)."""
        q = self._query(feature)
        if self.nslots:
            _lib.get_lib().call('clslam_ip_scores', self._db.data_ptr(), q.data_ptr(), self._scores.data_ptr(), self.nslots,
                                self.d, 1, ops._stream(q))
        s = self._scores[:self.nslots].cpu().numpy()
        occ = self._slot_ids[:self.nslots] >= 0
        return s[occ], self._slot_ids[:self.nslots][occ].copy()

    def similarity_matrix(self) -> Tuple[np.ndarray, List[int]]:
        """host copy of the slot-ordered similarity matrix and the slot -> id table (tests / inspection)"""
        n = self.nslots
        return self._sim[:n, :n].cpu().numpy(), [int(i) for i in self._slot_ids[:n]]
