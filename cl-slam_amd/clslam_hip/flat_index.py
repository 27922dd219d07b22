"""Exact inner-product index on the MI355X: the subset of faiss's flat index the reference
uses for loop-closure retrieval (loop_closure_detection/loop_closure_detection.py:35-76) and for the replay
buffer's diversity bookkeeping (slam/replay_buffer.py:96-152) -- faiss is not a dependency of this package.

    faiss.index_factory(d, 'Flat', METRIC_INNER_PRODUCT)              -> FlatIPIndex(d)
    faiss.IndexIDMap(faiss.index_factory(...))                         -> FlatIPIndex(d)   (add_with_ids / remove_ids)
    faiss.normalize_L2(x)                                              -> normalize_L2(x)  (in place, numpy fp32)
    index.add(x) / add_with_ids(x, ids) / ntotal / reconstruct(i) / reconstruct_n(i0, n) / remove_ids(ids)
    index.search(x, k) -> (D, I): float32 (nq,k) descending inner products, int64 (nq,k) ids; -FLT_MAX / -1 past ntotal
    faiss.vector_to_array(index.id_map)                                -> index.ids

Vectors live in HBM (capacity doubles); a search is two or three kernel launches (scores, top-k sort, merge).
Equal scores come back in insertion order (faiss leaves ties unspecified)."""
from typing import Tuple

import numpy as np
import torch

from . import _lib, ops

FLT_MAX = float(np.finfo(np.float32).max)


def normalize_L2(x) -> None:
    """faiss.normalize_L2 (fvec_renorm_L2): every row multiplied by 1/sqrt(<row,row>) in place, zero rows stay.
    A C-contiguous float32 numpy matrix is normalised on the host (what the reference hands over); a device tensor
    by clslam_l2_normalize_rows."""
    if isinstance(x, torch.Tensor):
        if not (x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()):
            raise TypeError('normalize_L2 needs a contiguous float32 matrix')
        _lib.get_lib().call('clslam_l2_normalize_rows', x.data_ptr(), x.shape[0], x.shape[1], ops._stream(x))
        return
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 2 and x.flags.c_contiguous):
        raise TypeError('normalize_L2 needs a C-contiguous float32 matrix')
    nrm2 = (x * x).sum(axis=1, dtype=np.float32)
    nz = nrm2 > 0
    x[nz] *= (np.float32(1.0) / np.sqrt(nrm2[nz]))[:, None]


class FlatIPIndex:
    def __init__(self, d: int, device=None) -> None:
        lib = _lib.get_lib()
        self.d = int(d)
        self.device = torch.device(device) if device is not None else torch.device(lib.device_type)
        self._db = torch.empty(64, self.d, device=self.device)
        self._ids = np.empty(0, dtype=np.int64)
        self.ntotal = 0

    # -- storage ---------------------------------------------------------------------------------------
    def _as_rows(self, x) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32)
        if t.ndim == 1:
            t = t[None]
        if t.ndim != 2 or t.shape[1] != self.d:
            raise ValueError(f'expected (n, {self.d}) vectors, got {tuple(t.shape)}')
        return t.to(self.device).contiguous()

    def add(self, x) -> None:
        self.add_with_ids(x, np.arange(self.ntotal, self.ntotal + len(torch.as_tensor(x).reshape(-1, self.d)), dtype=np.int64))

    def add_with_ids(self, x, ids) -> None:
        rows = self._as_rows(x)
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if len(ids) != rows.shape[0]:
            raise ValueError('one id per vector')
        need = self.ntotal + rows.shape[0]
        if need > self._db.shape[0]:
            grown = torch.empty(max(need, 2 * self._db.shape[0]), self.d, device=self.device)
            grown[:self.ntotal] = self._db[:self.ntotal]
            self._db = grown
        self._db[self.ntotal:need] = rows
        self._ids = np.concatenate([self._ids, ids])
        self.ntotal = need

    @property
    def ids(self) -> np.ndarray:
        """ids in storage order (faiss.vector_to_array(index.id_map))"""
        return self._ids.copy()

    def reconstruct(self, i: int) -> np.ndarray:
        if not 0 <= i < self.ntotal:
            raise IndexError(i)
        return self._db[i].cpu().numpy().copy()

    def reconstruct_n(self, i0: int, n: int) -> np.ndarray:
        if i0 < 0 or n < 0 or i0 + n > self.ntotal:
            raise IndexError((i0, n))
        return self._db[i0:i0 + n].cpu().numpy().copy()

    def row(self, i: int) -> torch.Tensor:
        """stored vector i as a device view (no host copy; what `reconstruct` returns after a D2H)"""
        if not 0 <= i < self.ntotal:
            raise IndexError(i)
        return self._db[i]

    def remove_ids(self, ids) -> int:
        """Drop every stored vector whose id is listed; the others keep their order (IndexFlat compaction)."""
        return self.remove_where(np.isin(self._ids, np.asarray(ids, dtype=np.int64).reshape(-1)))

    def remove_where(self, drop: np.ndarray) -> int:
        """Drop the storage positions flagged in the boolean vector `drop` (length ntotal), order kept."""
        keep = ~np.asarray(drop, dtype=bool).reshape(-1)
        if keep.shape[0] != self.ntotal:
            raise ValueError('one flag per stored vector')
        removed = int((~keep).sum())
        if removed:
            sel = torch.from_numpy(np.nonzero(keep)[0]).to(self.device)
            kept = self._db[:self.ntotal].index_select(0, sel)
            self.ntotal = int(keep.sum())
            self._db[:self.ntotal] = kept
            self._ids = self._ids[keep]
        return removed

    # -- search ----------------------------------------------------------------------------------------
    def scores(self, x) -> torch.Tensor:
        """(nq, ntotal) inner products on the device, storage order (no top-k, no host copy)."""
        q = self._as_rows(x)
        out = torch.empty(q.shape[0], self.ntotal, device=self.device)
        if q.shape[0] and self.ntotal:
            _lib.get_lib().call('clslam_ip_scores', self._db.data_ptr(), q.data_ptr(), out.data_ptr(), self.ntotal, self.d,
                                q.shape[0], ops._stream(out))
        return out

    def search(self, x, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = self._as_rows(x)
        nq, n, k = q.shape[0], self.ntotal, int(k)
        if k < 1:
            raise ValueError('k must be positive')
        D = np.full((nq, k), -FLT_MAX, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        if nq == 0 or n == 0:
            return D, I
        lib = _lib.get_lib()
        kk = min(k, 4096)
        chunks = lib.cdll.clslam_topk_chunks(n)
        while chunks * kk > 4096:            # keep the two-level merge inside one workgroup's sort
            kk //= 2
        if kk < min(k, n):
            raise ValueError(f'k={k} over {n} vectors exceeds the two-level top-k (chunks*k <= 4096)')
        scores = torch.empty(nq, n, device=self.device)
        stream = ops._stream(scores)
        lib.call('clslam_ip_scores', self._db.data_ptr(), q.data_ptr(), scores.data_ptr(), n, self.d, nq, stream)
        val = torch.empty(nq, kk, device=self.device)
        pos = torch.empty(nq, kk, dtype=torch.int32, device=self.device)
        cv = ci = None
        if chunks > 1:
            cv = torch.empty(nq, chunks, kk, device=self.device)
            ci = torch.empty(nq, chunks, kk, dtype=torch.int32, device=self.device)
        lib.call('clslam_topk_desc', scores.data_ptr(), n, nq, kk, None if cv is None else cv.data_ptr(),
                 None if ci is None else ci.data_ptr(), val.data_ptr(), pos.data_ptr(), stream)
        pos_h = pos.cpu().numpy().astype(np.int64)
        found = pos_h >= 0
        D[:, :kk] = np.where(found, val.cpu().numpy(), -FLT_MAX)
        I[:, :kk] = np.where(found, self._ids[np.clip(pos_h, 0, None)], -1)
        return D, I
