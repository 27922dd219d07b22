"""TEST INFRASTRUCTURE (never imported by the product): numpy restatement of the exact inner-product search the
reference obtains from faiss 'Flat' + METRIC_INNER_PRODUCT (loop_closure_detection/loop_closure_detection.py:35-36,
53-57; slam/replay_buffer.py:96-98,110,121-122) -- by definition every stored vector is scored with a dot
product and the k largest are returned in descending order, -1 ids past the stored count.  faiss itself is not
installed in the build container, so the tie order (insertion order here) is PARITY UNPINNED against faiss; the
scores and the ranking of distinct scores are the definition."""
import numpy as np

FLT_MAX = np.finfo(np.float32).max


def normalize_L2(x: np.ndarray) -> np.ndarray:
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(1))
    out = x.copy()
    nz = nrm > 0
    out[nz] = (x[nz] / nrm[nz, None]).astype(np.float32)
    return out


def search(db: np.ndarray, ids: np.ndarray, q: np.ndarray, k: int):
    """db (n,d), ids (n,), q (nq,d) -> D (nq,k) float32, I (nq,k) int64"""
    nq, n = q.shape[0], db.shape[0]
    D = np.full((nq, k), -FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if n == 0:
        return D, I
    scores = (q.astype(np.float64) @ db.astype(np.float64).T)
    for i in range(nq):
        order = np.lexsort((np.arange(n), -scores[i]))[:k]      # descending score, ties by position
        D[i, :len(order)] = scores[i, order]
        I[i, :len(order)] = ids[order]
    return D, I
