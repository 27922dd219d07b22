// the exact inner-product search the reference delegates to faiss
// (`faiss.index_factory(d, 'Flat', METRIC_INNER_PRODUCT)`: loop_closure_detection/loop_closure_detection.py:35-36,
// 53-57 -- one 576-d query against every stored frame, top 100; slam/replay_buffer.py:96-98,110,121-122,130 --
// 512-d encoder features, nearest neighbour and the full similarity matrix of <= ~100 samples).
// Brute force by definition ('Flat'): scores = DB . q over rows already L2-normalised by the caller
// (faiss.normalize_L2), then the k largest in descending order.  HBM-bound and tiny (4000 x 576 floats = 9 MB):
//   ip_scores_kernel : one wave per stored vector, 16-byte loads, DPP wave reduction
//   topk_sort_kernel : bitonic sort of 4096 (score, id) pairs per workgroup in LDS, first k kept; a second
//                      pass over the per-chunk winners merges more than 4096 vectors
// Order: descending score, equal scores by ascending position (deterministic; faiss's heap leaves ties unspecified).
#include "common.h"

#include <cfloat>
#include <climits>

namespace clslam {

constexpr int kSortN = 4096;

__global__ __launch_bounds__(256) void ip_scores_kernel(const float* __restrict__ db, const float* __restrict__ q,
                                                        float* __restrict__ scores, int n, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;              // wave-uniform
    const float* qv = q + (size_t)blockIdx.y * d;
    if (row >= n) return;
    const float* x = db + (size_t)row * d;
    float acc = 0.f;
    const int d4 = d & ~3;
    if ((d & 3) == 0) {                                  // rows are 16-byte aligned only then
        for (int i = lane * 4; i < d4; i += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + i);
            const float4 b = *reinterpret_cast<const float4*>(qv + i);
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
    } else {
        for (int i = lane; i < d; i += 64) acc = fmaf(x[i], qv[i], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) scores[(size_t)blockIdx.y * n + row] = acc;
}

// (score a, position ia) sorts before (b, ib)
__device__ __forceinline__ bool before(float a, int ia, float b, int ib) { return a > b || (a == b && ia < ib); }

// in_val [nq][n] (+ in_idx, or the position itself when null); workgroup (x = chunk, y = query) sorts elements
// [chunk*4096, +4096) and writes its first k to out_*[query][chunk][k] (ids -1 / -FLT_MAX past the data).
__global__ __launch_bounds__(256) void topk_sort_kernel(const float* __restrict__ in_val, const int* __restrict__ in_idx,
                                                        int n, int k, float* __restrict__ out_val, int* __restrict__ out_idx) {
    __shared__ float key[kSortN];
    __shared__ int idx[kSortN];
    const int chunk = blockIdx.x, qi = blockIdx.y, chunks = gridDim.x;
    const float* v = in_val + (size_t)qi * n;
    const int* id = in_idx ? in_idx + (size_t)qi * n : nullptr;
    for (int e = threadIdx.x; e < kSortN; e += 256) {
        const int g = chunk * kSortN + e;
        float s = -FLT_MAX;
        int i = INT_MAX;                                 // padding sorts last
        if (g < n) {
            const int src = id ? id[g] : g;
            if (src >= 0) {
                s = v[g];
                if (!(s == s)) s = -FLT_MAX;             // NaN similarity: never a match
                i = src;
            }
        }
        key[e] = s; idx[e] = i;
    }
    __syncthreads();
    for (int size = 2; size <= kSortN; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = threadIdx.x; t < kSortN / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));          // element with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;                  // direction of this bitonic block
                const float a = key[lo], b = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool ordered = before(a, ia, b, ib);
                if (ordered != desc) { key[lo] = b; key[hi] = a; idx[lo] = ib; idx[hi] = ia; }
            }
            __syncthreads();
        }
    }
    float* ov = out_val + ((size_t)qi * chunks + chunk) * k;
    int* oi = out_idx + ((size_t)qi * chunks + chunk) * k;
    for (int e = threadIdx.x; e < k; e += 256) {
        const bool real = e < kSortN && idx[e] != INT_MAX;
        ov[e] = real ? key[e] : -FLT_MAX;
        oi[e] = real ? idx[e] : -1;
    }
}

}  // namespace clslam

using namespace clslam;

extern "C" int clslam_ip_scores(const float* db, const float* queries, float* scores, int n, int d, int nq, void* stream) {
    CLSLAM_REQUIRE(n >= 0 && d >= 1 && nq >= 0, "ip_scores: bad sizes");
    if (n == 0 || nq == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(db && queries && scores, "ip_scores: null");
    CLSLAM_REQUIRE(nq <= 65535, "ip_scores: at most 65535 queries per call");
    hipLaunchKernelGGL(ip_scores_kernel, dim3(cdiv(n, 4), nq), dim3(256), 0, (hipStream_t)stream, db, queries, scores, n, d);
    return check_launch("ip_scores");
}

extern "C" int clslam_topk_chunks(int n) { return std::max(1, cdiv(n, kSortN)); }

extern "C" int clslam_topk_desc(const float* scores, int n, int nq, int k, float* cand_val, int* cand_idx, float* out_val,
                                int* out_idx, void* stream) {
    CLSLAM_REQUIRE(n >= 0 && nq >= 0 && k >= 1 && k <= kSortN, "topk_desc: k must be in [1, 4096]");
    if (nq == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(out_val && out_idx && (scores || n == 0), "topk_desc: null");
    CLSLAM_REQUIRE(nq <= 65535, "topk_desc: at most 65535 queries per call");
    const int chunks = clslam_topk_chunks(n);
    CLSLAM_REQUIRE((size_t)chunks * k <= kSortN, "topk_desc: n/4096 * k must not exceed 4096 (two-level merge)");
    hipStream_t st = (hipStream_t)stream;
    if (chunks == 1) {
        hipLaunchKernelGGL(topk_sort_kernel, dim3(1, nq), dim3(256), 0, st, scores, (const int*)nullptr, n, k, out_val, out_idx);
        return check_launch("topk_desc");
    }
    CLSLAM_REQUIRE(cand_val && cand_idx, "topk_desc: more than 4096 vectors need the candidate buffers");
    hipLaunchKernelGGL(topk_sort_kernel, dim3(chunks, nq), dim3(256), 0, st, scores, (const int*)nullptr, n, k, cand_val, cand_idx);
    int rc = check_launch("topk_desc");
    if (rc != CLSLAM_OK) return rc;
    hipLaunchKernelGGL(topk_sort_kernel, dim3(1, nq), dim3(256), 0, st, (const float*)cand_val, (const int*)cand_idx, chunks * k, k,
                       out_val, out_idx);
    return check_launch("topk_desc");
}
