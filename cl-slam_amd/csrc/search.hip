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


// faiss.normalize_L2 (fvec_renorm_L2): x_i *= 1/sqrt(<x_i,x_i>) for rows with a non-zero norm.  One wave per row.
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* __restrict__ x, int n, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;              // wave-uniform
    if (row >= n) return;
    float* r = x + (size_t)row * d;
    float acc = 0.f;
    for (int i = lane; i < d; i += 64) acc = fmaf(r[i], r[i], acc);
    acc = wave_sum(acc);
    if (!(acc > 0.f)) return;
    const float inv = 1.0f / sqrtf(acc);
    for (int i = lane; i < d; i += 64) r[i] *= inv;
}

// The replay buffer's diversity bookkeeping (slam/replay_buffer.py:104-152) for one candidate, in ONE workgroup:
// the similarity matrix S of the stored samples lives in HBM in SLOT order (a freed slot is re-used by the next
// accepted sample, exactly the reference's `fill_up_index`), so nothing is ever compacted or rebuilt.
//   scores[j] = <db[j], q> for every slot j < nslots (clslam_ip_scores ran before this launch)
//   1. similarity = max over occupied slots (0 when the buffer is empty, replay_buffer.py:107-110)
//   2. similarity < threshold: q goes to the first free slot (or slot nslots), row/column of S are its scores,
//      S[slot][slot] = <q,q>                                                (replay_buffer.py:112-139)
//   3. more than `capacity` samples now: evict argmax_j (sum_i S[i][j] - S[j][j]), the sample most similar to
//      all others (first maximum, column sums added in slot order like numpy's sum(0))   (replay_buffer.py:141-150)
// result: [0]=accepted, [1]=slot written (-1), [2]=slot evicted (-1), [3]=occupied slots afterwards, [4]=the bits of the
// nearest similarity (so that ONE 20-byte copy tells the host everything); sim_out[0] holds it as a float as well.
__global__ __launch_bounds__(256) void diversity_commit_kernel(float* __restrict__ db, float* __restrict__ S, int ld,
                                                               unsigned char* __restrict__ occupied, int nslots, int max_slots,
                                                               int d, int capacity, float threshold,
                                                               const float* __restrict__ q, const float* __restrict__ scores,
                                                               int* __restrict__ result, float* __restrict__ sim_out) {
    __shared__ float red_v[256];
    __shared__ int red_i[256];
    __shared__ int sh_slot, sh_count;
    const int tid = threadIdx.x;
    // 1. nearest stored sample; first free slot; occupancy
    float best = -FLT_MAX;
    int best_i = INT_MAX, free_slot = INT_MAX, cnt = 0;
    for (int j = tid; j < nslots; j += 256) {
        if (occupied[j]) {
            ++cnt;
            const float s = scores[j];
            if (s > best || (s == best && j < best_i)) { best = s; best_i = j; }
        } else if (j < free_slot) {
            free_slot = j;
        }
    }
    red_v[tid] = best; red_i[tid] = best_i;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st && before(red_v[tid + st], red_i[tid + st], red_v[tid], red_i[tid])) {
            red_v[tid] = red_v[tid + st]; red_i[tid] = red_i[tid + st];
        }
        __syncthreads();
    }
    const float similarity = red_i[0] == INT_MAX ? 0.f : red_v[0];
    __syncthreads();
    red_i[tid] = free_slot; red_v[tid] = (float)cnt;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) { red_i[tid] = min(red_i[tid], red_i[tid + st]); red_v[tid] += red_v[tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        sh_slot = red_i[0] == INT_MAX ? nslots : red_i[0];
        sh_count = (int)red_v[0];
        sim_out[0] = similarity;
        result[4] = __builtin_bit_cast(int, similarity);
    }
    __syncthreads();
    const int slot = sh_slot;
    const bool accept = similarity < threshold && slot < max_slots;
    if (!accept) {
        if (tid == 0) { result[0] = 0; result[1] = -1; result[2] = -1; result[3] = sh_count; }
        return;
    }
    // 2. store the sample and its similarities
    float qq = 0.f;
    for (int i = tid; i < d; i += 256) {
        const float v = q[i];
        db[(size_t)slot * d + i] = v;
        qq = fmaf(v, v, qq);
    }
    __syncthreads();
    red_v[tid] = qq;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) red_v[tid] += red_v[tid + st];
        __syncthreads();
    }
    qq = red_v[0];
    __syncthreads();
    const int n_after = max(nslots, slot + 1);
    for (int j = tid; j < n_after; j += 256) {
        const float s = j == slot ? qq : (occupied[j] ? scores[j] : -1.f);
        S[(size_t)slot * ld + j] = s;
        S[(size_t)j * ld + slot] = s;
    }
    __syncthreads();
    if (tid == 0) occupied[slot] = 1;
    __syncthreads();
    const int count = sh_count + 1;
    int evict = -1;
    if (count > capacity) {
        // 3. column sums in slot order (coalesced across the columns), minus the self-similarity
        float bv = -FLT_MAX;
        int bi = INT_MAX;
        for (int j = tid; j < n_after; j += 256) {
            if (!occupied[j]) continue;
            float acc = 0.f;
            for (int i = 0; i < n_after; ++i)
                if (occupied[i]) acc += S[(size_t)i * ld + j];
            acc -= S[(size_t)j * ld + j];
            if (acc > bv) { bv = acc; bi = j; }          // ascending j per thread: first maximum kept
        }
        red_v[tid] = bv; red_i[tid] = bi;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st && before(red_v[tid + st], red_i[tid + st], red_v[tid], red_i[tid])) {
                red_v[tid] = red_v[tid + st]; red_i[tid] = red_i[tid + st];
            }
            __syncthreads();
        }
        evict = red_i[0];
        __syncthreads();
        for (int j = tid; j < n_after; j += 256) {          // replay_buffer.py:143-145
            S[(size_t)evict * ld + j] = -1.f;
            S[(size_t)j * ld + evict] = -1.f;
        }
        if (tid == 0) occupied[evict] = 0;
    }
    if (tid == 0) { result[0] = 1; result[1] = slot; result[2] = evict; result[3] = count - (evict >= 0 ? 1 : 0); }
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

extern "C" int clslam_l2_normalize_rows(float* x, int n, int d, void* stream) {
    CLSLAM_REQUIRE(n >= 0 && d >= 1, "l2_normalize_rows: bad sizes");
    if (n == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(x, "l2_normalize_rows: null");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, n, d);
    return check_launch("l2_normalize_rows");
}

extern "C" int clslam_diversity_commit(float* db, float* sim, int ld, unsigned char* occupied, int nslots, int max_slots, int d,
                                       int capacity, float threshold, const float* query, const float* scores, int* result,
                                       float* similarity, void* stream) {
    CLSLAM_REQUIRE(db && sim && occupied && query && result && similarity, "diversity_commit: null");
    CLSLAM_REQUIRE(nslots >= 0 && nslots <= max_slots && max_slots <= ld && d >= 1 && capacity >= 1,
                   "diversity_commit: bad sizes");
    CLSLAM_REQUIRE(scores || nslots == 0, "diversity_commit: scores missing");
    hipLaunchKernelGGL(diversity_commit_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, db, sim, ld, occupied, nslots,
                       max_slots, d, capacity, threshold, query, scores, result, similarity);
    return check_launch("diversity_commit");
}
