// Brute-force neighbour queries: pointops2 knnquery, pointgroup_ops knn_batch and
// ballquery_batch_p.  One lane per query point; the candidate cloud of the block's batch
// segment(s) is staged through LDS in 256-point tiles (coalesced 12-byte reads, broadcast LDS
// reads) instead of every lane streaming the whole cloud from global memory.
//
// Distances are evaluated exactly as the reference source writes them —
// (a-b)*(a-b) + (c-d)*(c-d) + (e-f)*(e-f), each operation rounded, no FMA contraction — so that
// exact ties resolve identically (this file is compiled with -ffp-contract=off and uses
// explicit __fmul_rn/__fadd_rn).
#include "common.hpp"

namespace {
constexpr int NQ_BLOCK = 256;

__device__ __forceinline__ float dist2_rn(float ax, float ay, float az, float bx, float by,
                                          float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Block-wide candidate sweep.  Each lane has its own [start,end) candidate range; the block
// walks the union range tile by tile and calls visit(i, x, y, z) for candidates in the lane's
// own range, in ascending i.  `active` lanes with an empty range still take part in barriers.
template <class Visit>
__device__ __forceinline__ void sweep_candidates(const float *__restrict__ cand, int start,
                                                 int end, bool active, Visit visit) {
    __shared__ float tile[NQ_BLOCK * 3];
    __shared__ int range[2];
    if (threadIdx.x == 0) { range[0] = 0x7fffffff; range[1] = 0; }
    doda_sync();
    if (active && end > start) {
        atomicMin(&range[0], start);
        atomicMax(&range[1], end);
    }
    doda_sync();
    const int lo = range[0], hi = range[1];
    for (int t0 = lo; t0 < hi; t0 += NQ_BLOCK) {
        const int n_here = hi - t0 < NQ_BLOCK ? hi - t0 : NQ_BLOCK;
        for (int e = threadIdx.x; e < n_here * 3; e += NQ_BLOCK) tile[e] = cand[(long long)t0 * 3 + e];
        doda_sync();
        if (active) {
            const int b = start > t0 ? start - t0 : 0;
            const int e = end - t0 < n_here ? end - t0 : n_here;
            for (int c = b; c < e; ++c) visit(t0 + c, tile[c * 3], tile[c * 3 + 1], tile[c * 3 + 2]);
        }
        doda_sync();
    }
}

// ---- pointops2 knnquery (knnquery_cuda_kernel.cu:20-108) -------------------------------------
// Max-heap of the nsample best: sift-down picks the larger child and stops only when the parent
// is STRICTLY greater; replacement requires d2 STRICTLY below the root.  Final order comes from
// an in-place heap sort.  These three rules fix the tie order, so they are restated exactly.
__device__ __forceinline__ void sift_down(float *d, int *ix, int n) {
    int root = 0;
    for (int child = 1; child < n; child = 2 * root + 1) {
        if (child + 1 < n && d[child + 1] > d[child]) ++child;
        if (d[root] > d[child]) return;
        const float td = d[root]; d[root] = d[child]; d[child] = td;
        const int ti = ix[root]; ix[root] = ix[child]; ix[child] = ti;
        root = child;
    }
}

__device__ __forceinline__ int batch_of(int q, const int32_t *__restrict__ ends, int nbatch) {
    int b = 0;
    while (b < nbatch - 1 && q >= ends[b]) ++b;  // reference get_bt_idx, bounded
    return b;
}

__global__ __launch_bounds__(NQ_BLOCK) void knnquery_k1(int m, const float *__restrict__ xyz,
                                                        const float *__restrict__ new_xyz,
                                                        const int32_t *__restrict__ offset,
                                                        const int32_t *__restrict__ new_offset,
                                                        int nbatch, int32_t *__restrict__ idx,
                                                        float *__restrict__ dist2) {
    const int q = blockIdx.x * NQ_BLOCK + threadIdx.x;
    const bool active = q < m;
    int start = 0, end = 0;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const int b = batch_of(q, new_offset, nbatch);
        start = b == 0 ? 0 : offset[b - 1];
        end = offset[b];
        qx = new_xyz[q * 3LL]; qy = new_xyz[q * 3LL + 1]; qz = new_xyz[q * 3LL + 2];
    }
    float best = 1e10f;
    int besti = start;
    sweep_candidates(xyz, start, end, active, [&](int i, float x, float y, float z) {
        const float d = dist2_rn(qx, qy, qz, x, y, z);
        if (d < best) { best = d; besti = i; }
    });
    if (active) { idx[q] = besti; dist2[q] = best; }
}

__global__ __launch_bounds__(NQ_BLOCK) void knnquery_any(int m, int k,
                                                         const float *__restrict__ xyz,
                                                         const float *__restrict__ new_xyz,
                                                         const int32_t *__restrict__ offset,
                                                         const int32_t *__restrict__ new_offset,
                                                         int nbatch, int32_t *__restrict__ idx,
                                                         float *__restrict__ dist2) {
    const int q = blockIdx.x * NQ_BLOCK + threadIdx.x;
    const bool active = q < m;
    int start = 0, end = 0;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const int b = batch_of(q, new_offset, nbatch);
        start = b == 0 ? 0 : offset[b - 1];
        end = offset[b];
        qx = new_xyz[q * 3LL]; qy = new_xyz[q * 3LL + 1]; qz = new_xyz[q * 3LL + 2];
    }
    float bd[100];
    int bi[100];
    for (int j = 0; j < k; ++j) { bd[j] = 1e10f; bi[j] = start; }
    sweep_candidates(xyz, start, end, active, [&](int i, float x, float y, float z) {
        const float d = dist2_rn(qx, qy, qz, x, y, z);
        if (d < bd[0]) { bd[0] = d; bi[0] = i; sift_down(bd, bi, k); }
    });
    if (active) {
        for (int n = k - 1; n > 0; --n) {  // heap sort, ascending
            const float td = bd[0]; bd[0] = bd[n]; bd[n] = td;
            const int ti = bi[0]; bi[0] = bi[n]; bi[n] = ti;
            sift_down(bd, bi, n);
        }
        for (int j = 0; j < k; ++j) { idx[(long long)q * k + j] = bi[j]; dist2[(long long)q * k + j] = bd[j]; }
    }
}

// ---- pointgroup_ops knn_batch (knn.cu:7-50) ---------------------------------------------------
__global__ __launch_bounds__(NQ_BLOCK) void knn_batch_kernel(int n, int k,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ query_xyz,
                                                             const int32_t *__restrict__ batch_idxs,
                                                             const int32_t *__restrict__ qoff,
                                                             int32_t *__restrict__ idx) {
    const int p = blockIdx.x * NQ_BLOCK + threadIdx.x;
    const bool active = p < n;
    int start = 0, end = 0;
    float ox = 0, oy = 0, oz = 0;
    if (active) {
        const int b = batch_idxs[p];
        start = qoff[b]; end = qoff[b + 1];
        ox = xyz[p * 3LL]; oy = xyz[p * 3LL + 1]; oz = xyz[p * 3LL + 2];
    }
    float best[40];
    int besti[40];
    for (int j = 0; j < k; ++j) { best[j] = 1e20f; besti[j] = 0; }
    sweep_candidates(query_xyz, start, end, active, [&](int i, float x, float y, float z) {
        const float d = dist2_rn(ox, oy, oz, x, y, z);
        for (int s = 0; s < k; ++s) {
            if (d < best[s]) {  // first slot strictly worse: shift the tail down, insert
                for (int r = k - 1; r > s; --r) { best[r] = best[r - 1]; besti[r] = besti[r - 1]; }
                best[s] = d; besti[s] = i;
                break;
            }
        }
    });
    if (active)
        for (int j = 0; j < k; ++j) idx[(long long)p * k + j] = besti[j];
}

// ---- pointgroup_ops ballquery_batch_p (bfs_cluster.cu:15-60), deterministic two-pass ---------
__global__ __launch_bounds__(NQ_BLOCK) void ball_count(int n, float radius,
                                                       const float *__restrict__ xyz,
                                                       const int32_t *__restrict__ batch_idxs,
                                                       const int32_t *__restrict__ boff,
                                                       int32_t *__restrict__ counts) {
    const int p = blockIdx.x * NQ_BLOCK + threadIdx.x;
    const bool active = p < n;
    int start = 0, end = 0;
    float ox = 0, oy = 0, oz = 0;
    if (active) {
        const int b = batch_idxs[p];
        start = boff[b]; end = boff[b + 1];
        ox = xyz[p * 3LL]; oy = xyz[p * 3LL + 1]; oz = xyz[p * 3LL + 2];
    }
    const float r2 = __fmul_rn(radius, radius);
    int cnt = 0;
    sweep_candidates(xyz, start, end, active, [&](int, float x, float y, float z) {
        if (dist2_rn(ox, oy, oz, x, y, z) < r2 && cnt < 1000) ++cnt;
    });
    if (active) counts[p] = cnt;
}

__global__ __launch_bounds__(NQ_BLOCK) void ball_fill(int n, int mean_active, float radius,
                                                      const float *__restrict__ xyz,
                                                      const int32_t *__restrict__ batch_idxs,
                                                      const int32_t *__restrict__ boff,
                                                      const int32_t *__restrict__ counts,
                                                      const int32_t *__restrict__ starts,
                                                      int32_t *__restrict__ idx,
                                                      int32_t *__restrict__ start_len) {
    const int p = blockIdx.x * NQ_BLOCK + threadIdx.x;
    const bool active = p < n;
    int start = 0, end = 0, cnt = 0, s0 = 0;
    float ox = 0, oy = 0, oz = 0;
    if (active) {
        const int b = batch_idxs[p];
        start = boff[b]; end = boff[b + 1];
        ox = xyz[p * 3LL]; oy = xyz[p * 3LL + 1]; oz = xyz[p * 3LL + 2];
        cnt = counts[p]; s0 = starts[p];
        start_len[p * 2LL] = s0;
        start_len[p * 2LL + 1] = cnt;
    }
    const long long thre = (long long)n * mean_active;
    int room = 0;
    if (active && s0 < thre) room = (s0 + (long long)cnt >= thre) ? (int)(thre - s0) : cnt;
    const float r2 = __fmul_rn(radius, radius);
    int w = 0;
    sweep_candidates(xyz, start, end, active, [&](int i, float x, float y, float z) {
        if (w < room && dist2_rn(ox, oy, oz, x, y, z) < r2) { idx[s0 + w] = i; ++w; }
    });
}
}  // namespace

extern "C" int doda_knnquery(int32_t m, int32_t nsample, const float *xyz, const float *new_xyz,
                             const int32_t *offset, const int32_t *new_offset, int32_t nbatch,
                             int32_t *idx, float *dist2, doda_stream_t stream) {
    if (m < 0 || nsample <= 0 || nbatch <= 0) return DODA_ERR_INVALID;
    if (nsample > 100) return DODA_ERR_UNSUPPORTED;
    if (m == 0) return DODA_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx || !dist2) return DODA_ERR_INVALID;
    const dim3 grid(div_up(m, NQ_BLOCK)), block(NQ_BLOCK);
    if (nsample == 1)
        hipLaunchKernelGGL(knnquery_k1, grid, block, 0, as_stream(stream), m, xyz, new_xyz, offset,
                           new_offset, nbatch, idx, dist2);
    else
        hipLaunchKernelGGL(knnquery_any, grid, block, 0, as_stream(stream), m, nsample, xyz,
                           new_xyz, offset, new_offset, nbatch, idx, dist2);
    return doda_check_launch();
}

extern "C" int doda_knn_batch(int32_t n, int32_t m, int32_t k, const float *xyz,
                              const float *query_xyz, const int32_t *batch_idxs,
                              const int32_t *query_batch_offsets, int32_t *idx,
                              doda_stream_t stream) {
    (void)m;
    if (n < 0 || k <= 0) return DODA_ERR_INVALID;
    if (k > 40) return DODA_ERR_UNSUPPORTED;
    if (n == 0) return DODA_OK;
    if (!xyz || !query_xyz || !batch_idxs || !query_batch_offsets || !idx) return DODA_ERR_INVALID;
    hipLaunchKernelGGL(knn_batch_kernel, dim3(div_up(n, NQ_BLOCK)), dim3(NQ_BLOCK), 0,
                       as_stream(stream), n, k, xyz, query_xyz, batch_idxs, query_batch_offsets,
                       idx);
    return doda_check_launch();
}

extern "C" size_t doda_ballquery_workspace_bytes(int32_t n) {
    const size_t ni = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    return 2 * ni + align_up(scan_ws_ints(n) * 4, 256) + 256;
}

extern "C" int doda_ballquery_batch_p(int32_t n, int32_t mean_active, float radius,
                                      const float *xyz, const int32_t *batch_idxs,
                                      const int32_t *batch_offsets, int32_t *idx,
                                      int32_t *start_len, int32_t *total_h, void *ws,
                                      size_t ws_bytes, doda_stream_t stream) {
    if (n < 0 || mean_active < 0 || !total_h) return DODA_ERR_INVALID;
    *total_h = 0;
    if (n == 0) return DODA_OK;
    if (!xyz || !batch_idxs || !batch_offsets || !idx || !start_len || !ws) return DODA_ERR_INVALID;
    if (ws_bytes < doda_ballquery_workspace_bytes(n)) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const size_t ni = align_up((size_t)n * 4, 256);
    char *p = (char *)ws;
    int32_t *counts = (int32_t *)p;
    int32_t *starts = (int32_t *)(p + ni);
    int32_t *scan = (int32_t *)(p + 2 * ni);
    int32_t *total_d = (int32_t *)(p + 2 * ni + align_up(scan_ws_ints(n) * 4, 256));
    const dim3 grid(div_up(n, NQ_BLOCK)), block(NQ_BLOCK);
    hipLaunchKernelGGL(ball_count, grid, block, 0, s, n, radius, xyz, batch_idxs, batch_offsets,
                       counts);
    int st = exclusive_scan_i32(counts, starts, n, total_d, scan, s);
    if (st != DODA_OK) return st;
    hipLaunchKernelGGL(ball_fill, grid, block, 0, s, n, mean_active, radius, xyz, batch_idxs,
                       batch_offsets, counts, starts, idx, start_len);
    st = doda_check_launch();
    if (st != DODA_OK) return st;
    if (hipMemcpyAsync(total_h, total_d, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess)
        return DODA_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return DODA_ERR_LAUNCH;
    return DODA_OK;
}
