// Cross-entropy with ignore_index over point logits [N, C] (reference model/unet.py:107-108,196:
// nn.CrossEntropyLoss(ignore_index=ignore_label) on the Linear head's scores), forward and backward.
//
// torch builds this from ~12 kernels (log_softmax, gather, mask, reductions; softmax backward,
// scatter, scaling) that each stream the 104 MB logit matrix of a 1.3 M-point batch.  Here: one pass
// that writes the per-point log-sum-exp and block partials of (loss, valid count), a fixed-order
// combine, and one backward pass  dlogits = (softmax - onehot) * g / n_valid.  Deterministic.
#include "common.hpp"

namespace {
constexpr int CE_BLOCK = 256;
constexpr int CE_MAX_C = 64;

__global__ __launch_bounds__(CE_BLOCK) void ce_fwd(const float *__restrict__ logits,
                                                   const long long *__restrict__ labels, int n, int c,
                                                   long long ignore_index, float *__restrict__ lse,
                                                   float *__restrict__ partial /*[blocks][2]*/) {
    __shared__ float red[2][CE_BLOCK / 64];
    float loss = 0.f, cnt = 0.f;
    for (long long t = (long long)blockIdx.x * CE_BLOCK + threadIdx.x; t < n; t += (long long)gridDim.x * CE_BLOCK) {
        const float *row = logits + t * c;
        float m = -INFINITY;   // two sweeps over the 80-byte row (the second one hits L1): no private array
#pragma unroll 4
        for (int k = 0; k < c; ++k) m = fmaxf(m, row[k]);
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < c; ++k) s += __expf(row[k] - m);
        const float l = m + __logf(s);
        lse[t] = l;
        const long long lab = labels[t];
        if (lab != ignore_index && lab >= 0 && lab < c) {
            loss += l - row[lab];
            cnt += 1.f;
        }
    }
    // fixed-order block reduction: wave butterfly, then the four wave sums in order
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        loss += __shfl_xor(loss, d, 64);
        cnt += __shfl_xor(cnt, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
    doda_sync();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < CE_BLOCK / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        partial[2 * blockIdx.x] = a;
        partial[2 * blockIdx.x + 1] = b;
    }
}

// out[0] = sum(loss) / max(n_valid, 1), out[1] = n_valid
__global__ __launch_bounds__(64) void ce_final(const float *__restrict__ partial, int nblocks, float *__restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) { a += (double)partial[2 * k]; b += (double)partial[2 * k + 1]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (threadIdx.x == 0) {
        out[0] = (float)(a / (b > 1.0 ? b : 1.0));
        out[1] = (float)b;
    }
}

__global__ __launch_bounds__(CE_BLOCK) void ce_bwd(const float *__restrict__ logits,
                                                   const long long *__restrict__ labels,
                                                   const float *__restrict__ lse, const float *__restrict__ out,
                                                   const float *__restrict__ grad, long long n_elem, int c,
                                                   long long ignore_index, float *__restrict__ dlogits) {
    const float nv = out[1] > 1.f ? out[1] : 1.f;
    const float scale = grad[0] / nv;
    for (long long e = (long long)blockIdx.x * CE_BLOCK + threadIdx.x; e < n_elem; e += (long long)gridDim.x * CE_BLOCK) {
        const long long t = e / c;
        const int k = (int)(e - t * c);
        const long long lab = labels[t];
        float g = 0.f;
        if (lab != ignore_index && lab >= 0 && lab < c)
            g = (__expf(logits[e] - lse[t]) - (k == (int)lab ? 1.f : 0.f)) * scale;
        dlogits[e] = g;
    }
}

int ce_blocks(int n) {
    int nb = div_up(n, CE_BLOCK * 4);
    if (nb > 2048) nb = 2048;
    return nb < 1 ? 1 : nb;
}
}  // namespace

extern "C" size_t doda_cross_entropy_workspace_bytes(int32_t n) {
    return align_up((size_t)ce_blocks(n > 0 ? n : 1) * 2 * 4, 256);
}

extern "C" int doda_cross_entropy_fwd(const float *logits, const int64_t *labels, int32_t n, int32_t c,
                                      int64_t ignore_index, float *lse, float *out, void *ws, size_t ws_bytes,
                                      doda_stream_t stream) {
    if (n < 0 || c <= 0 || !out) return DODA_ERR_INVALID;
    if (c > CE_MAX_C) return DODA_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    if (n == 0) { hipMemsetAsync(out, 0, 8, s); return DODA_OK; }
    if (!logits || !labels || !lse || !ws) return DODA_ERR_INVALID;
    const int nb = ce_blocks(n);
    if (ws_bytes < (size_t)nb * 8) return DODA_ERR_WORKSPACE;
    hipLaunchKernelGGL(ce_fwd, dim3(nb), dim3(CE_BLOCK), 0, s, logits, (const long long *)labels, n, c,
                       (long long)ignore_index, lse, (float *)ws);
    hipLaunchKernelGGL(ce_final, dim3(1), dim3(64), 0, s, (const float *)ws, nb, out);
    return doda_check_launch();
}

extern "C" int doda_cross_entropy_bwd(const float *logits, const int64_t *labels, const float *lse,
                                      const float *out, const float *grad, int32_t n, int32_t c,
                                      int64_t ignore_index, float *dlogits, doda_stream_t stream) {
    if (n < 0 || c <= 0) return DODA_ERR_INVALID;
    if (n == 0) return DODA_OK;
    if (!logits || !labels || !lse || !out || !grad || !dlogits) return DODA_ERR_INVALID;
    const long long n_elem = (long long)n * c;
    long long nb = div_up(n_elem, (long long)CE_BLOCK * 4);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(ce_bwd, dim3((unsigned)nb), dim3(CE_BLOCK), 0, as_stream(stream), logits,
                       (const long long *)labels, lse, out, grad, n_elem, c, (long long)ignore_index, dlogits);
    return doda_check_launch();
}

// ---- segmentation meters (ABI 12) ----------------------------------------------------------------------------------------------
// reference util/common_utils.py:233-246 intersectionAndUnionGPU (called per iteration through update_meter, :249; tool/test.py:82):
// three class histograms over the points — intersection, prediction area, target area — which the reference takes with torch.histc
// on the CPU (three device-to-host copies per iteration) and this repo's harness took with three scatter_add_ over 800 k points onto
// 21 addresses (612 us per iteration: the whole difference between `python -m doda_amd.train` and the resident-batch step).  Here:
// one pass, per-workgroup histograms in LDS, 3 k global integer atomics per workgroup.  Integer counts: order-independent, exact.
//   hist[0][c] += #{valid p: pred_p == label_p == c},  hist[1][c] += #{valid p: pred_p == c},  hist[2][c] += #{valid p: label_p == c}
// valid: label != ignore_index and 0 <= label < k; pred_p = clamp(preds[p2v ? p2v[p] : p], 0, k - 1) (preds int32 or int64).
namespace {
constexpr int SM_MAX_K = 256;
template <bool P64>
__global__ __launch_bounds__(256) void seg_meters(const void *__restrict__ preds, const int32_t *__restrict__ p2v,
                                                  const long long *__restrict__ labels, int n, int k, long long ignore_index,
                                                  unsigned long long *__restrict__ hist) {
    __shared__ unsigned h[3][SM_MAX_K];
    for (int e = threadIdx.x; e < 3 * SM_MAX_K; e += 256) (&h[0][0])[e] = 0u;
    doda_sync();
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
        const long long t = labels[p];
        if (t == ignore_index || t < 0 || t >= k) continue;
        const long long src = p2v ? (long long)p2v[p] : p;
        long long q = P64 ? ((const long long *)preds)[src] : (long long)((const int32_t *)preds)[src];
        q = q < 0 ? 0 : (q >= k ? k - 1 : q);
        if (q == t) atomicAdd(&h[0][(int)t], 1u);
        atomicAdd(&h[1][(int)q], 1u);
        atomicAdd(&h[2][(int)t], 1u);
    }
    doda_sync();
    for (int e = threadIdx.x; e < 3 * k; e += 256) {
        const unsigned v = h[e / k][e % k];
        if (v) atomicAdd(hist + e, (unsigned long long)v);
    }
}
}  // namespace

extern "C" int doda_seg_meters(const void *preds, int32_t preds_are_int64, const int32_t *p2v, const int64_t *labels, int32_t n,
                               int32_t k, int64_t ignore_index, int64_t *hist, doda_stream_t stream) {
    if (n < 0 || k <= 0 || !hist) return DODA_ERR_INVALID;
    if (k > SM_MAX_K) return DODA_ERR_UNSUPPORTED;
    if (n == 0) return DODA_OK;
    if (!preds || !labels) return DODA_ERR_INVALID;
    int nb = (int)div_up((long long)n, 256ll * 4);
    if (nb > 1024) nb = 1024;
    if (preds_are_int64)
        hipLaunchKernelGGL(seg_meters<true>, dim3(nb), dim3(256), 0, as_stream(stream), preds, p2v, (const long long *)labels, n, k,
                           (long long)ignore_index, (unsigned long long *)hist);
    else
        hipLaunchKernelGGL(seg_meters<false>, dim3(nb), dim3(256), 0, as_stream(stream), preds, p2v, (const long long *)labels, n, k,
                           (long long)ignore_index, (unsigned long long *)hist);
    return doda_check_launch();
}
