// Indice-pair max pooling on the gather table (spconv v1.2 indice_maxpool /
// indice_maxpool_backward; not called by DODA, named by north_star).  Output-stationary like
// the convolutions: one lane per (output row, channel), each output written once.
// Upstream semantics: out starts at 0 and is max-ed with every paired input
// (out = max(out, in)), so an output with pairs never drops below 0; backward routes dy to
// every paired input whose value equals the pooled output.
#include "common.hpp"

namespace {
__global__ __launch_bounds__(256) void maxpool_fwd(const float *__restrict__ x, int c,
                                                   const int32_t *__restrict__ tbl, int ld, int K,
                                                   int n_out, float *__restrict__ y) {
    const long long total = (long long)n_out * c;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(e / c), ch = (int)(e - (long long)t * c);
        float m = 0.f;
        for (int o = 0; o < K; ++o) {
            const int j = tbl[(long long)o * ld + t];
            if (j >= 0) {
                const float v = x[(long long)j * c + ch];
                m = v > m ? v : m;
            }
        }
        y[e] = m;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd(const float *__restrict__ x,
                                                   const float *__restrict__ y,
                                                   const float *__restrict__ dy, int c,
                                                   const int32_t *__restrict__ tbl, int ld, int K,
                                                   int n_out, float *__restrict__ dx) {
    const long long total = (long long)n_out * c;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(e / c), ch = (int)(e - (long long)t * c);
        const float m = y[e], g = dy[e];
        for (int o = 0; o < K; ++o) {
            const int j = tbl[(long long)o * ld + t];
            if (j >= 0 && x[(long long)j * c + ch] == m) atomicAdd(&dx[(long long)j * c + ch], g);
        }
    }
}
}  // namespace

extern "C" int doda_maxpool_fwd_f32(const float *x, int32_t c, const int32_t *tbl, int32_t ld,
                                    int32_t K, int32_t n_out, float *y, doda_stream_t stream) {
    if (c <= 0 || K <= 0 || n_out < 0 || ld < n_out) return DODA_ERR_INVALID;
    if (n_out == 0) return DODA_OK;
    if (!x || !tbl || !y) return DODA_ERR_INVALID;
    const long long total = (long long)n_out * c;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_fwd, dim3(grid), dim3(256), 0, as_stream(stream), x, c, tbl, ld, K,
                       n_out, y);
    return doda_check_launch();
}

extern "C" int doda_maxpool_bwd_f32(const float *x, const float *y, const float *dy, int32_t c,
                                    const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out,
                                    float *dx, doda_stream_t stream) {
    if (c <= 0 || K <= 0 || n_out < 0 || ld < n_out) return DODA_ERR_INVALID;
    if (n_out == 0) return DODA_OK;
    if (!x || !y || !dy || !tbl || !dx) return DODA_ERR_INVALID;
    const long long total = (long long)n_out * c;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_bwd, dim3(grid), dim3(256), 0, as_stream(stream), x, y, dy, c, tbl,
                       ld, K, n_out, dx);
    return doda_check_launch();
}
