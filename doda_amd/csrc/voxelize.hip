// Voxel feature pooling: PG_OP.voxelize_fp / voxelize_bp / point_recover_fp / point_recover_bp.
// Semantics follow lib/pointgroup_ops/src/voxelize/voxelize.cu:10-53 of the reference:
//   out[r,c] = ((0 + m*f[p1,c]) + m*f[p2,c]) + ...   each product rounded before the add,
// so the arithmetic is written with explicit round-to-nearest mul/add (no FMA contraction).
//
// HBM-bound integer/gather work.  One lane per (row, plane) element: lanes of a wave cover
// consecutive output elements (coalesced stores), the rule row is shared through L1.
// Algorithmic bytes per voxel row: 4*(1+maxActive) (rule) + 4*C*(cnt+1) (features in, row out).
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void voxel_pool_fwd(const float *__restrict__ feats,
                                                      float *__restrict__ out,
                                                      const int32_t *__restrict__ rules,
                                                      int n_rows, int row_w, int n_plane,
                                                      int average) {
    const long long total = (long long)n_rows * n_plane;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e / n_plane), plane = (int)(e - (long long)row * n_plane);
        const int32_t *r = rules + (long long)row * row_w;
        const int cnt = r[0];
        const float mult = (average && cnt > 0) ? __fdiv_rn(1.0f, (float)cnt) : 1.0f;
        float acc = out[e];
        for (int i = 1; i <= cnt; ++i) {
            const float v = feats[(long long)r[i] * n_plane + plane];
            acc = __fadd_rn(acc, __fmul_rn(mult, v));
        }
        out[e] = acc;
    }
}

// d_feats[p_i, c] += mult * d_out[r, c].  atomicAdd keeps the reference's accumulate-into
// semantics when a caller hands in a non-zero d_feats or a map that repeats a point.
__global__ __launch_bounds__(256) void voxel_pool_bwd(const float *__restrict__ d_out,
                                                      float *__restrict__ d_feats,
                                                      const int32_t *__restrict__ rules,
                                                      int n_rows, int row_w, int n_plane,
                                                      int average) {
    const long long total = (long long)n_rows * n_plane;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e / n_plane), plane = (int)(e - (long long)row * n_plane);
        const int32_t *r = rules + (long long)row * row_w;
        const int cnt = r[0];
        const float mult = (average && cnt > 0) ? __fdiv_rn(1.0f, (float)cnt) : 1.0f;
        const float g = __fmul_rn(mult, d_out[e]);
        for (int i = 1; i <= cnt; ++i) atomicAdd(&d_feats[(long long)r[i] * n_plane + plane], g);
    }
}

int launch_pool(bool fwd, const float *src, float *dst, const int32_t *rules, int average,
                int n_active, int max_active, int n_plane, doda_stream_t stream) {
    if (n_active < 0 || max_active < 0 || n_plane < 0) return DODA_ERR_INVALID;
    if (n_active == 0 || n_plane == 0) return DODA_OK;
    if (!src || !dst || !rules) return DODA_ERR_INVALID;
    const long long total = (long long)n_active * n_plane;
    const int grid = (int)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    if (fwd)
        hipLaunchKernelGGL(voxel_pool_fwd, dim3(grid), dim3(256), 0, as_stream(stream), src, dst,
                           rules, n_active, max_active + 1, n_plane, average);
    else
        hipLaunchKernelGGL(voxel_pool_bwd, dim3(grid), dim3(256), 0, as_stream(stream), src, dst,
                           rules, n_active, max_active + 1, n_plane, average);
    return doda_check_launch();
}
}  // namespace

extern "C" int doda_voxelize_fp(const float *feats, float *out, const int32_t *rules, int32_t mode,
                                int32_t n_active, int32_t max_active, int32_t n_plane,
                                doda_stream_t stream) {
    return launch_pool(true, feats, out, rules, mode == 4, n_active, max_active, n_plane, stream);
}

extern "C" int doda_voxelize_bp(const float *d_out, float *d_feats, const int32_t *rules,
                                int32_t mode, int32_t n_active, int32_t max_active,
                                int32_t n_plane, doda_stream_t stream) {
    return launch_pool(false, d_out, d_feats, rules, mode == 4, n_active, max_active, n_plane,
                       stream);
}

// voxelize.cpp:184-205: point_recover_fp == voxelize_bp kernel, average=false, (voxel feats ->
// point feats); point_recover_bp == voxelize_fp kernel, average=false.
extern "C" int doda_point_recover_fp(const float *feats, float *out, const int32_t *rules,
                                     int32_t n_active, int32_t max_active, int32_t n_plane,
                                     doda_stream_t stream) {
    return launch_pool(false, feats, out, rules, 0, n_active, max_active, n_plane, stream);
}

extern "C" int doda_point_recover_bp(const float *d_out, float *d_feats, const int32_t *rules,
                                     int32_t n_active, int32_t max_active, int32_t n_plane,
                                     doda_stream_t stream) {
    return launch_pool(true, d_out, d_feats, rules, 0, n_active, max_active, n_plane, stream);
}
