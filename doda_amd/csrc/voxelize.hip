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

// The network's input rows in one launch: the pooled row of voxel r over up to two point-feature matrices side by side
// (rgb | xyz: model/unet.py:89-91 concatenates them before pooling), rounded to the network's feature type and zero-padded to
// c_out channels -- what voxelize_fp + cast + the input layer's channel padding produce in four launches (zero fill, pool,
// fp32 -> bf16 copy, pad).  Same arithmetic as voxel_pool_fwd (acc starts at 0: the wrapper's zero-initialised output);
// bf16 = round to nearest even of the fp32 result (torch's .to(bfloat16)).  One lane per (row, output channel).
template <typename OUT>
__global__ __launch_bounds__(256) void voxel_pool_rows(const float *__restrict__ fa, int ca, const float *__restrict__ fb, int cb,
                                                       const int32_t *__restrict__ rules, int n_rows, int row_w, int average,
                                                       OUT *__restrict__ out, int c_out) {
    const long long total = (long long)n_rows * c_out;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(e / c_out), plane = (int)(e - (long long)row * c_out);
        float acc = 0.0f;
        if (plane < ca + cb) {
            const int32_t *r = rules + (long long)row * row_w;
            const int cnt = r[0];
            const float mult = (average && cnt > 0) ? __fdiv_rn(1.0f, (float)cnt) : 1.0f;
            const bool first = plane < ca;
            const float *src = first ? fa + plane : fb + (plane - ca);
            const int ld = first ? ca : cb;
            for (int i = 1; i <= cnt; ++i) acc = __fadd_rn(acc, __fmul_rn(mult, src[(long long)r[i] * ld]));
        }
        if constexpr (sizeof(OUT) == 2) out[e] = __builtin_bit_cast(unsigned short, (__bf16)acc);
        else out[e] = acc;
    }
}

// One lane per voxel row (up to 8 pooled channels, rows of whole 16-byte pieces): the rule row is read once per row instead of
// once per channel, 64 consecutive rows per wave leave as one contiguous store.  (The (row, channel) form above kept 3 of 16
// lanes busy at 3 -> 16 channels: 91 us at 600 k voxels against 28 us for the pooling alone.)
template <typename OUT>
__global__ __launch_bounds__(256) void voxel_pool_rows_lane(const float *__restrict__ fa, int ca, const float *__restrict__ fb, int cb,
                                                            const int32_t *__restrict__ rules, int n_rows, int row_w, int average,
                                                            OUT *__restrict__ out, int c_out) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    constexpr int EPP = 16 / (int)sizeof(OUT);      // elements per 16-byte piece
    const int pieces = c_out / EPP;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += gridDim.x * blockDim.x) {
        const int32_t *r = rules + (long long)row * row_w;
        const int cnt = r[0];
        const float mult = (average && cnt > 0) ? __fdiv_rn(1.0f, (float)cnt) : 1.0f;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
        for (int i = 1; i <= cnt; ++i) {
            const long long p = r[i];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < ca) acc[c] = __fadd_rn(acc[c], __fmul_rn(mult, fa[p * ca + c]));
                else if (c < ca + cb) acc[c] = __fadd_rn(acc[c], __fmul_rn(mult, fb[p * cb + (c - ca)]));
            }
        }
        u32x4_ *o = reinterpret_cast<u32x4_ *>(out + (long long)row * c_out);
        if constexpr (sizeof(OUT) == 2) {
            u32x4_ v;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[2 * k]) |
                       ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[2 * k + 1]) << 16);
            o[0] = v;
            for (int k = 1; k < pieces; ++k) o[k] = (u32x4_){0u, 0u, 0u, 0u};
        } else {
            o[0] = (u32x4_){__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
            if (pieces > 1) o[1] = (u32x4_){__float_as_uint(acc[4]), __float_as_uint(acc[5]), __float_as_uint(acc[6]), __float_as_uint(acc[7])};
            for (int k = 2; k < pieces; ++k) o[k] = (u32x4_){0u, 0u, 0u, 0u};
        }
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

extern "C" int doda_voxelize_fp_rows(const float *feats_a, int32_t c_a, const float *feats_b, int32_t c_b, const int32_t *rules,
                                     int32_t mode, int32_t n_active, int32_t max_active, void *out, int32_t c_out,
                                     int32_t out_elem_bytes, doda_stream_t stream) {
    if (n_active < 0 || max_active < 0 || c_a < 1 || c_b < 0 || c_out < c_a + c_b || (out_elem_bytes != 2 && out_elem_bytes != 4) ||
        (c_b > 0 && !feats_b))
        return DODA_ERR_INVALID;
    if (n_active == 0) return DODA_OK;
    if (!feats_a || !rules || !out) return DODA_ERR_INVALID;
    if (c_a + c_b <= 8 && (c_out * out_elem_bytes) % 16 == 0 && (c_out * out_elem_bytes >= 32 || c_a + c_b <= 16 / out_elem_bytes) &&
        ((uintptr_t)out & 15) == 0) {
        const int grid = (n_active + 255) / 256 < 256 * 16 ? (n_active + 255) / 256 : 256 * 16;
        if (out_elem_bytes == 2)
            hipLaunchKernelGGL(voxel_pool_rows_lane<unsigned short>, dim3(grid), dim3(256), 0, as_stream(stream), feats_a, (int)c_a,
                               feats_b, (int)c_b, rules, (int)n_active, (int)max_active + 1, (int)(mode == 4), (unsigned short *)out,
                               (int)c_out);
        else
            hipLaunchKernelGGL(voxel_pool_rows_lane<float>, dim3(grid), dim3(256), 0, as_stream(stream), feats_a, (int)c_a, feats_b,
                               (int)c_b, rules, (int)n_active, (int)max_active + 1, (int)(mode == 4), (float *)out, (int)c_out);
        return doda_check_launch();
    }
    const long long total = (long long)n_active * c_out;
    const int grid = (int)((total + 255) / 256 < 256 * 32 ? (total + 255) / 256 : 256 * 32);
    if (out_elem_bytes == 2)
        hipLaunchKernelGGL(voxel_pool_rows<unsigned short>, dim3(grid), dim3(256), 0, as_stream(stream), feats_a, (int)c_a, feats_b,
                           (int)c_b, rules, (int)n_active, (int)max_active + 1, (int)(mode == 4), (unsigned short *)out, (int)c_out);
    else
        hipLaunchKernelGGL(voxel_pool_rows<float>, dim3(grid), dim3(256), 0, as_stream(stream), feats_a, (int)c_a, feats_b, (int)c_b,
                           rules, (int)n_active, (int)max_active + 1, (int)(mode == 4), (float *)out, (int)c_out);
    return doda_check_launch();
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
