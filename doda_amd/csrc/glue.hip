// Small element-wise kernels of the step's head and input layer that torch would run as two kernels each.
//
// cast_colsum: the Linear head's backward (reference model/unet.py:64 `self.linear`, through autograd) needs the score
//   gradient twice — as the bf16 operand of the two gather-GEMMs (d_feats, d_W) and summed over the points for d_bias:
//   torch: a cast kernel + a reduce kernel, each streaming the fp32 [N, C] matrix (64 MB at 800k points x 20 classes).
//   Here one pass writes the bf16 copy and per-workgroup column sums (fixed order: deterministic); the caller adds the
//   few hundred partial rows.
// pad_channels: the xyz input layer (3 or 6 channels) runs on rows zero-padded to 4 / 16 channels (spconv/conv.py): torch's
//   constant_pad_nd is a fill + a strided copy; here one kernel.
#include "common.hpp"

namespace {
constexpr int GL_BLOCK = 256;
constexpr int GL_MAX_C = 64;

__device__ __forceinline__ unsigned short gl_f2bf(float f) {
    // gfx950: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) — the integer form cost ten instructions and an
    // EXEC round trip per value in the store epilogues
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

// x fp32 [n, c] -> y bf16 [n, c] (torch's round-to-nearest-even cast), partial[block][c] = sum over the block's rows
__global__ __launch_bounds__(GL_BLOCK) void cast_colsum(const float *__restrict__ x, long long n, int c,
                                                        unsigned short *__restrict__ y, float *__restrict__ partial) {
    // thread = (row lane tid / c, column tid % c): consecutive threads read consecutive elements; rows stride by the row
    // lanes of the whole grid
    const int lanes = GL_BLOCK / c;              // rows per block sweep (c <= 64 -> >= 4)
    const int col = threadIdx.x % c, rl = threadIdx.x / c;
    float s = 0.f;
    if (rl < lanes) {
        for (long long r = (long long)blockIdx.x * lanes + rl; r < n; r += (long long)gridDim.x * lanes) {
            const float v = x[r * c + col];
            y[r * c + col] = gl_f2bf(v);
            s += v;
        }
    }
    // fixed-order reduce over the row lanes: through LDS, column by column
    __shared__ float acc[GL_BLOCK];
    acc[threadIdx.x] = rl < lanes ? s : 0.f;
    doda_sync();
    if (threadIdx.x < c) {
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += acc[k * c + threadIdx.x];
        partial[(long long)blockIdx.x * c + threadIdx.x] = t;
    }
}

// y[r, 0:c_in] = x[r, :], y[r, c_in:c_out] = 0; elements of 2 or 4 bytes.  One thread per 16-byte piece of an output row
// (the rows are 8 - 64 bytes long): its 4 or 8 elements are gathered from the input row, one 16-byte store.
// Requires c_out * esz to be a multiple of 16 and y 16-byte aligned (the launcher checks; else the scalar form).
template <int ESZ>
__global__ __launch_bounds__(GL_BLOCK) void pad_channels16(const unsigned char *__restrict__ x, long long n, int c_in, int c_out,
                                                           unsigned char *__restrict__ y) {
    constexpr int EPP = 16 / ESZ;                       // elements per piece
    const int ppr = c_out / EPP;                        // pieces per row
    const long long total = n * ppr;
    for (long long e = (long long)blockIdx.x * GL_BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * GL_BLOCK) {
        const long long r = e / ppr;
        const int c0 = (int)(e - r * ppr) * EPP;
        typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ v = {0u, 0u, 0u, 0u};
        if constexpr (ESZ == 2) {
            const unsigned short *src = reinterpret_cast<const unsigned short *>(x) + r * c_in;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < c_in) v[k >> 1] |= (unsigned)src[c0 + k] << (16 * (k & 1));
        } else {
            const unsigned int *src = reinterpret_cast<const unsigned int *>(x) + r * c_in;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < c_in) v[k] = src[c0 + k];
        }
        reinterpret_cast<u32x4_ *>(y)[e] = v;
    }
}
__global__ __launch_bounds__(GL_BLOCK) void pad_channels(const unsigned char *__restrict__ x, long long n, int c_in, int c_out,
                                                         int esz, unsigned char *__restrict__ y) {
    const long long total = n * c_out;
    for (long long e = (long long)blockIdx.x * GL_BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * GL_BLOCK) {
        const long long r = e / c_out;
        const int col = (int)(e - r * c_out);
        if (esz == 2) {
            reinterpret_cast<unsigned short *>(y)[e] = col < c_in ? reinterpret_cast<const unsigned short *>(x)[r * c_in + col] : (unsigned short)0;
        } else {
            reinterpret_cast<unsigned int *>(y)[e] = col < c_in ? reinterpret_cast<const unsigned int *>(x)[r * c_in + col] : 0u;
        }
    }
}
}  // namespace

extern "C" int32_t doda_cast_colsum_blocks(int64_t n, int32_t c) {
    if (n <= 0 || c <= 0 || c > GL_MAX_C) return 0;
    const long long lanes = GL_BLOCK / c;
    long long b = (n + lanes * 16 - 1) / (lanes * 16);   // ~16 rows per thread
    if (b > 1024) b = 1024;
    return (int32_t)(b < 1 ? 1 : b);
}

extern "C" int doda_cast_colsum_f32_bf16(const float *x, int64_t n, int32_t c, uint16_t *y, float *partial,
                                         int32_t n_blocks, doda_stream_t stream) {
    if (n == 0) return DODA_OK;
    if (n < 0 || c <= 0 || c > GL_MAX_C) return DODA_ERR_UNSUPPORTED;
    if (!x || !y || !partial || n_blocks != doda_cast_colsum_blocks(n, c)) return DODA_ERR_INVALID;
    hipLaunchKernelGGL(cast_colsum, dim3(n_blocks), dim3(GL_BLOCK), 0, as_stream(stream), x, (long long)n, (int)c,
                       (unsigned short *)y, partial);
    return doda_check_launch();
}

extern "C" int doda_pad_channels(const void *x, int64_t n, int32_t c_in, int32_t c_out, int32_t elem_bytes, void *y,
                                 doda_stream_t stream) {
    if (n == 0) return DODA_OK;
    if (n < 0 || c_in <= 0 || c_out < c_in || (elem_bytes != 2 && elem_bytes != 4)) return DODA_ERR_INVALID;
    if (!x || !y) return DODA_ERR_INVALID;
    if (((long long)c_out * elem_bytes) % 16 == 0 && ((uintptr_t)y & 15) == 0) {
        const long long pieces = (long long)n * (c_out * elem_bytes / 16);
        long long grid = (pieces + GL_BLOCK - 1) / GL_BLOCK;
        if (grid > 8192) grid = 8192;
        if (elem_bytes == 2)
            hipLaunchKernelGGL(pad_channels16<2>, dim3((unsigned)grid), dim3(GL_BLOCK), 0, as_stream(stream),
                               (const unsigned char *)x, (long long)n, (int)c_in, (int)c_out, (unsigned char *)y);
        else
            hipLaunchKernelGGL(pad_channels16<4>, dim3((unsigned)grid), dim3(GL_BLOCK), 0, as_stream(stream),
                               (const unsigned char *)x, (long long)n, (int)c_in, (int)c_out, (unsigned char *)y);
        return doda_check_launch();
    }
    const long long total = (long long)n * c_out;
    long long grid = (total + GL_BLOCK * 8 - 1) / (GL_BLOCK * 8);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pad_channels, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(GL_BLOCK), 0, as_stream(stream),
                       (const unsigned char *)x, (long long)n, (int)c_in, (int)c_out, (int)elem_bytes, (unsigned char *)y);
    return doda_check_launch();
}
