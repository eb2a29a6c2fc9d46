// Device helpers shared by the sparse-convolution translation units (spconv_gather.hip: dense-table kernels and
// the C ABI; spconv_tile.hip: the LDS-staged kernels over a tilebook).  gfx950 only.
#pragma once
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    // gfx950: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) — the integer form cost ten instructions and an
    // EXEC round trip per value in the store epilogues
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ void mma_bf16_k32(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
// 16 fp32 channels: lane (i, g) holds channels 4g..4g+3; four v_mfma_f32_16x16x4_f32 (an exact fmaf chain)
__device__ __forceinline__ void mma_f32_k16(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xf[q], acc, 0, 0, 0);
}

// inclusive prefix sum along the 16 lanes of a DPP row (row_shr with zero fill): lane 15 ends up with
// the row's total.  Plain VALU adds in a fixed order — no LDS traffic, deterministic.
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));
    return v;
}

}  // namespace

// BatchNorm statistics riding in the store epilogue (reference model/unet_block.py:
// 23-30,46-49,67-79: every conv sits between two BatchNorm1d+ReLU pairs).  Per workgroup row tile p:
//   forward call  : stats[p][0][c] = sum_t y[t,c],   stats[p][1][c] = sum_t y[t,c]^2     (BN after the conv)
//   data-grad call: stats[p][0][c] = sum_t dz[t,c],  stats[p][1][c] = sum_t dz[t,c] * xhat[t,c]
//                   with dz = y * [gamma*xhat + beta > 0] (when bn_relu), xhat = (bn_x - mean) * invstd:
//                   the two sums the backward of the BN(+ReLU) BEFORE the conv needs over dy = y.
// y is taken as stored (after its bf16 rounding), so a standalone pass over the stored tensor would see
// the same values.  One writer per (p, c), fixed summation order: deterministic.
struct EpiArgs {   // plain data, shared across translation units
    float *stats;            // [n_part][2][nc] or null
    const void *bn_x;        // [n_out, nc] in the dtype of y, or null (forward statistics)
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   // [nc] each
    int bn_relu;
    int res_bcast;           // ABI 6: `res` is ONE row [nc] added to every output row (a bias): conv_fast only
};

namespace doda_tile {
bool enabled();   // doda_set_option(DODA_OPT_TILE_KERNEL)
void set_enabled(bool on);
bool pipeline_enabled();   // doda_set_option(DODA_OPT_TILE_PIPELINE): conv_tile16 for 16 -> 16 layers of many tiles
void set_pipeline(bool on);
bool dual_enabled();       // doda_set_option(DODA_OPT_TILE_DUAL): both channel blocks of a 32-output-channel layer in one pass
void set_dual(bool on);
bool up_enabled();         // doda_set_option(DODA_OPT_CONV_UP): conv_up32 for one-source-per-row tables
void set_up(bool on);
// conv_tile over `tilebook` (doda_tilebook_build of tbl).  mode 0: bf16 16 channels, 1: bf16 32 channels, 2: fp32 16
// channels; out32: fp32 output rows.  *n_part (if given) receives the number of statistics rows.
int launch_conv_tile(int mode, bool out32, const void *x, unsigned x_bytes, const void *wp, unsigned wp_bytes, int nc, int NB,
                     const int32_t *tbl, int ld, int n_out, const void *tilebook, void *y, unsigned y_bytes, const void *res,
                     const EpiArgs &ep, int *n_part, hipStream_t s);
// conv_up32: bf16, 32 input channels, K <= 8, a table with (about) one source row per output row (inverse convolution forward,
// strided convolution data gradient); wp = wide-packed fragments [o][nb][64] x 16 B.  *n_part: statistics rows (one per 256 rows).
int launch_conv_up32(bool out32, const void *x, unsigned x_bytes, const void *wp, unsigned wp_bytes, int nc, int NB, int K,
                     const int32_t *tbl, int ld, int n_out, void *y, unsigned y_bytes, const void *res, const EpiArgs &ep,
                     int *n_part, hipStream_t s);
}  // namespace doda_tile

namespace doda_wlds {
bool enabled();
void set_enabled(bool on);
// bf16 48 -> 48 channels, K = 27, weights in LDS (spconv_wlds.hip); wp = wide-packed fragments [27][2][3][64] x 16 B
int launch_conv48(const void *x, unsigned x_bytes, const void *wp, const int32_t *tbl, unsigned tbl_bytes, int ld, int n_out,
                  void *y, unsigned y_bytes, const void *res, const EpiArgs &ep, int *n_part, hipStream_t s);
}  // namespace doda_wlds
