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
// 16 fp32 channels: lane (i, g) holds channels 4g..4g+3 of a weight column / a feature row.
//
// Round 5: the fp32 matrix rate of this part is 1/16 of the bf16 rate (157 TFLOP/s; a 16-channel unit is
// four dependent v_mfma_f32_16x16x4_f32 = 128 cycles), and the fp32 kernels sat on exactly that chain.  Both operands are
// split IN REGISTERS into a bf16 head and a bf16 tail (x = hi + lo + e, |e| <= 2^-17 |x|: hi = RNE(x), lo = RNE(x - hi), the
// subtraction exact) and the unit becomes TWO v_mfma_f32_16x16x32_bf16 (32 cycles): the k = 32 of one instruction is
// (4 channels of the lane group) x (head, tail) — A = [w_hi | w_lo] against B = [x_hi | x_hi], then against [x_lo | x_lo]:
// all four partial products, fp32 accumulate.  The gathered bytes do not change (fp32 rows, fp32 fragment buffer).  Error of a
// product <= ~2^-16 relative (north_star: 1e-4 on fp32 features).  OPT-IN per launch (DODA_F32_SPLIT_ROWS /
// DODA_F32_WGRAD_SPLIT_ROWS, see run_gather): measured -6 % on the fp32 step, but the U-Net's gradients move from ~1e-3 to
// ~7e-3 (elementwise, relative) away from the fp64 golden, and fp32 is the parity precision here.  DODA_F32_EXACT_MFMA=1 at
// build time removes the split instantiations altogether.
#ifndef DODA_F32_EXACT_MFMA
#define DODA_F32_EXACT_MFMA 0
#endif
struct SplitBf16 { unsigned hi[2], lo[2]; };   // 4 values: heads packed (v0,v1),(v2,v3), tails likewise
__device__ __forceinline__ SplitBf16 split_bf16x4(const f32x4 &v) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    SplitBf16 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 a = {v[2 * h], v[2 * h + 1]};
        const bf16x2 hb = __builtin_convertvector(a, bf16x2);                 // v_cvt_pk_bf16_f32: round to nearest even
        const unsigned hu = __builtin_bit_cast(unsigned, hb);
        const f32x2 res = {a[0] - __uint_as_float(hu << 16), a[1] - __uint_as_float(hu & 0xffff0000u)};   // exact
        r.hi[h] = hu;
        r.lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(res, bf16x2));
    }
    return r;
}
// SPLIT is a compile-time choice (a run-time branch inside conv_fast's ring of hand-counted asm loads made hipcc copy
// registers whose loads were still in flight: 49 parity tests failed); the host picks the instantiation per launch
// (EpiArgs::f32_split: layers of many rows, where the matrix pipe is what the kernel waits for and thousands of rows average
// the rounding; small layers keep the exact chain).
template <bool SPLIT = false>
__device__ __forceinline__ void mma_f32_k16(f32x4 &acc, const u32x4 &w, const u32x4 &x) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
    if constexpr (DODA_F32_EXACT_MFMA || !SPLIT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[q], xf[q], acc, 0, 0, 0);
    } else {
        const SplitBf16 ws = split_bf16x4(wf), xs = split_bf16x4(xf);
        const u32x4 a = {ws.hi[0], ws.hi[1], ws.lo[0], ws.lo[1]};
        const u32x4 b1 = {xs.hi[0], xs.hi[1], xs.hi[0], xs.hi[1]}, b2 = {xs.lo[0], xs.lo[1], xs.lo[0], xs.lo[1]};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b2), acc, 0, 0, 0);
    }
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
    int f32_split;           // round 5: fp32 units as two bf16 MFMAs on head / tail splits (mma_f32_k16); set by run_gather
    double *stats_tot;       // ABI 9: [DODA_STATS_SLOTS][2][nc / 4][16] totals (4 of 16 used), accumulated with fp64 atomics INSTEAD of the rows, or null
};
#ifndef DODA_STATS_SLOTS
#define DODA_STATS_SLOTS 8      // (include/doda_hip.h)
#endif
#if defined(__HIPCC__)
// A workgroup's (sum, sum of squares) of the four columns col .. col + 3: its row of `stats`, or — ABI 9 — added to the
// totals of slot (part mod 8).  The additions are hardware fp64 atomics without return (fire and forget: the workgroup does
// not wait, the kernel's end does); every addend is an fp32 value, so a sum of a few thousand of them is exact in fp64 unless
// their magnitudes span more than ~2^16, and then differs from any other order by an ulp of fp64 — the statistics derived
// from the totals are fp32.
__device__ __forceinline__ void stats_emit(const EpiArgs &ep, long long part, int nc, int col, const f32x4 &a1, const f32x4 &a2) {
    if (ep.stats_tot) {
// (layout: every group of four channels of a (slot, sum) pair owns a 128-byte line — 16 doubles, the first four used.
        // With the 2 x nc doubles of a slot packed, the 64 workgroups of a slot sent ~1000 atomics to each line and the L2
        // serialised them: +1.7 us at the end of conv_tile16; padded, the kernel is as fast as with rows.)
        const size_t k = (size_t)((unsigned)part & (unsigned)(DODA_STATS_SLOTS - 1));
        double *t1 = ep.stats_tot + ((k * 2 + 0) * (size_t)(nc / 4) + (size_t)(col / 4)) * 16;
        double *t2 = ep.stats_tot + ((k * 2 + 1) * (size_t)(nc / 4) + (size_t)(col / 4)) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsafeAtomicAdd(t1 + q, (double)a1[q]);
            unsafeAtomicAdd(t2 + q, (double)a2[q]);
        }
    } else {
        float *dst = ep.stats + part * 2 * nc + col;
        *reinterpret_cast<f32x4 *>(dst) = a1;
        *reinterpret_cast<f32x4 *>(dst + nc) = a2;
    }
}
#endif

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
