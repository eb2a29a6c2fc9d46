// Device helpers shared by the sparse-convolution translation units (spconv_gather.hip: dense-table kernels and
// the C ABI; spconv_tile.hip: the LDS-staged kernels over a tilebook).  gfx950 only.
#pragma once
#include "common.hpp"
#include "bn_totals.hpp"

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
    // ABI 11: row strides in ELEMENTS of x / y / res / bn_x (0: dense — kc, nc, nc, nc): a column slice of a wider matrix, e.g. one
    // half of a U-Net level's concatenation (reference model/unet_block.py:89-93), is read / written in place.  conv_fast only.
    unsigned x_ld, y_ld, res_ld, bnx_ld;
};
// ABI 11: a BatchNorm folded into the gather of the convolution that consumes it (conv_fast<..., PRE>; reference
// model/unet_block.py:23-30,46-49,67-79: BatchNorm1d -> ReLU -> conv, and the backward of that chain).
//   kind 1 (forward): the gathered rows of x are normalised on their way into the MFMA — xn = [relu]((x - mean) * invstd * gamma +
//          beta), rounded to the storage type — with mean / invstd derived by every workgroup from the producer's fp64 totals
//          (tot.ta / tb; null: running statistics = evaluation mode); workgroup 0 publishes save_mean / save_invstd and the
//          running statistics; the launch also writes xn once per row to `side` (the weight gradient's operand).
//   kind 2 (backward): the gathered rows are du = ca * ([yv > 0] dz - cb - xhat * cd) (+ add, kind 3) computed from the rows of
//          x = dz, `aux` = the BatchNorm's input and `add`; the coefficient vectors come from the totals (sum dz, sum dz xhat)
//          the producing data-grad call accumulated; workgroup 0 writes dgamma / dbeta; `side` receives du once per row.
struct PreArgs {
    int kind, relu;
    int rows;                       // rows of x (= of side / aux / add)
    TotArgs tot;                    // (bn_totals.hpp)
    const float *gamma, *beta;
    const float *mean, *invstd;     // kind >= 2: the BatchNorm's saved vectors
    void *side;
    unsigned side_ld;
    const void *aux, *add;
    unsigned aux_ld, add_ld;
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

#if defined(__HIPCC__)
namespace {
// ---- ABI 11: BatchNorm folded into the gather (PreArgs above) ---------------------------------------------------------------
// A gathered 16-byte piece (8 bf16 / 4 fp32 values of one row) is transformed in registers between its load and its MFMA.
// `pm` = all ones for a present neighbour, zero for an absent one (an absent row contributes zeros, not relu(beta - mean ...)).
// Two forms of the arithmetic, chosen by the storage type, used by EVERY kernel of the per-layer backend that applies a BatchNorm
// below its row threshold — the folded gathers, their once-per-row side output and the standalone lay_bn sweeps (layers.hip) — so
// that folding an op never changes a bit:
//   fp32 rows: the standalone sweeps' order (bn_totals.hpp bn_fwd_elem / bn_bwd_elem: every operation rounds) — fp32 is the parity
//              precision (reference lib/pointgroup_ops/src/cuda.cu:11-13), and (x - mean) first is the well-conditioned order;
//   bf16 rows: ONE fused multiply-add per value on per-channel products — forward y = x sc + sh (sc = invstd gamma,
//              sh = beta - mean sc), backward du = [u G + H > 0] (A dz) + (E - C u) — a quarter of the vector instructions; the
//              result is rounded to bf16 (2^-9) anyway, and the kernels are paced by exactly these instructions (every gathered
//              row is transformed once per kernel offset that reads it).
constexpr int PRE_MAX_C = BN_TOT_MAX_C;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int ESZ, int KIND> struct PreForm {
    static constexpr bool FMA = ESZ == 2;
    static constexpr int NV = 16 / ESZ;                                                    // values per piece
    static constexpr int NVEC = FMA ? (KIND == 1 ? 2 : 5) : (KIND == 1 ? 4 : 7);           // per-channel vectors
};
template <int ESZ, int KIND> struct PreCo { float v[PreForm<ESZ, KIND>::NVEC][PreForm<ESZ, KIND>::NV]; };   // a lane's channels, in registers

// What a thread (= channel) reads for the vectors, requested BEFORE anything is waited for (one memory round trip for the
// statistics, the gather table and the side output's rows together: at the coarse levels a kernel IS its chain of round trips)
struct PreRaw { double d[2 * BN_TOT_SLOTS]; float ga, be, mu, is; };
template <int KIND>
__device__ __forceinline__ PreRaw pre_request(const PreArgs &pre, int kc) {
    // NO control flow around the loads — neither per lane (threads past kc re-read channel 0 and are zeroed in pre_finish) nor
    // uniform (evaluation mode reads gamma's first bytes in place of totals, training mode in place of running statistics):
    // with a branch here hipcc waits for the whole batch at the merge point, before the table and side-output loads are issued
    PreRaw r;
    const int ch = (int)threadIdx.x < kc ? (int)threadIdx.x : 0;
    r.ga = pre.gamma[ch];
    r.be = pre.beta[ch];
    const bool totals = pre.tot.ta != nullptr;
    const float *pa = KIND == 1 ? (totals || !pre.tot.rm ? pre.gamma : (const float *)pre.tot.rm) : pre.mean;
    const float *pb = KIND == 1 ? (totals || !pre.tot.rv ? pre.gamma : (const float *)pre.tot.rv) : pre.invstd;
    r.mu = pa[ch];
    r.is = pb[ch];
    const bool first = ch < pre.tot.ca;
    const double *src = totals ? (first ? pre.tot.ta : pre.tot.tb) : (const double *)pre.gamma;   // (gamma: >= 16 floats, 16-byte aligned)
    const int cw = first ? pre.tot.ca : kc - pre.tot.ca, cc = first ? ch : ch - pre.tot.ca;
    const size_t g = totals ? (size_t)(cw / 4) : 0, base = totals ? (size_t)(cc / 4) * 16 + (size_t)(cc & 3) : 0;
#pragma unroll
    for (int k = 0; k < BN_TOT_SLOTS; ++k) {      // (layout: stats_emit — a 128-byte line per four channels)
        r.d[2 * k] = src[(size_t)(k * 2 + 0) * g * 16 + base];
        r.d[2 * k + 1] = src[(size_t)(k * 2 + 1) * g * 16 + base];
    }
    return r;
}
// The vectors of channel threadIdx.x into LDS (zero past kc); workgroup 0 publishes what later kernels need — the arithmetic of
// tot_fwd_channel / tot_bwd_channel (bn_totals.hpp) on the sums in their order.
template <int ESZ, int KIND>
__device__ __forceinline__ void pre_finish(const PreArgs &pre, int kc, const PreRaw &r, float (*co)[PRE_MAX_C]) {
    typedef PreForm<ESZ, KIND> F;
    const int ch = threadIdx.x;
    if (ch >= PRE_MAX_C) return;
    float mu = 0.f, is = 0.f, ga = r.ga, be = r.be, ca = 0.f, cb = 0.f, cd = 0.f;
    if (ch < kc) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < BN_TOT_SLOTS; ++k) { s1 += r.d[2 * k]; s2 += r.d[2 * k + 1]; }
        const bool publish = blockIdx.x == 0;
        const TotArgs &t = pre.tot;
        if constexpr (KIND == 1) {
            if (t.ta) {
                const double d = s1 / t.m;
                double var = s2 / t.m - d * d;
                if (var < 0.0) var = 0.0;
                mu = (float)d;
                is = (float)(1.0 / sqrt(var + (double)t.eps));
                if (publish) {
                    t.out_a[ch] = mu;
                    t.out_b[ch] = is;
                    if (t.rm) {
                        const double unbiased = t.m > 1 ? var * (double)t.m / (double)(t.m - 1) : var;
                        t.rm[ch] = (float)((1.0 - t.momentum) * (double)t.rm[ch] + t.momentum * d);
                        t.rv[ch] = (float)((1.0 - t.momentum) * (double)t.rv[ch] + t.momentum * unbiased);
                    }
                    if (ch == 0 && t.nbt) *t.nbt = *t.nbt + 1;
                }
            } else {
                mu = r.mu;
                is = 1.0f / sqrtf(r.is + t.eps);
            }
        } else {
            mu = r.mu;
            is = r.is;
            ca = ga * is;
            cb = (float)(s1 / t.m);
            cd = (float)(s2 / t.m);
            if (publish) {
                if (t.accum) { t.out_b[ch] += (float)s1; t.out_a[ch] += (float)s2; }
                else { t.out_b[ch] = (float)s1; t.out_a[ch] = (float)s2; }      // dbeta, dgamma
            }
        }
    } else {
        ga = 0.f; be = 0.f;
    }
    if constexpr (!F::FMA) {
        co[0][ch] = mu; co[1][ch] = is; co[2][ch] = ga; co[3][ch] = be;
        if constexpr (KIND >= 2) { co[4][ch] = ca; co[5][ch] = cb; co[6][ch] = cd; }
    } else if constexpr (KIND == 1) {
        const float sc = is * ga;
        co[0][ch] = sc;
        co[1][ch] = be - mu * sc;
    } else {
        const float G = is * ga, C = ca * cd * is;
        co[0][ch] = G;                          // mask: u G + H > 0
        co[1][ch] = be - mu * G;
        co[2][ch] = ca;                         // A
        co[3][ch] = C;
        co[4][ch] = ca * (cd * is * mu - cb);   // E
    }
}
// a lane's NV consecutive channels from c0 on: LDS -> registers
template <int ESZ, int KIND>
__device__ __forceinline__ void pre_load_co(const float (*co)[PRE_MAX_C], int c0, PreCo<ESZ, KIND> &cv) {
    typedef PreForm<ESZ, KIND> F;
#pragma unroll
    for (int v = 0; v < F::NVEC; ++v)
#pragma unroll
        for (int q = 0; q < F::NV; q += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(&co[v][c0 + q]);
            cv.v[v][q] = t[0]; cv.v[v][q + 1] = t[1]; cv.v[v][q + 2] = t[2]; cv.v[v][q + 3] = t[3];
        }
}
// one 16-byte piece: x (forward: the BatchNorm's input; backward: dz), u (backward: the BatchNorm's input), a (kind 3: the skip gradient)
template <int ESZ, int KIND>
__device__ __forceinline__ u32x4 pre_piece(const u32x4 &x, const u32x4 &u, const u32x4 &a, const PreCo<ESZ, KIND> &cv, int relu, unsigned pm) {
    u32x4 o;
    if constexpr (ESZ == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t;
            if constexpr (KIND == 1) {
                t = bn_fwd_elem(__uint_as_float(x[q]), cv.v[0][q], cv.v[1][q], cv.v[2][q], cv.v[3][q]);
                if (relu) t = t > 0.f ? t : 0.f;
            } else {
                t = bn_bwd_elem(__uint_as_float(u[q]), __uint_as_float(x[q]), cv.v[0][q], cv.v[1][q], cv.v[2][q], cv.v[3][q], cv.v[4][q],
                                cv.v[5][q], cv.v[6][q], relu);
                if constexpr (KIND == 3) t += __uint_as_float(a[q]);
            }
            o[q] = __float_as_uint(t) & pm;
        }
    } else {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {     // two bf16 values per 32-bit word: packed fp32 math (v_pk_fma_f32)
            const f32x2 xv = {__uint_as_float(x[j] << 16), __uint_as_float(x[j] & 0xffff0000u)};
            f32x2 r;
            if constexpr (KIND == 1) {
                const f32x2 sc = {cv.v[0][2 * j], cv.v[0][2 * j + 1]}, sh = {cv.v[1][2 * j], cv.v[1][2 * j + 1]};
                r = __builtin_elementwise_fma(xv, sc, sh);
            } else {
                const f32x2 uv = {__uint_as_float(u[j] << 16), __uint_as_float(u[j] & 0xffff0000u)};
                const f32x2 G = {cv.v[0][2 * j], cv.v[0][2 * j + 1]}, H = {cv.v[1][2 * j], cv.v[1][2 * j + 1]};
                const f32x2 A = {cv.v[2][2 * j], cv.v[2][2 * j + 1]}, C = {cv.v[3][2 * j], cv.v[3][2 * j + 1]};
                const f32x2 E = {cv.v[4][2 * j], cv.v[4][2 * j + 1]};
                const f32x2 t = __builtin_elementwise_fma(-C, uv, E);
                const f32x2 full = __builtin_elementwise_fma(A, xv, t);
                r = full;
                if (relu) {
                    const f32x2 yv = __builtin_elementwise_fma(uv, G, H);
                    r[0] = yv[0] > 0.f ? full[0] : t[0];
                    r[1] = yv[1] > 0.f ? full[1] : t[1];
                }
                if constexpr (KIND == 3) {
                    const f32x2 av = {__uint_as_float(a[j] << 16), __uint_as_float(a[j] & 0xffff0000u)};
                    r = r + av;
                }
            }
            unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
            if constexpr (KIND == 1) {
                // ReLU AFTER the rounding, on the packed pair: max(int16, 0) zeroes every negative bf16 (and -0) — the same bits as
                // relu-then-round (rounding is monotonic and keeps 0)
                if (relu) w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), (s16x2){0, 0}));
            }
            o[j] = w & pm;
        }
    }
    return o;
}
}  // namespace
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

namespace doda_layers {
// doda_set_option(DODA_OPT_PRE_FWD_ROWS / DODA_OPT_PRE_BWD_ROWS): a BatchNorm op of doda_layers_run with at most this many rows is
// folded into the next convolution's gather (0: never); defaults from DODA_PRE_FWD_ROWS / DODA_PRE_BWD_ROWS (16384 / 0)
long long fwd_rows();
long long bwd_rows();
void set_fwd_rows(long long v);
void set_bwd_rows(long long v);
}  // namespace doda_layers

namespace doda_wlds {
bool enabled();
void set_enabled(bool on);
// bf16 48 -> 48 channels, K = 27, weights in LDS (spconv_wlds.hip); wp = wide-packed fragments [27][2][3][64] x 16 B
int launch_conv48(const void *x, unsigned x_bytes, const void *wp, const int32_t *tbl, unsigned tbl_bytes, int ld, int n_out,
                  void *y, unsigned y_bytes, const void *res, const EpiArgs &ep, int *n_part, hipStream_t s);
}  // namespace doda_wlds
