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
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
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
//
// In-kernel finish (round 3).  The partial rows used to be summed by a separate `final` launch of the BatchNorm
// (bn.hip: 75 launches of 6-7 us per U-Net step, each a handful of workgroups chasing L2 round trips).  With `totals`
// set, every workgroup publishes its rows with write-through stores, takes a ticket, and the LAST one to retire sums
// all rows — fixed order, fp64 — into totals[2][nc]; the BatchNorm's apply pass derives mean / invstd (or the backward
// coefficients) from the totals itself (bn_apply_tot / bn_bwd_apply_tot): the dependent launch is gone.
struct EpiArgs {   // plain data, shared across translation units
    float *stats;            // [n_part][2][nc] or null
    const void *bn_x;        // [n_out, nc] in the dtype of y, or null (forward statistics)
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   // [nc] each
    int bn_relu;
    int fin_rows;            // partial rows this launch writes (set by the launcher)
    double *totals;          // [2][nc] or null (the caller reduces the rows itself)
    unsigned *ticket;        // device counter of retired workgroups: zero between launches, re-armed by the last one
    // BatchNorm(+ReLU) PROLOGUE (ABI 6, doda_conv_epilogue.pre_*): applied to every gathered input row; null = none
    const float *pre_mean, *pre_invstd, *pre_gamma, *pre_beta;   // [kc] each
    int pre_relu;
    void *pre_out;           // [n_in, kc] normalised rows (bf16) or null
    int res_bcast;           // ABI 6: `res` is ONE row [nc] added to every output row (a bias): conv_fast only
};

constexpr unsigned FIN_GROUP = 32;         // workgroups per first-level ticket
constexpr unsigned FIN_STRIDE = 64;        // unsigned per ticket: one 256-byte line each
constexpr unsigned FIN_MAX_GROUPS = 511;   // -> launches of up to 16352 workgroups (larger ones leave the rows to the caller)

namespace doda_fin {
bool enabled();                               // doda_spconv_set_stats_finish (default on)
void set_enabled(bool on);
unsigned *ticket_for(hipStream_t s);          // one counter per (device, stream): launches of a stream do not overlap
// Launchers whose kernel finishes the statistics call arm(): fills fin_rows / ticket, or clears totals when the
// finish is switched off or no counter could be allocated.  Returns true when the launch will write `totals`.
bool arm(EpiArgs &ep, int rows, unsigned n_wg, hipStream_t s);
extern thread_local int last_finished;        // set by arm(): did the last launch of this thread write totals
}  // namespace doda_fin

namespace {

// ---- BatchNorm(+ReLU) prologue: the per-channel vectors of the eight channels a lane's 16-byte row piece holds, and
// bn.hip's bn_apply arithmetic on that piece — the same operations in the same order (the library is built with
// -ffp-contract=off), the same bf16 rounding: a conv with the prologue equals BatchNorm launch + conv bit for bit
// (a NaN may come out with another payload).
struct PreVec { f32x4 mu[2], is[2], ga[2], be[2]; };
__device__ __forceinline__ void pre_load(PreVec &p, const EpiArgs &ep, unsigned c0) {   // channels c0 .. c0 + 7
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        p.mu[h] = *reinterpret_cast<const f32x4 *>(ep.pre_mean + c0 + 4 * h);
        p.is[h] = *reinterpret_cast<const f32x4 *>(ep.pre_invstd + c0 + 4 * h);
        p.ga[h] = *reinterpret_cast<const f32x4 *>(ep.pre_gamma + c0 + 4 * h);
        p.be[h] = *reinterpret_cast<const f32x4 *>(ep.pre_beta + c0 + 4 * h);
    }
}
// two fp32 -> packed bf16 by the hardware's v_cvt_pk_bf16_f32 (round to nearest even: the bits f2bf() produces for every
// finite value and infinity; f2bf's NaN test compiles to EXEC-masked control flow — 4 scalar + ~8 vector instructions per
// element, which made the prologue cost 32 us per level-1 layer instead of 4)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ u32x4 pre_apply8(const u32x4 &r, const PreVec &p, int relu) {
    u32x4 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 v = {__uint_as_float(r[2 * h] << 16), __uint_as_float(r[2 * h] & 0xffff0000u),
                         __uint_as_float(r[2 * h + 1] << 16), __uint_as_float(r[2 * h + 1] & 0xffff0000u)};
        f32x4 t = (v - p.mu[h]) * p.is[h] * p.ga[h] + p.be[h];
        if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = t[q] > 0.f ? t[q] : 0.f;
        }
        o[2 * h] = cvt_pk_bf16(t[0], t[1]);
        o[2 * h + 1] = cvt_pk_bf16(t[2], t[3]);
    }
    return o;
}

// statistics rows travel between workgroups of ONE launch that may sit on different XCDs (one L2 each): write-through
// stores, L2-bypassing loads (sc0 sc1 = system-coherent on gfx950), ordered by hand around the ticket
__device__ __forceinline__ void stats_store4(float *dst, const f32x4 &v) {
    // (s_nop: a store of more than 64 bits reads its data registers a few cycles after issue and hipcc does not insert the
    // wait states for instructions it cannot see: the next VALU write to one of them would be stored instead)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void stats_load4(f32x4 &v, const float *src) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(src) : "memory");
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Called by EVERY thread of EVERY workgroup as the last thing of a kernel whose epilogue wrote statistics rows with
// stats_store4.  Wave w of the last workgroup takes the channel quads f = w, w + waves, ...; lane l the rows l, l + 64,
// ...: a fixed order whatever the arrival order of the workgroups.
// ONE_WAVE: only wave 0 of the workgroup is still running (the split-K form of conv_fast): no barriers.
template <bool ONE_WAVE = false>
__device__ __forceinline__ void stats_finish(const EpiArgs &ep, int nc) {
    if (!ep.totals) return;
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's rows are written through
    // Two levels of tickets: a device-scope atomic is executed at the memory side, ~10 ns each when thousands of
    // workgroups hit ONE address (conv_fast 64 -> 64: +27 us with a single counter).  Workgroups b, b+1, ... of a group
    // of FIN_GROUP count on the group's own line; the last of a group counts on the root.
    auto take = [&]() -> unsigned {
        const unsigned n_wg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned grp = bid / FIN_GROUP, n_grp = (n_wg + FIN_GROUP - 1) / FIN_GROUP;
        const unsigned in_grp = n_wg - grp * FIN_GROUP < FIN_GROUP ? n_wg - grp * FIN_GROUP : FIN_GROUP;
        unsigned *gt = ep.ticket + (1u + grp) * FIN_STRIDE;
        if (__hip_atomic_fetch_add(gt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_grp - 1u) return 0u;
        __hip_atomic_store(gt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // re-armed
        return __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grp - 1u ? 1u : 0u;
    };
    if constexpr (ONE_WAVE) {
        unsigned t = 0;
        if (threadIdx.x == 0) t = take();
        if (__builtin_amdgcn_readfirstlane((int)t) == 0) return;
    } else {
        __syncthreads();
        if (threadIdx.x == 0) s_last = take();
        __syncthreads();
        if (!s_last) return;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = ONE_WAVE ? 1 : (int)(blockDim.x >> 6);
    const int nf = nc >> 2, rows = ep.fin_rows;
    for (int f = wid; f < nf; f += nw) {
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        const float *base = ep.stats + f * 4;
        int r = lane;
        for (; r + 3 * 64 < rows; r += 4 * 64) {          // eight loads in flight per lane
            f32x4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *p = base + (long long)(r + k * 64) * 2 * nc;
                stats_load4(a[k], p);
                stats_load4(b[k], p + nc);
            }
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                         : : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) { s1[q] += (double)a[k][q]; s2[q] += (double)b[k][q]; }
        }
        for (; r < rows; r += 64) {
            f32x4 a, b;
            const float *p = base + (long long)r * 2 * nc;
            stats_load4(a, p);
            stats_load4(b, p + nc);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += (double)a[q]; s2[q] += (double)b[q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] = wave_sum_f64(s1[q]); s2[q] = wave_sum_f64(s2[q]); }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ep.totals[f * 4 + q] = s1[q]; ep.totals[nc + f * 4 + q] = s2[q]; }
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ep.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

namespace doda_tile {
bool enabled();   // doda_spconv_set_tile_kernel
// conv_tile over `tilebook` (doda_tilebook_build of tbl).  mode 0: bf16 16 channels, 1: bf16 32 channels, 2: fp32 16
// channels; out32: fp32 output rows.  *n_part (if given) receives the number of statistics rows.
int launch_conv_tile(int mode, bool out32, const void *x, unsigned x_bytes, const void *wp, unsigned wp_bytes, int nc, int NB,
                     const int32_t *tbl, int ld, int n_out, const void *tilebook, void *y, unsigned y_bytes, const void *res,
                     const EpiArgs &ep, int *n_part, hipStream_t s);
// does launch_conv_tile take ep.pre_* (BatchNorm prologue) for this mode / output type
inline bool takes_prologue(int mode, bool out32) { return (mode == 0 || mode == 1) && !out32; }
void pack_pair_layout2(const float *w, void *out, hipStream_t s);   // defined next to the pack kernels
}  // namespace doda_tile

namespace doda_dma {
bool enabled();
void set_enabled(bool on);
// bf16 16 -> 16, K = 27 over a tilebook: one persistent workgroup per CU fed by LDS-DMA (spconv_dma.hip); wp = pair-packed
int launch_conv16(const void *x, unsigned x_bytes, const void *wp, unsigned wp_bytes, const int32_t *tbl, int ld, int n_out,
                  const void *tilebook, void *y, unsigned y_bytes, const void *res, const EpiArgs &ep, int *n_part, hipStream_t s);
}  // namespace doda_dma

namespace doda_wlds {
bool enabled();
void set_enabled(bool on);
// bf16 48 -> 48 channels, K = 27, weights in LDS (spconv_wlds.hip); wp = wide-packed fragments [27][2][3][64] x 16 B
int launch_conv48(const void *x, unsigned x_bytes, const void *wp, const int32_t *tbl, unsigned tbl_bytes, int ld, int n_out,
                  void *y, unsigned y_bytes, const void *res, const EpiArgs &ep, int *n_part, hipStream_t s);
}  // namespace doda_wlds
