// Fused BatchNorm1d(+ReLU) on sparse-tensor features [M, C].
//
// DODA's U-Net is pre-activation: every sparse conv is preceded by BatchNorm1d(eps=1e-4,
// momentum=0.1) -> ReLU on the [M, C] feature matrix (reference model/unet.py:28,42-45,
// model/unet_block.py:23-30,46-49,67-79: 65 BN+ReLU pairs per forward).  Measured on MI355X,
// PyTorch's channels-last BN reductions take ~40 us per call at these shapes and BN is the largest
// single item of the training step, so the pair is provided as three HBM-bound passes:
//   stats   : per-channel mean / biased variance, shifted by the first row so that
//             E[(x-k)^2] - E[x-k]^2 does not cancel (fp32 partials per block, fp64 combine);
//             also applies the running-stat update (momentum, unbiased variance);
//   apply   : y = [relu]((x - mean) * invstd * gamma + beta);
//   backward: dz = dy * [y > 0] (mask recomputed from x, y is never re-read); per-channel sums of
//             dz and dz*xhat (block partials, fp64 combine), then
//             dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)), dgamma, dbeta.
// Lanes own 4-channel fragments (8 B bf16 / 16 B fp32); consecutive lanes walk consecutive
// fragments, so every access is a contiguous wave-wide burst.  All reductions are fixed-order:
// results are run-to-run deterministic.  Algorithmic bytes: fwd 3 passes, bwd 5 passes of M*C*s.
#include "common.hpp"
#include "bn_totals.hpp"
#include <stdlib.h>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    // gfx950: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) — the integer form cost ten instructions and an
    // EXEC round trip per value in the store epilogues
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

struct F32 {
    typedef float elem;
    static constexpr int W = 1;   // 4-channel fragments per 16-byte access
    static __device__ __forceinline__ f32x4 load4(const elem *p) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ void store4(elem *p, const f32x4 &v) { *reinterpret_cast<f32x4 *>(p) = v; }
    static __device__ __forceinline__ void loadw(const elem *p, f32x4 (&v)[1]) { v[0] = load4(p); }
    static __device__ __forceinline__ void storew(elem *p, const f32x4 (&v)[1]) { store4(p, v[0]); }
};
struct BF16 {
    typedef unsigned short elem;
    static __device__ __forceinline__ f32x4 load4(const elem *p) {
        const s16x4 r = *reinterpret_cast<const s16x4 *>(p);
        return (f32x4){bf2f((unsigned short)r[0]), bf2f((unsigned short)r[1]), bf2f((unsigned short)r[2]), bf2f((unsigned short)r[3])};
    }
    static __device__ __forceinline__ void store4(elem *p, const f32x4 &v) {
        s16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (short)f2bf(v[q]);
        *reinterpret_cast<s16x4 *>(p) = o;
    }
    static constexpr int W = 2;   // two 4-channel fragments (8 channels) per 16-byte access
    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void loadw(const elem *p, f32x4 (&v)[2]) {
        const s16x8_ r = *reinterpret_cast<const s16x8_ *>(p);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            v[h] = (f32x4){bf2f((unsigned short)r[4 * h]), bf2f((unsigned short)r[4 * h + 1]), bf2f((unsigned short)r[4 * h + 2]),
                           bf2f((unsigned short)r[4 * h + 3])};
    }
    static __device__ __forceinline__ void storew(elem *p, const f32x4 (&v)[2]) {
        s16x8_ o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (short)f2bf(v[q >> 2][q & 3]);
        *reinterpret_cast<s16x8_ *>(p) = o;
    }
};

constexpr int BN_BLOCK = 256;
constexpr int BN_MAX_BLOCKS = 1024;

// geometry shared by the reduction kernels: a block covers RPB rows per sweep, lane = (row lane, frag)
struct Geo {
    int nf;    // fragments per row = C/4
    int rpb;   // row lanes per block
};

// ---- pass 1: per-block partial sums of (x-k) and (x-k)^2 -------------------------------------
template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_stats_partial(const typename T::elem *__restrict__ x,
                                                             int m, int c, Geo g,
                                                             float *__restrict__ partial /*[blocks][2][C]*/) {
    extern __shared__ float lds[];  // [rpb][2][C]
    const int f = threadIdx.x % g.nf, rl = threadIdx.x / g.nf;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rl < g.rpb) {
        const f32x4 k = T::load4(x + f * 4);  // shift: first row
        for (long long r = (long long)blockIdx.x * g.rpb + rl; r < m; r += (long long)gridDim.x * g.rpb) {
            const f32x4 v = T::load4(x + r * c + f * 4) - k;
            s1 += v;
            s2 += v * v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds[(rl * 2 + 0) * c + f * 4 + q] = s1[q];
            lds[(rl * 2 + 1) * c + f * 4 + q] = s2[q];
        }
    }
    doda_sync();
    for (int e = threadIdx.x; e < 2 * c; e += BN_BLOCK) {
        float t = 0.f;
        for (int r = 0; r < g.rpb; ++r) t += lds[r * 2 * c + e];
        partial[(long long)blockIdx.x * 2 * c + e] = t;
    }
}

__device__ __forceinline__ double wave_sum(double v) {  // fixed-order butterfly: deterministic
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- combine (one wave per channel): mean, invstd, running stats -------------------------------
template <class T>
__global__ __launch_bounds__(64) void bn_stats_final(const typename T::elem *__restrict__ x,
                                                     const float *__restrict__ partial, int nblocks,
                                                     int m, int c, float eps, float momentum,
                                                     float *__restrict__ mean, float *__restrict__ invstd,
                                                     float *__restrict__ running_mean,
                                                     float *__restrict__ running_var,
                                                     long long *__restrict__ num_batches_tracked) {
    const int ch = blockIdx.x;
    // everything lane 0 needs at the end is requested before the partial sums are read: one memory
    // round trip (2-3 us for a kernel this small) instead of two
    const f32x4 k4 = T::load4(x + (ch & ~3));
    float rm_old = 0.f, rv_old = 0.f;
    long long n_tracked = 0;
    if (threadIdx.x == 0 && running_mean) { rm_old = running_mean[ch]; rv_old = running_var[ch]; }
    if (threadIdx.x == 0 && ch == 0 && num_batches_tracked) n_tracked = *num_batches_tracked;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        s1 += (double)partial[(long long)b * 2 * c + ch];
        s2 += (double)partial[(long long)b * 2 * c + c + ch];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x != 0) return;
    if (ch == 0 && num_batches_tracked) *num_batches_tracked = n_tracked + 1;
    const double k = (double)k4[ch & 3];
    const double d = s1 / m;
    double var = s2 / m - d * d;
    if (var < 0.0) var = 0.0;
    const double mu = k + d;
    mean[ch] = (float)mu;
    invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = m > 1 ? var * (double)m / (double)(m - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * (double)rm_old + momentum * mu);
        running_var[ch] = (float)((1.0 - momentum) * (double)rv_old + momentum * unbiased);
    }
}

// ---- statistics from TOTALS (ABI 9) ---------------------------------------------------------------------------------
// The conv epilogues can add their workgroups' (sum, sum of squares) straight into DODA_STATS_SLOTS rows of fp64 totals
// (spconv_common.hpp stats_emit: hardware fp64 atomics, no return) instead of writing one row per workgroup.  The apply
// sweeps below then need no reduction launch in front of them: the first c threads of every workgroup add the eight slots of
// their channel (fixed order, fp64), do the arithmetic of bn_fwd_final_stats / bn_bwd_final_stats and park the per-channel
// vectors in LDS; workgroup 0 publishes what later kernels need.  (Round 3's form of this — the conv kernel's LAST workgroup
// summing the rows behind a ticket — made the conv kernels 15-160 % slower: profiles/r03_stats_finish_ab.txt.)
// ta / tb: the totals of the producers of the columns [0, ca) and [ca, c) (tb null: one producer) — a channel concatenation.
// (TotArgs, tot_sums and the per-channel arithmetic: bn_totals.hpp — shared with the conv kernels that fold a BatchNorm into their gather)
__device__ __forceinline__ void tot_fwd_prologue(const TotArgs &t, int c, float *v_mu, float *v_is) {
    for (int ch = threadIdx.x; ch < c; ch += BN_BLOCK) {
        float mu, is;
        tot_fwd_channel(t, c, ch, blockIdx.x == 0, mu, is);
        v_mu[ch] = mu;
        v_is[ch] = is;
    }
    doda_sync();
}
__device__ __forceinline__ void tot_bwd_prologue(const TotArgs &t, int c, const float *__restrict__ invstd,
                                                 const float *__restrict__ gamma, float *v_co /*[3][c]*/) {
    for (int ch = threadIdx.x; ch < c; ch += BN_BLOCK) {
        float ca, cb, cd;
        tot_bwd_channel(t, c, ch, blockIdx.x == 0, invstd[ch], gamma[ch], ca, cb, cd);   // dx = a * (dz - b - xhat * d)
        v_co[ch] = ca;
        v_co[c + ch] = cb;
        v_co[2 * c + ch] = cd;
    }
    doda_sync();
}

// ---- pass 2: normalise + affine (+ReLU) --------------------------------------------------------
// Round 3: the launcher picks a grid whose thread count is a multiple of the fragments per row, so a thread's channels
// never change while it strides over the rows: the per-channel vectors are loaded ONCE into registers (they used to be
// four extra vector loads per 8 bytes of payload — the sweep was bound by the texture path's instruction rate, 3.3 TB/s
// at level 1, not by HBM), and bf16 rows move in 16-byte accesses (two fragments per thread).  Same arithmetic, same
// order: results unchanged bit for bit.  FIXED = false: the general form (any grid).
template <class T, bool FIXED, bool TOT = false>
__global__ __launch_bounds__(BN_BLOCK) void bn_apply(const typename T::elem *__restrict__ x,
                                                     long long n_frag, int nf,
                                                     const float *mean,
                                                     const float *invstd,
                                                     const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, int relu,
                                                     typename T::elem *__restrict__ y, const TotArgs tot = TotArgs()) {
    // (round 6) FIXED: this thread's first rows are REQUESTED before the totals prologue (its fp64 arithmetic and barrier are a
    // dependent round trip of their own): with the whole grid resident at once (launch_apply_tot caps it) every workgroup used to sit
    // through the prologue with no row in flight — 12-16 us per level-1 sweep of 38 MB in the step, 7.5 us without the prologue
    f32x4 pre_v[T::W];
    [[maybe_unused]] bool pre_have = false;
    if constexpr (FIXED && TOT) {
        const long long n_w0 = n_frag / T::W, e00 = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
        pre_have = e00 < n_w0;
        T::loadw(x + (pre_have ? e00 : 0) * (4 * T::W), pre_v);
    }
    if constexpr (TOT) {   // mean / invstd from the totals, through LDS
        __shared__ __attribute__((aligned(16))) float v_mu[BN_TOT_MAX_C], v_is[BN_TOT_MAX_C];
        tot_fwd_prologue(tot, nf * 4, v_mu, v_is);
        mean = v_mu;
        invstd = v_is;
    }
    if constexpr (FIXED) {
        constexpr int W = T::W;
        const long long n_w = n_frag / W, e0 = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
        const int f = (int)(e0 % (nf / W)) * W;
        f32x4 mu[W], is[W], ga[W], be[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            mu[w] = *reinterpret_cast<const f32x4 *>(mean + (f + w) * 4);
            is[w] = *reinterpret_cast<const f32x4 *>(invstd + (f + w) * 4);
            ga[w] = *reinterpret_cast<const f32x4 *>(gamma + (f + w) * 4);
            be[w] = *reinterpret_cast<const f32x4 *>(beta + (f + w) * 4);
        }
        for (long long e = e0; e < n_w; e += (long long)gridDim.x * BN_BLOCK) {
            f32x4 v[W];
            if (TOT && e == e0) {
#pragma unroll
                for (int w = 0; w < W; ++w) v[w] = pre_v[w];
            } else T::loadw(x + e * (4 * W), v);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                f32x4 o = (v[w] - mu[w]) * is[w] * ga[w] + be[w];
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
                }
                v[w] = o;
            }
            T::storew(y + e * (4 * W), v);
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; e < n_frag;
         e += (long long)gridDim.x * BN_BLOCK) {
        const int f = (int)(e % nf);
        const f32x4 v = T::load4(x + e * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + f * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
        const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
        f32x4 o = (v - mu) * is * ga + be;
        if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
        }
        T::store4(y + e * 4, o);
    }
}

// grid of an apply sweep: as many workgroups as fragments need (<= 4096), rounded DOWN so that the thread count is a
// multiple of the 16-byte columns per row (then FIXED applies); *fixed = false when no such grid exists
inline int apply_grid(long long n_frag, int nf, int W, bool *fixed, int cap = 4096) {
    const long long n_w = n_frag / W;
    const int cols = nf / W;
    long long grid = (n_w + BN_BLOCK - 1) / BN_BLOCK;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    *fixed = false;
    if (nf % W != 0 || n_frag % W != 0) return (int)((n_frag + BN_BLOCK - 1) / BN_BLOCK < 4096 ? (n_frag + BN_BLOCK - 1) / BN_BLOCK : 4096);
    int a = cols, b = BN_BLOCK;            // q = cols / gcd(cols, BN_BLOCK): grid must be a multiple of q
    while (b) { const int t = a % b; a = b; b = t; }
    const int q = cols / a;
    if (grid >= q) {
        *fixed = true;
        return (int)(grid - grid % q);
    }
    return (int)((n_frag + BN_BLOCK - 1) / BN_BLOCK < 4096 ? (n_frag + BN_BLOCK - 1) / BN_BLOCK : 4096);
}

// ---- backward pass 1: per-block partial sums of dz and dz*xhat --------------------------------
template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_partial(const typename T::elem *__restrict__ x,
                                                           const typename T::elem *__restrict__ dy,
                                                           int m, int c, Geo g,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int relu,
                                                           float *__restrict__ partial) {
    extern __shared__ float lds[];
    const int f = threadIdx.x % g.nf, rl = threadIdx.x / g.nf;
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rl < g.rpb) {
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + f * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
        const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
        for (long long r = (long long)blockIdx.x * g.rpb + rl; r < m; r += (long long)gridDim.x * g.rpb) {
            const f32x4 xh = (T::load4(x + r * c + f * 4) - mu) * is;
            f32x4 dz = T::load4(dy + r * c + f * 4);
            if (relu) {
                const f32x4 yv = xh * ga + be;
#pragma unroll
                for (int q = 0; q < 4; ++q) dz[q] = yv[q] > 0.f ? dz[q] : 0.f;
            }
            s1 += dz;
            s2 += dz * xh;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lds[(rl * 2 + 0) * c + f * 4 + q] = s1[q];
            lds[(rl * 2 + 1) * c + f * 4 + q] = s2[q];
        }
    }
    doda_sync();
    for (int e = threadIdx.x; e < 2 * c; e += BN_BLOCK) {
        float t = 0.f;
        for (int r = 0; r < g.rpb; ++r) t += lds[r * 2 * c + e];
        partial[(long long)blockIdx.x * 2 * c + e] = t;
    }
}

__global__ __launch_bounds__(64) void bn_bwd_final(const float *__restrict__ partial, int nblocks,
                                                   int m, int c, const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma,
                                                   float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                   float *__restrict__ coef /*[3][C]*/) {
    const int ch = blockIdx.x;
    const float a_coef = gamma[ch] * invstd[ch];   // requested before the partial sums (see bn_stats_final)
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        s1 += (double)partial[(long long)b * 2 * c + ch];
        s2 += (double)partial[(long long)b * 2 * c + c + ch];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x != 0) return;
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
    // dx = a * (dz - b - xhat * d)
    coef[ch] = a_coef;
    coef[c + ch] = (float)(s1 / m);
    coef[2 * c + ch] = (float)(s2 / m);
}

template <class T, bool FIXED, bool TOT = false>
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_apply(const typename T::elem *__restrict__ x,
                                                         const typename T::elem *__restrict__ dy,
                                                         long long n_frag, int nf, int c,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int relu,
                                                         const float *coef,
                                                         typename T::elem *__restrict__ dx,
                                                         const typename T::elem *__restrict__ add, int add_ld = 0,
                                                         const TotArgs tot = TotArgs()) {
    f32x4 pre_x[T::W], pre_dz[T::W], pre_av[T::W];   // (round 6: first rows requested before the totals prologue, see bn_apply)
    if constexpr (FIXED && TOT) {
        const int cols0 = nf / T::W;
        const long long n_w0 = n_frag / T::W, e00 = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
        const long long ee = e00 < n_w0 ? e00 : 0;
        T::loadw(x + ee * (4 * T::W), pre_x);
        T::loadw(dy + ee * (4 * T::W), pre_dz);
        if (add) T::loadw(add_ld ? add + (ee / cols0) * add_ld + (int)(ee % cols0) * T::W * 4 : add + ee * (4 * T::W), pre_av);
    }
    if constexpr (TOT) {   // the three coefficient vectors from the totals, through LDS
        __shared__ __attribute__((aligned(16))) float v_co[3 * BN_TOT_MAX_C];
        tot_bwd_prologue(tot, c, invstd, gamma, v_co);
        coef = v_co;
    }
    if constexpr (FIXED) {   // (see bn_apply: per-thread channel vectors in registers, 16-byte accesses)
        constexpr int W = T::W;
        const int cols = nf / W;
        const long long n_w = n_frag / W, e0 = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
        const int f = (int)(e0 % cols) * W;
        f32x4 mu[W], is[W], ga[W], be[W], ca[W], cb[W], cd[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            mu[w] = *reinterpret_cast<const f32x4 *>(mean + (f + w) * 4);
            is[w] = *reinterpret_cast<const f32x4 *>(invstd + (f + w) * 4);
            ga[w] = *reinterpret_cast<const f32x4 *>(gamma + (f + w) * 4);
            be[w] = *reinterpret_cast<const f32x4 *>(beta + (f + w) * 4);
            ca[w] = *reinterpret_cast<const f32x4 *>(coef + (f + w) * 4);
            cb[w] = *reinterpret_cast<const f32x4 *>(coef + c + (f + w) * 4);
            cd[w] = *reinterpret_cast<const f32x4 *>(coef + 2 * c + (f + w) * 4);
        }
        for (long long e = e0; e < n_w; e += (long long)gridDim.x * BN_BLOCK) {
            f32x4 xv[W], dz[W], av[W];
            if (TOT && e == e0) {
#pragma unroll
                for (int w = 0; w < W; ++w) { xv[w] = pre_x[w]; dz[w] = pre_dz[w]; av[w] = pre_av[w]; }
            } else {
                T::loadw(x + e * (4 * W), xv);
                T::loadw(dy + e * (4 * W), dz);
                if (add) T::loadw(add_ld ? add + (e / cols) * add_ld + f * 4 : add + e * (4 * W), av);
            }
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const f32x4 xh = (xv[w] - mu[w]) * is[w];
                if (relu) {
                    const f32x4 yv = xh * ga[w] + be[w];
#pragma unroll
                    for (int q = 0; q < 4; ++q) dz[w][q] = yv[q] > 0.f ? dz[w][q] : 0.f;
                }
                f32x4 o = ca[w] * (dz[w] - cb[w] - xh * cd[w]);
                if (add) o += av[w];
                xv[w] = o;
            }
            T::storew(dx + e * (4 * W), xv);
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; e < n_frag;
         e += (long long)gridDim.x * BN_BLOCK) {
        const int f = (int)(e % nf);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + f * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
        const f32x4 xh = (T::load4(x + e * 4) - mu) * is;
        f32x4 dz = T::load4(dy + e * 4);
        if (relu) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
            const f32x4 yv = xh * ga + be;
#pragma unroll
            for (int q = 0; q < 4; ++q) dz[q] = yv[q] > 0.f ? dz[q] : 0.f;
        }
        const f32x4 a = *reinterpret_cast<const f32x4 *>(coef + f * 4);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(coef + c + f * 4);
        const f32x4 d = *reinterpret_cast<const f32x4 *>(coef + 2 * c + f * 4);
        f32x4 o = a * (dz - b - xh * d);
        // a second gradient of x (residual / skip path) summed here; add_ld != 0: its rows lie add_ld elements apart
        // (a column slice of a wider matrix: the gradient of torch.cat's input, reference model/unet_block.py:93)
        if (add) o += T::load4(add_ld ? add + (e / nf) * add_ld + f * 4 : add + e * 4);
        T::store4(dx + e * 4, o);
    }
}

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
inline int plain_grid(long long n_frag) {
    return (int)((n_frag + BN_BLOCK - 1) / BN_BLOCK < 4096 ? (n_frag + BN_BLOCK - 1) / BN_BLOCK : 4096);
}
template <class T>
void launch_apply(const typename T::elem *x, long long n_frag, int nf, const float *mean, const float *invstd,
                  const float *gamma, const float *beta, int relu, typename T::elem *y, hipStream_t s) {
    bool fixed;
    int grid = apply_grid(n_frag, nf, T::W, &fixed);
    if (fixed && !(al16(x) && al16(y))) { fixed = false; grid = plain_grid(n_frag); }
    if (fixed)
        hipLaunchKernelGGL((bn_apply<T, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, n_frag, nf, mean, invstd, gamma, beta, relu, y);
    else
        hipLaunchKernelGGL((bn_apply<T, false>), dim3(grid), dim3(BN_BLOCK), 0, s, x, n_frag, nf, mean, invstd, gamma, beta, relu, y);
}
template <class T>
void launch_apply_tot(const typename T::elem *x, long long n_frag, int nf, const float *gamma, const float *beta, int relu,
                      typename T::elem *y, const TotArgs &tot, hipStream_t s) {
    // (round 6) at most 1024 workgroups (4 per CU, all resident at once): every workgroup pays the totals prologue — 16 doubles per
    // channel out of padded 128-byte lines, fp64 division / square root, a barrier — once per ~5 x 16 bytes per thread at level 1
    // instead of once per 16 bytes.  In-step sums of bn_apply + bn_bwd_apply (tools/layers_prof.sh, DODA_BN_TOT_GRID): 4096
    // workgroups 613 us, 2048 573, 1024 518, 768 518, 512 531, 256 667
    static const int cap = [] { const char *e = getenv("DODA_BN_TOT_GRID"); const int v = e ? atoi(e) : 1024; return v < 64 ? 64 : v > 4096 ? 4096 : v; }();
    bool fixed;
    int grid = apply_grid(n_frag, nf, T::W, &fixed, cap);
    if (fixed && !(al16(x) && al16(y))) { fixed = false; grid = plain_grid(n_frag); }
    if (fixed)
        hipLaunchKernelGGL((bn_apply<T, true, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, n_frag, nf, nullptr, nullptr, gamma, beta, relu, y, tot);
    else
        hipLaunchKernelGGL((bn_apply<T, false, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, n_frag, nf, nullptr, nullptr, gamma, beta, relu, y, tot);
}
template <class T>
void launch_bwd_apply_tot(const typename T::elem *x, const typename T::elem *dy, long long n_frag, int nf, int c,
                          const float *mean, const float *invstd, const float *gamma, const float *beta, int relu,
                          typename T::elem *dx, const typename T::elem *add, int add_ld, const TotArgs &tot, hipStream_t s) {
    static const int cap = [] { const char *e = getenv("DODA_BN_TOT_GRID"); const int v = e ? atoi(e) : 1024; return v < 64 ? 64 : v > 4096 ? 4096 : v; }();
    bool fixed;
    int grid = apply_grid(n_frag, nf, T::W, &fixed, cap);
    if (fixed && !(al16(x) && al16(dy) && al16(dx) && (!add || (al16(add) && add_ld % (4 * T::W) == 0)))) {
        fixed = false;
        grid = plain_grid(n_frag);
    }
    if (fixed)
        hipLaunchKernelGGL((bn_bwd_apply<T, true, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, dy, n_frag, nf, c, mean, invstd, gamma,
                           beta, relu, nullptr, dx, add, add_ld, tot);
    else
        hipLaunchKernelGGL((bn_bwd_apply<T, false, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, dy, n_frag, nf, c, mean, invstd, gamma,
                           beta, relu, nullptr, dx, add, add_ld, tot);
}
template <class T>
void launch_bwd_apply(const typename T::elem *x, const typename T::elem *dy, long long n_frag, int nf, int c,
                      const float *mean, const float *invstd, const float *gamma, const float *beta, int relu,
                      const float *coef, typename T::elem *dx, const typename T::elem *add, int add_ld, hipStream_t s) {
    bool fixed;
    int grid = apply_grid(n_frag, nf, T::W, &fixed);
    if (fixed && !(al16(x) && al16(dy) && al16(dx) && (!add || (al16(add) && add_ld % (4 * T::W) == 0)))) {
        fixed = false;
        grid = plain_grid(n_frag);
    }
    if (fixed)
        hipLaunchKernelGGL((bn_bwd_apply<T, true>), dim3(grid), dim3(BN_BLOCK), 0, s, x, dy, n_frag, nf, c, mean, invstd, gamma,
                           beta, relu, coef, dx, add, add_ld);
    else
        hipLaunchKernelGGL((bn_bwd_apply<T, false>), dim3(grid), dim3(BN_BLOCK), 0, s, x, dy, n_frag, nf, c, mean, invstd, gamma,
                           beta, relu, coef, dx, add, add_ld);
}

// ---- small-M variants: ONE launch per direction --------------------------------------------------
// Deep U-Net levels have a few hundred to ~10k rows; three launches per BN pass are then pure
// launch-floor time.  Here a block owns one 4-channel fragment column over ALL rows, so the
// statistics need no cross-block step: sweep 1 reduces, sweep 2 applies (rows stay in L2).
constexpr int BN_SMALL_ROWS = 4096;   // measured: at ~11k rows the C/4 blocks of this form take up to 80 us

__device__ __forceinline__ f32x4 block_sum4(f32x4 v, float (*lds)[4]) {  // 256 threads, fixed order
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], d, 64);
    }
    const int wid = threadIdx.x >> 6;
    doda_sync();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lds[wid][q] = v[q];
    }
    doda_sync();
    f32x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (lds[0][q] + lds[1][q]) + (lds[2][q] + lds[3][q]);
    return t;
}

// both sums through ONE LDS exchange (the small kernels are launch-floor kernels: every barrier pair counts)
__device__ __forceinline__ void block_sum4x2(f32x4 &a, f32x4 &b, float (*lds)[8]) {  // 256 threads, fixed order
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] += __shfl_xor(a[q], d, 64); b[q] += __shfl_xor(b[q], d, 64); }
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { lds[wid][q] = a[q]; lds[wid][4 + q] = b[q]; }
    }
    doda_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = (lds[0][q] + lds[1][q]) + (lds[2][q] + lds[3][q]);
        b[q] = (lds[0][4 + q] + lds[1][4 + q]) + (lds[2][4 + q] + lds[3][4 + q]);
    }
}

template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_small_fwd(const typename T::elem *__restrict__ x, int m,
                                                         int c, float eps, float momentum,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         float *__restrict__ running_mean,
                                                         float *__restrict__ running_var,
                                                         long long *__restrict__ nbt, int relu,
                                                         typename T::elem *__restrict__ y,
                                                         float *__restrict__ mean,
                                                         float *__restrict__ invstd) {
    __shared__ float lds[4][8];
    const int f = blockIdx.x;
    // Every global read of the kernel is issued up front and independently (rows past m re-read row
    // m-1 and are masked): a dependent round trip to L2/HBM costs 2-3 us here and the old form had
    // four of them in series (shift row, stats sweep, affine parameters, apply sweep).
    constexpr int RPT = BN_SMALL_ROWS / BN_BLOCK;
    const f32x4 k = T::load4(x + f * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
    // (the running statistics and the batch counter too: updating them element by element at the end
    // cost thread 0 five more dependent round trips)
    f32x4 rm = {0, 0, 0, 0}, rv = {0, 0, 0, 0};
    long long n_tracked = 0;
    if (threadIdx.x == 0 && running_mean) {
        rm = *reinterpret_cast<const f32x4 *>(running_mean + f * 4);
        rv = *reinterpret_cast<const f32x4 *>(running_var + f * 4);
    }
    if (threadIdx.x == 0 && f == 0 && nbt) n_tracked = *nbt;
    f32x4 xv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = threadIdx.x + i * BN_BLOCK;
        xv[i] = T::load4(x + (long long)(r < m ? r : m - 1) * c + f * 4);
    }
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const f32x4 v = xv[i] - k;
        const float w = threadIdx.x + i * BN_BLOCK < m ? 1.f : 0.f;
        s1 += v * w;
        s2 += v * v * w;
    }
    block_sum4x2(s1, s2, lds);
    f32x4 mu, is;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double d = (double)s1[q] / m;
        double var = (double)s2[q] / m - d * d;
        if (var < 0.0) var = 0.0;
        mu[q] = (float)((double)k[q] + d);
        is[q] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = m > 1 ? var * (double)m / (double)(m - 1) : var;
        rm[q] = (float)((1.0 - momentum) * (double)rm[q] + momentum * ((double)k[q] + d));
        rv[q] = (float)((1.0 - momentum) * (double)rv[q] + momentum * unbiased);
    }
    if (threadIdx.x == 0) {
        *reinterpret_cast<f32x4 *>(mean + f * 4) = mu;
        *reinterpret_cast<f32x4 *>(invstd + f * 4) = is;
        if (running_mean) {
            *reinterpret_cast<f32x4 *>(running_mean + f * 4) = rm;
            *reinterpret_cast<f32x4 *>(running_var + f * 4) = rv;
        }
        if (f == 0 && nbt) *nbt = n_tracked + 1;
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = threadIdx.x + i * BN_BLOCK;
        if (r < m) {
            f32x4 o = (xv[i] - mu) * is * ga + be;
            if (relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
            }
            T::store4(y + (long long)r * c + f * 4, o);
        }
    }
}

template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_small_bwd(const typename T::elem *__restrict__ x,
                                                         const typename T::elem *__restrict__ dy,
                                                         int m, int c, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int relu,
                                                         typename T::elem *__restrict__ dx,
                                                         float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta,
                                                         const typename T::elem *__restrict__ add, int add_ld = 0) {
    __shared__ float lds[4][8];
    const int f = blockIdx.x;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + f * 4);
    const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
    constexpr int RPT = BN_SMALL_ROWS / BN_BLOCK;
    f32x4 xh[RPT], dz[RPT], ad[RPT];   // all reads up front, kept for the apply pass (see bn_small_fwd)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = threadIdx.x + i * BN_BLOCK;
        const long long off = (long long)(r < m ? r : m - 1) * c + f * 4;
        xh[i] = T::load4(x + off);
        dz[i] = T::load4(dy + off);
        ad[i] = add ? T::load4(add_ld ? add + (long long)(r < m ? r : m - 1) * add_ld + f * 4 : add + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        xh[i] = (xh[i] - mu) * is;
        if (relu) {
            const f32x4 yv = xh[i] * ga + be;
#pragma unroll
            for (int q = 0; q < 4; ++q) dz[i][q] = yv[q] > 0.f ? dz[i][q] : 0.f;
        }
        const float w = threadIdx.x + i * BN_BLOCK < m ? 1.f : 0.f;
        s1 += dz[i] * w;
        s2 += dz[i] * xh[i] * w;
    }
    block_sum4x2(s1, s2, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { dbeta[f * 4 + q] = s1[q]; dgamma[f * 4 + q] = s2[q]; }
    }
    const float inv_m = 1.f / (float)m;
    const f32x4 a = ga * is, b = s1 * inv_m, d = s2 * inv_m;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = threadIdx.x + i * BN_BLOCK;
        if (r < m) T::store4(dx + (long long)r * c + f * 4, a * (dz[i] - b - xh[i] * d) + ad[i]);
    }
}

// ---- statistics delivered by a conv epilogue (doda_spconv_gather_ex) ---------------------------------
// `stats` = [rows][2][c] partial sums, one row per workgroup tile of the conv.  One block per 4-channel
// fragment: 256 threads walk the rows with 16-byte loads, fp64 sums, fixed-order tree in LDS.
__device__ __forceinline__ void block_sum_d4(double (&v)[4], double (*lds)[4]) {   // 256 threads, fixed order
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
    const int wid = threadIdx.x >> 6;
    doda_sync();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lds[wid][q] = v[q];
    }
    doda_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (lds[0][q] + lds[1][q]) + (lds[2][q] + lds[3][q]);
}

// both sums at once: one pair of barriers instead of two (the `final` kernels are launch-floor kernels: 75 per step)
__device__ __forceinline__ void block_sum_d8(double (&a)[4], double (&b)[4], double (*lds)[8]) {   // 256 threads, fixed order
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = wave_sum(a[q]); b[q] = wave_sum(b[q]); }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { lds[wid][q] = a[q]; lds[wid][4 + q] = b[q]; }
    }
    doda_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = (lds[0][q] + lds[1][q]) + (lds[2][q] + lds[3][q]);
        b[q] = (lds[0][4 + q] + lds[1][4 + q]) + (lds[2][4 + q] + lds[3][4 + q]);
    }
}

// this thread's share of the partial rows of channel quad f.  Four rows per trip with all eight loads
// issued before the first add: a level-1 layer has ~4700 partial rows and four blocks to reduce them, so
// the kernel is a chain of L2 round trips (13.4 us with one row per trip)
__device__ __forceinline__ void sum_partials(const float *__restrict__ stats, int rows, int c, int f,
                                             double (&s1)[4], double (&s2)[4]) {
    int r = threadIdx.x;
    for (; r + 3 * BN_BLOCK < rows; r += 4 * BN_BLOCK) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *p = stats + (long long)(r + k * BN_BLOCK) * 2 * c + f * 4;
            a[k] = *reinterpret_cast<const f32x4 *>(p);
            b[k] = *reinterpret_cast<const f32x4 *>(p + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += (double)a[k][q]; s2[q] += (double)b[k][q]; }
    }
    for (; r < rows; r += BN_BLOCK) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(stats + (long long)r * 2 * c + f * 4);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(stats + (long long)r * 2 * c + c + f * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += (double)a[q]; s2[q] += (double)b[q]; }
    }
}

// (bodies as device functions: the chained kernels below run them in their first workgroups)
__device__ __forceinline__ void fwd_final_body(int f, const float *__restrict__ stats, int rows, int m, int c, float eps,
                                               float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                                               float *__restrict__ running_mean, float *__restrict__ running_var,
                                               long long *__restrict__ nbt) {
    __shared__ double lds[4][8];
    // threads 0..3 finish one channel each (the fp64 divide / square root chains of the four channels side by side);
    // what they need at the end is requested before the partial sums: one round trip, not two
    const int q = threadIdx.x & 3;
    float rm = 0.f, rv = 0.f;
    long long n_tracked = 0;
    if (threadIdx.x < 4 && running_mean) {
        rm = running_mean[f * 4 + q];
        rv = running_var[f * 4 + q];
    }
    if (threadIdx.x == 0 && f == 0 && nbt) n_tracked = *nbt;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    sum_partials(stats, rows, c, f, s1, s2);
    block_sum_d8(s1, s2, lds);
    if (threadIdx.x >= 4) return;
    double a1 = s1[0], a2 = s2[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) { if (q == k) { a1 = s1[k]; a2 = s2[k]; } }
    const double d = a1 / m;
    double var = a2 / m - d * d;
    if (var < 0.0) var = 0.0;
    mean[f * 4 + q] = (float)d;
    invstd[f * 4 + q] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = m > 1 ? var * (double)m / (double)(m - 1) : var;
        running_mean[f * 4 + q] = (float)((1.0 - momentum) * (double)rm + momentum * d);
        running_var[f * 4 + q] = (float)((1.0 - momentum) * (double)rv + momentum * unbiased);
    }
    if (threadIdx.x == 0 && f == 0 && nbt) *nbt = n_tracked + 1;
}

__global__ __launch_bounds__(BN_BLOCK) void bn_fwd_final_stats(const float *__restrict__ stats, int rows, int m, int c,
                                                               float eps, float momentum, float *__restrict__ mean,
                                                               float *__restrict__ invstd,
                                                               float *__restrict__ running_mean,
                                                               float *__restrict__ running_var,
                                                               long long *__restrict__ nbt) {
    fwd_final_body(blockIdx.x, stats, rows, m, c, eps, momentum, mean, invstd, running_mean, running_var, nbt);
}

__device__ __forceinline__ void bwd_final_body(int f, const float *__restrict__ stats, int rows, int m, int c,
                                               const float *__restrict__ invstd, const float *__restrict__ gamma,
                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                               float *__restrict__ coef /*[3][C]*/) {
    __shared__ double lds[4][8];
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
    const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    sum_partials(stats, rows, c, f, s1, s2);
    block_sum_d8(s1, s2, lds);
    if (threadIdx.x != 0) return;
    f32x4 db, dg, a, bb, dd;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        db[q] = (float)s1[q];
        dg[q] = (float)s2[q];
        a[q] = ga[q] * is[q];
        bb[q] = (float)(s1[q] / m);
        dd[q] = (float)(s2[q] / m);
    }
    *reinterpret_cast<f32x4 *>(dbeta + f * 4) = db;
    *reinterpret_cast<f32x4 *>(dgamma + f * 4) = dg;
    *reinterpret_cast<f32x4 *>(coef + f * 4) = a;           // dx = a * (dz - b - xhat * d)
    *reinterpret_cast<f32x4 *>(coef + c + f * 4) = bb;
    *reinterpret_cast<f32x4 *>(coef + 2 * c + f * 4) = dd;
}

__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_final_stats(const float *__restrict__ stats, int rows, int m, int c,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               float *__restrict__ coef /*[3][C]*/) {
    bwd_final_body(blockIdx.x, stats, rows, m, c, invstd, gamma, dgamma, dbeta, coef);
}

// ---- final + apply in ONE launch (round 3) ----------------------------------------------------------
// The `final` kernels above are four to twelve blocks chasing L2 round trips: 6-7 us each, 75 of them per U-Net
// step.  When the conv epilogue delivers few partial rows (the persistent tile kernels write ONE per workgroup:
// <= 768 rows, ~100 KB) every workgroup of the apply pass can afford to reduce them itself — same rows, same
// order, same fp64 combine in every workgroup, so all of them arrive at identical statistics — and the dependent
// launch disappears.  A fixed grid of BN_FUSED_BLOCKS workgroups keeps the redundant reads (grid x partial bytes,
// L2 hits) well below the tensor itself; mean / invstd / gamma / beta (or the backward coefficients) then sit in
// LDS for the apply sweep instead of being re-read per element.  Workgroup 0 publishes the statistics.
constexpr int BN_FUSED_BLOCKS = 256;
constexpr int BN_FUSED_MAX_PARTIAL_BYTES = 160 * 1024;
constexpr int BN_FUSED_MAX_C = 64;          // channels whose vectors fit the LDS arrays below

// every thread's share of the partial rows of its fragment: thread = (row lane rl, fragment f); returns the totals of
// fragment f in all threads with rl == 0 ... through LDS, fixed order
__device__ __forceinline__ void fused_reduce(const float *__restrict__ stats, int rows, int c, int nf, int rpb,
                                             double (*red)[8], double (&s1)[4], double (&s2)[4]) {
    const int f = threadIdx.x % nf, rl = threadIdx.x / nf;
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
    if (rl < rpb) {
        int r = rl;
        for (; r + 3 * rpb < rows; r += 4 * rpb) {      // four rows in flight
            f32x4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *p = stats + (long long)(r + k * rpb) * 2 * c + f * 4;
                a[k] = *reinterpret_cast<const f32x4 *>(p);
                b[k] = *reinterpret_cast<const f32x4 *>(p + c);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) { s1[q] += (double)a[k][q]; s2[q] += (double)b[k][q]; }
        }
        for (; r < rows; r += rpb) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(stats + (long long)r * 2 * c + f * 4);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(stats + (long long)r * 2 * c + c + f * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += (double)a[q]; s2[q] += (double)b[q]; }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[threadIdx.x][q] = s1[q]; red[threadIdx.x][4 + q] = s2[q]; }
    doda_sync();
    if (rl == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] = 0.0; s2[q] = 0.0; }
        for (int k = 0; k < rpb; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += red[k * nf + f][q]; s2[q] += red[k * nf + f][4 + q]; }
        }
    }
}

template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_fused_fwd(const typename T::elem *__restrict__ x, long long n_frag, int nf,
                                                         int rpb, const float *__restrict__ stats, int rows, int m,
                                                         float eps, float momentum, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float *__restrict__ mean,
                                                         float *__restrict__ invstd, float *__restrict__ running_mean,
                                                         float *__restrict__ running_var, long long *__restrict__ nbt,
                                                         int relu, typename T::elem *__restrict__ y) {
    __shared__ double red[BN_BLOCK][8];
    __shared__ f32x4 v_mu[BN_FUSED_MAX_C / 4], v_is[BN_FUSED_MAX_C / 4];
    const int c = nf * 4;
    double s1[4], s2[4];
    fused_reduce(stats, rows, c, nf, rpb, red, s1, s2);
    if (threadIdx.x < nf) {
        const int f = threadIdx.x;
        f32x4 mu, is;
        double dmu[4], dvar[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double d = s1[q] / m;
            double var = s2[q] / m - d * d;
            if (var < 0.0) var = 0.0;
            dmu[q] = d;
            dvar[q] = var;
            mu[q] = (float)d;
            is[q] = (float)(1.0 / sqrt(var + (double)eps));
        }
        v_mu[f] = mu;      // (the apply sweep evaluates (x - mean) * invstd * gamma + beta exactly as bn_apply does)
        v_is[f] = is;
        if (blockIdx.x == 0) {
            *reinterpret_cast<f32x4 *>(mean + f * 4) = mu;
            *reinterpret_cast<f32x4 *>(invstd + f * 4) = is;
            if (running_mean) {
                f32x4 rm = *reinterpret_cast<const f32x4 *>(running_mean + f * 4);
                f32x4 rv = *reinterpret_cast<const f32x4 *>(running_var + f * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double unbiased = m > 1 ? dvar[q] * (double)m / (double)(m - 1) : dvar[q];
                    rm[q] = (float)((1.0 - momentum) * (double)rm[q] + momentum * dmu[q]);
                    rv[q] = (float)((1.0 - momentum) * (double)rv[q] + momentum * unbiased);
                }
                *reinterpret_cast<f32x4 *>(running_mean + f * 4) = rm;
                *reinterpret_cast<f32x4 *>(running_var + f * 4) = rv;
            }
            if (f == 0 && nbt) *nbt = *nbt + 1;
        }
    }
    doda_sync();
    for (long long e = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; e < n_frag; e += (long long)gridDim.x * BN_BLOCK) {
        const int f = (int)(e % nf);
        const f32x4 v = T::load4(x + e * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
        const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
        f32x4 o = (v - v_mu[f]) * v_is[f] * ga + be;
        if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = o[q] > 0.f ? o[q] : 0.f;
        }
        T::store4(y + e * 4, o);
    }
}

template <class T>
__global__ __launch_bounds__(BN_BLOCK) void bn_fused_bwd(const typename T::elem *__restrict__ x,
                                                         const typename T::elem *__restrict__ dy, long long n_frag,
                                                         int nf, int rpb, const float *__restrict__ stats, int rows,
                                                         int m, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         int relu, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                         typename T::elem *__restrict__ dx,
                                                         const typename T::elem *__restrict__ add) {
    __shared__ double red[BN_BLOCK][8];
    __shared__ f32x4 v_a[BN_FUSED_MAX_C / 4], v_b[BN_FUSED_MAX_C / 4], v_d[BN_FUSED_MAX_C / 4];
    const int c = nf * 4;
    double s1[4], s2[4];
    fused_reduce(stats, rows, c, nf, rpb, red, s1, s2);
    if (threadIdx.x < nf) {
        const int f = threadIdx.x;
        const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
        f32x4 db, dg, a, bb, dd;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            db[q] = (float)s1[q];
            dg[q] = (float)s2[q];
            a[q] = ga[q] * is[q];
            bb[q] = (float)(s1[q] / m);
            dd[q] = (float)(s2[q] / m);
        }
        v_a[f] = a; v_b[f] = bb; v_d[f] = dd;
        if (blockIdx.x == 0) {
            *reinterpret_cast<f32x4 *>(dbeta + f * 4) = db;
            *reinterpret_cast<f32x4 *>(dgamma + f * 4) = dg;
        }
    }
    doda_sync();
    for (long long e = (long long)blockIdx.x * BN_BLOCK + threadIdx.x; e < n_frag; e += (long long)gridDim.x * BN_BLOCK) {
        const int f = (int)(e % nf);
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + f * 4);
        const f32x4 is = *reinterpret_cast<const f32x4 *>(invstd + f * 4);
        const f32x4 xh = (T::load4(x + e * 4) - mu) * is;
        f32x4 dz = T::load4(dy + e * 4);
        if (relu) {
            const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + f * 4);
            const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + f * 4);
            const f32x4 yv = xh * ga + be;
#pragma unroll
            for (int q = 0; q < 4; ++q) dz[q] = yv[q] > 0.f ? dz[q] : 0.f;
        }
        f32x4 o = v_a[f] * (dz - v_b[f] - xh * v_d[f]);
        if (add) o += T::load4(add + e * 4);
        T::store4(dx + e * 4, o);
    }
}

constexpr long long BN_FUSED_SMALL_ELEMS = 2500000;   // m * c up to which the one-launch form is the default (levels 3-4: 35-46k x 48, 8-11k x 64)
inline bool fused_ok(int rows, int c, int m) {
    // For LARGE tensors off by default: measured in the U-Net step (same box, alternating runs) 7.7-7.8 ms against
    // 6.45-6.54 ms — a 256-workgroup apply sweep is far below the HBM rate the 4096-block one reaches, and more workgroups
    // multiply the redundant partial reads; DODA_BN_FUSED_FINAL=1 switches it on everywhere (parity tests).
    // SMALL tensors (levels 3-4 of the U-Net; the levels below take the one-launch bn_small_*):
    // both launches are launch-floor kernels (5 + 5 us) and the sweep is a few workgroups — one launch (round 3,
    // DODA_BN_FUSED_SMALL=0 switches that off).
    static const bool on = getenv("DODA_BN_FUSED_FINAL") && getenv("DODA_BN_FUSED_FINAL")[0] == '1';
    // DODA_BN_FUSED_SMALL: 0 = off, 1 / unset = the default threshold, larger values = that many elements
    static const long long small_elems = [] {
        const char *e = getenv("DODA_BN_FUSED_SMALL");
        if (!e) return BN_FUSED_SMALL_ELEMS;
        const long long v = atoll(e);
        return v == 1 ? BN_FUSED_SMALL_ELEMS : v;
    }();
    const bool want = on || (long long)m * c <= small_elems;
    return want && c <= BN_FUSED_MAX_C && c / 4 <= BN_BLOCK && (long long)rows * 2 * c * 4 <= BN_FUSED_MAX_PARTIAL_BYTES;
}

Geo make_geo(int c) {
    Geo g;
    g.nf = c / 4;
    g.rpb = BN_BLOCK / g.nf;
    if (g.rpb < 1) g.rpb = 1;
    return g;
}

int n_blocks_for(int m, const Geo &g) {
    int nb = div_up(m, g.rpb * 8);  // >= 8 rows per lane
    if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    return nb;
}

template <class T>
int run_fwd(const void *x_, int m, int c, float eps, float momentum, const float *gamma,
            const float *beta, float *running_mean, float *running_var, long long *nbt, int training,
            int relu, void *y_, float *mean, float *invstd, void *ws, size_t ws_bytes, hipStream_t s) {
    typedef typename T::elem elem;
    const elem *x = (const elem *)x_;
    elem *y = (elem *)y_;
    const Geo g = make_geo(c);
    if (training && m <= BN_SMALL_ROWS) {
        hipLaunchKernelGGL((bn_small_fwd<T>), dim3(c / 4), dim3(BN_BLOCK), 0, s, x, m, c, eps, momentum,
                           gamma, beta, running_mean, running_var, nbt, relu, y, mean, invstd);
        return doda_check_launch();
    }
    if (training) {
        const int nb = n_blocks_for(m, g);
        if (ws_bytes < (size_t)nb * 2 * c * 4) return DODA_ERR_WORKSPACE;
        float *partial = (float *)ws;
        hipLaunchKernelGGL((bn_stats_partial<T>), dim3(nb), dim3(BN_BLOCK), (size_t)g.rpb * 2 * c * 4, s,
                           x, m, c, g, partial);
        hipLaunchKernelGGL((bn_stats_final<T>), dim3(c), dim3(64), 0, s, x, partial, nb, m,
                           c, eps, momentum, mean, invstd, running_mean, running_var, nbt);
    }
    const long long n_frag = (long long)m * g.nf;
    launch_apply<T>(x, n_frag, g.nf, mean, invstd, gamma, beta, relu, y, s);
    return doda_check_launch();
}

template <class T>
int run_bwd(const void *x_, const void *dy_, int m, int c, const float *mean, const float *invstd,
            const float *gamma, const float *beta, int relu, void *dx_, float *dgamma, float *dbeta,
            void *ws, size_t ws_bytes, const void *add_, hipStream_t s, int add_ld = 0) {
    typedef typename T::elem elem;
    const elem *x = (const elem *)x_, *dy = (const elem *)dy_, *add = (const elem *)add_;
    elem *dx = (elem *)dx_;
    const Geo g = make_geo(c);
    if (add_ld == c) add_ld = 0;   // dense
    if (m <= BN_SMALL_ROWS) {
        hipLaunchKernelGGL((bn_small_bwd<T>), dim3(c / 4), dim3(BN_BLOCK), 0, s, x, dy, m, c, mean, invstd,
                           gamma, beta, relu, dx, dgamma, dbeta, add, add_ld);
        return doda_check_launch();
    }
    const int nb = n_blocks_for(m, g);
    if (ws_bytes < (size_t)nb * 2 * c * 4 + (size_t)3 * c * 4) return DODA_ERR_WORKSPACE;
    float *partial = (float *)ws;
    float *coef = partial + (size_t)nb * 2 * c;
    hipLaunchKernelGGL((bn_bwd_partial<T>), dim3(nb), dim3(BN_BLOCK), (size_t)g.rpb * 2 * c * 4, s, x, dy,
                       m, c, g, mean, invstd, gamma, beta, relu, partial);
    hipLaunchKernelGGL(bn_bwd_final, dim3(c), dim3(64), 0, s, partial, nb, m, c, invstd,
                       gamma, dgamma, dbeta, coef);
    const long long n_frag = (long long)m * g.nf;
    launch_bwd_apply<T>(x, dy, n_frag, g.nf, c, mean, invstd, gamma, beta, relu, coef, dx, add, add_ld, s);
    return doda_check_launch();
}

bool bn_args_bad(int m, int c, int elem_bytes) {
    return m <= 0 || c <= 0 || (c % 4) != 0 || c > 1024 || (elem_bytes != 2 && elem_bytes != 4);
}
}  // namespace

extern "C" size_t doda_bn_workspace_bytes(int32_t m, int32_t c) {
    if (m <= 0 || c <= 0 || c % 4) return 256;
    const Geo g = make_geo(c);
    return align_up((size_t)n_blocks_for(m, g) * 2 * c * 4 + (size_t)3 * c * 4, 256);
}

extern "C" int doda_bn_relu_fwd(const void *x, int32_t m, int32_t c, int32_t elem_bytes, float eps,
                                float momentum, const float *gamma, const float *beta,
                                float *running_mean, float *running_var,
                                int64_t *num_batches_tracked, int32_t training, int32_t relu,
                                void *y, float *save_mean, float *save_invstd, void *ws,
                                size_t ws_bytes, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !ws) return DODA_ERR_INVALID;
    if (elem_bytes == 4)
        return run_fwd<F32>(x, m, c, eps, momentum, gamma, beta, running_mean, running_var,
                            (long long *)num_batches_tracked, training, relu, y, save_mean, save_invstd,
                            ws, ws_bytes, as_stream(stream));
    return run_fwd<BF16>(x, m, c, eps, momentum, gamma, beta, running_mean, running_var,
                         (long long *)num_batches_tracked, training, relu, y, save_mean, save_invstd, ws,
                         ws_bytes, as_stream(stream));
}

extern "C" int doda_bn_relu_bwd(const void *x, const void *dy, int32_t m, int32_t c,
                                int32_t elem_bytes, const float *save_mean,
                                const float *save_invstd, const float *gamma, const float *beta,
                                int32_t relu, void *dx, float *dgamma, float *dbeta, void *ws,
                                size_t ws_bytes, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !dy || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !ws)
        return DODA_ERR_INVALID;
    if (elem_bytes == 4)
        return run_bwd<F32>(x, dy, m, c, save_mean, save_invstd, gamma, beta, relu, dx, dgamma, dbeta, ws,
                            ws_bytes, nullptr, as_stream(stream));
    return run_bwd<BF16>(x, dy, m, c, save_mean, save_invstd, gamma, beta, relu, dx, dgamma, dbeta, ws,
                         ws_bytes, nullptr, as_stream(stream));
}

// dx = BN backward + add: `add` ([m, c], dtype of x) is a second gradient of the same x — in a
// pre-activation residual block x feeds both the BatchNorm and the skip connection — so the
// gradient accumulation rides in the apply pass instead of a separate elementwise kernel.
extern "C" int doda_bn_relu_bwd_add(const void *x, const void *dy, int32_t m, int32_t c,
                                       int32_t elem_bytes, const float *save_mean,
                                       const float *save_invstd, const float *gamma, const float *beta,
                                       int32_t relu, const void *add, int32_t add_ld, void *dx, float *dgamma,
                                       float *dbeta, void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !dy || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !ws || !add)
        return DODA_ERR_INVALID;
    if (add_ld < c || add_ld % 4 || ((uintptr_t)add % (4 * (size_t)elem_bytes))) return DODA_ERR_INVALID;
    if (elem_bytes == 4)
        return run_bwd<F32>(x, dy, m, c, save_mean, save_invstd, gamma, beta, relu, dx, dgamma, dbeta, ws,
                            ws_bytes, add, as_stream(stream), add_ld);
    return run_bwd<BF16>(x, dy, m, c, save_mean, save_invstd, gamma, beta, relu, dx, dgamma, dbeta, ws,
                         ws_bytes, add, as_stream(stream), add_ld);
}
// ---- BatchNorm(+ReLU) over statistics that a conv epilogue accumulated ------------------------------
template <class T>
static int run_fwd_stats(const void *x_, int m, int c, const float *stats, int rows, float eps, float momentum,
                         const float *gamma, const float *beta, float *rm, float *rv, long long *nbt, int relu,
                         void *y_, float *mean, float *invstd, hipStream_t s) {
    typedef typename T::elem elem;
    const Geo g = make_geo(c);
    if (fused_ok(rows, c, m)) {      // final + apply in one launch: few partial rows
        const long long nfr = (long long)m * g.nf;
        const int grid = (int)((nfr + BN_BLOCK - 1) / BN_BLOCK < BN_FUSED_BLOCKS ? (nfr + BN_BLOCK - 1) / BN_BLOCK : BN_FUSED_BLOCKS);
        hipLaunchKernelGGL((bn_fused_fwd<T>), dim3(grid), dim3(BN_BLOCK), 0, s, (const elem *)x_, nfr, g.nf, g.rpb, stats, rows,
                           m, eps, momentum, gamma, beta, mean, invstd, rm, rv, nbt, relu, (elem *)y_);
        return doda_check_launch();
    }
    const long long n_frag = (long long)m * g.nf;
    hipLaunchKernelGGL(bn_fwd_final_stats, dim3(c / 4), dim3(BN_BLOCK), 0, s, stats, rows, m, c, eps, momentum, mean,
                       invstd, rm, rv, nbt);
    launch_apply<T>((const elem *)x_, n_frag, g.nf, mean, invstd, gamma, beta, relu, (elem *)y_, s);
    return doda_check_launch();
}

template <class T>
static int run_bwd_stats(const void *x_, const void *dy_, int m, int c, const float *stats, int rows,
                         const float *mean, const float *invstd, const float *gamma, const float *beta, int relu,
                         const void *add_, void *dx_, float *dgamma, float *dbeta, float *coef, hipStream_t s,
                         int add_ld = 0) {
    typedef typename T::elem elem;
    const Geo g = make_geo(c);
    if (add_ld == c || !add_) add_ld = 0;   // dense
    if (!add_ld && fused_ok(rows, c, m)) {      // final + apply in one launch: few partial rows (opt-in; dense `add` only)
        const long long nfr = (long long)m * g.nf;
        const int grid = (int)((nfr + BN_BLOCK - 1) / BN_BLOCK < BN_FUSED_BLOCKS ? (nfr + BN_BLOCK - 1) / BN_BLOCK : BN_FUSED_BLOCKS);
        hipLaunchKernelGGL((bn_fused_bwd<T>), dim3(grid), dim3(BN_BLOCK), 0, s, (const elem *)x_, (const elem *)dy_, nfr, g.nf,
                           g.rpb, stats, rows, m, mean, invstd, gamma, beta, relu, dgamma, dbeta, (elem *)dx_, (const elem *)add_);
        return doda_check_launch();
    }
    const long long n_frag = (long long)m * g.nf;
    hipLaunchKernelGGL(bn_bwd_final_stats, dim3(c / 4), dim3(BN_BLOCK), 0, s, stats, rows, m, c, invstd, gamma, dgamma,
                       dbeta, coef);
    launch_bwd_apply<T>((const elem *)x_, (const elem *)dy_, n_frag, g.nf, c, mean, invstd, gamma, beta, relu, coef,
                        (elem *)dx_, (const elem *)add_, add_ld, s);
    return doda_check_launch();
}

extern "C" int doda_bn_relu_fwd_stats(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const float *stats,
                                      int32_t stats_rows, float eps, float momentum, const float *gamma,
                                      const float *beta, float *running_mean, float *running_var,
                                      int64_t *num_batches_tracked, int32_t relu, void *y, float *save_mean,
                                      float *save_invstd, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !y || !stats || stats_rows <= 0 || !gamma || !beta || !save_mean || !save_invstd) return DODA_ERR_INVALID;
    if (elem_bytes == 4)
        return run_fwd_stats<F32>(x, m, c, stats, stats_rows, eps, momentum, gamma, beta, running_mean, running_var,
                                  (long long *)num_batches_tracked, relu, y, save_mean, save_invstd, as_stream(stream));
    return run_fwd_stats<BF16>(x, m, c, stats, stats_rows, eps, momentum, gamma, beta, running_mean, running_var,
                               (long long *)num_batches_tracked, relu, y, save_mean, save_invstd, as_stream(stream));
}

extern "C" int doda_bn_relu_apply(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const float *mean,
                                  const float *invstd, const float *gamma, const float *beta, int32_t relu, void *y,
                                  doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !y || !mean || !invstd || !gamma || !beta) return DODA_ERR_INVALID;
    const Geo g = make_geo(c);
    const long long n_frag = (long long)m * g.nf;
    if (elem_bytes == 4)
        launch_apply<F32>((const float *)x, n_frag, g.nf, mean, invstd, gamma, beta, relu, (float *)y, as_stream(stream));
    else
        launch_apply<BF16>((const unsigned short *)x, n_frag, g.nf, mean, invstd, gamma, beta, relu, (unsigned short *)y,
                           as_stream(stream));
    return doda_check_launch();
}

// ABI 6: the reduction alone — the apply pass rides in the consuming convolution's prologue (spconv_tile.hip, PRE)
extern "C" int doda_bn_fwd_final(const float *stats, int32_t stats_rows, int32_t m, int32_t c, float eps, float momentum,
                                 float *running_mean, float *running_var, int64_t *num_batches_tracked, float *save_mean,
                                 float *save_invstd, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, 2)) return DODA_ERR_UNSUPPORTED;
    if (!stats || stats_rows <= 0 || !save_mean || !save_invstd || (!running_mean != !running_var)) return DODA_ERR_INVALID;
    hipLaunchKernelGGL(bn_fwd_final_stats, dim3(c / 4), dim3(BN_BLOCK), 0, as_stream(stream), stats, stats_rows, m, c, eps,
                       momentum, save_mean, save_invstd, running_mean, running_var, (long long *)num_batches_tracked);
    return doda_check_launch();
}

// ABI 9: BatchNorm(+ReLU) over TOTALS that conv epilogues accumulated (doda_conv_epilogue.stats_totals): ONE launch.
// totals_b / c_a: the columns [c_a, c) come from a second producer (a channel concatenation); totals_b null: c_a is ignored.
extern "C" int doda_bn_relu_fwd_totals(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const double *totals,
                                       const double *totals_b, int32_t c_a, float eps, float momentum, const float *gamma,
                                       const float *beta, float *running_mean, float *running_var,
                                       int64_t *num_batches_tracked, int32_t relu, void *y, float *save_mean,
                                       float *save_invstd, doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes) || c > BN_TOT_MAX_C) return DODA_ERR_UNSUPPORTED;
    if (!x || !y || !totals || !gamma || !beta || !save_mean || !save_invstd || (!running_mean != !running_var))
        return DODA_ERR_INVALID;
    if (totals_b && (c_a <= 0 || c_a >= c)) return DODA_ERR_INVALID;
    TotArgs t;
    t.ta = totals; t.tb = totals_b; t.ca = totals_b ? c_a : c; t.m = m; t.eps = eps; t.momentum = momentum;
    t.rm = running_mean; t.rv = running_var; t.nbt = (long long *)num_batches_tracked;
    t.out_a = save_mean; t.out_b = save_invstd;
    const Geo g = make_geo(c);
    const long long n_frag = (long long)m * g.nf;
    if (elem_bytes == 4)
        launch_apply_tot<F32>((const float *)x, n_frag, g.nf, gamma, beta, relu, (float *)y, t, as_stream(stream));
    else
        launch_apply_tot<BF16>((const unsigned short *)x, n_frag, g.nf, gamma, beta, relu, (unsigned short *)y, t, as_stream(stream));
    return doda_check_launch();
}

extern "C" int doda_bn_relu_bwd_totals(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes,
                                       const double *totals, const float *save_mean, const float *save_invstd,
                                       const float *gamma, const float *beta, int32_t relu, const void *add, int32_t add_ld,
                                       void *dx, float *dgamma, float *dbeta, doda_stream_t stream) {
    if (add && (add_ld < c || add_ld % 4 || ((uintptr_t)add % (4 * (size_t)elem_bytes)))) return DODA_ERR_INVALID;
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes) || c > BN_TOT_MAX_C) return DODA_ERR_UNSUPPORTED;
    if (!x || !dy || !dx || !totals || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta) return DODA_ERR_INVALID;
    TotArgs t;
    t.ta = totals; t.tb = nullptr; t.ca = c; t.m = m; t.eps = 0.f; t.momentum = 0.f;
    t.rm = nullptr; t.rv = nullptr; t.nbt = nullptr;
    t.out_a = dgamma; t.out_b = dbeta;
    const Geo g = make_geo(c);
    const long long n_frag = (long long)m * g.nf;
    if (add_ld == c || !add) add_ld = 0;   // dense
    if (elem_bytes == 4)
        launch_bwd_apply_tot<F32>((const float *)x, (const float *)dy, n_frag, g.nf, c, save_mean, save_invstd, gamma, beta, relu,
                                  (float *)dx, (const float *)add, add_ld, t, as_stream(stream));
    else
        launch_bwd_apply_tot<BF16>((const unsigned short *)x, (const unsigned short *)dy, n_frag, g.nf, c, save_mean, save_invstd,
                                   gamma, beta, relu, (unsigned short *)dx, (const unsigned short *)add, add_ld, t, as_stream(stream));
    return doda_check_launch();
}

extern "C" int doda_bn_relu_bwd_stats(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes,
                                         const float *stats, int32_t stats_rows, const float *save_mean,
                                         const float *save_invstd, const float *gamma, const float *beta, int32_t relu,
                                         const void *add, int32_t add_ld, void *dx, float *dgamma, float *dbeta,
                                         float *coef_ws, doda_stream_t stream) {
    if (add && (add_ld < c || add_ld % 4 || ((uintptr_t)add % (4 * (size_t)elem_bytes)))) return DODA_ERR_INVALID;
    if (m == 0) return DODA_OK;
    if (bn_args_bad(m, c, elem_bytes)) return DODA_ERR_UNSUPPORTED;
    if (!x || !dy || !dx || !stats || stats_rows <= 0 || !gamma || !beta || !save_mean || !save_invstd || !dgamma ||
        !dbeta || !coef_ws)
        return DODA_ERR_INVALID;
    if (elem_bytes == 4)
        return run_bwd_stats<F32>(x, dy, m, c, stats, stats_rows, save_mean, save_invstd, gamma, beta, relu, add, dx,
                                  dgamma, dbeta, coef_ws, as_stream(stream), add_ld);
    return run_bwd_stats<BF16>(x, dy, m, c, stats, stats_rows, save_mean, save_invstd, gamma, beta, relu, add, dx,
                               dgamma, dbeta, coef_ws, as_stream(stream), add_ld);
}
