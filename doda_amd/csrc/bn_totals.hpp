// BatchNorm over fp64 TOTALS (ABI 9), shared by the BatchNorm sweeps (bn.hip) and by the convolution kernels that fold a
// BatchNorm into their gather (ABI 11, spconv_gather.hip conv_fast<..., PRE>: reference model/unet_block.py:23-30,46-49,67-79 —
// every conv sits behind BatchNorm1d -> ReLU).  One place for the arithmetic, so a BatchNorm applied by its own sweep and the
// same BatchNorm applied inside the consuming conv's gather give the same bits.  gfx950 only.
#pragma once
#include "common.hpp"

constexpr int BN_TOT_MAX_C = 256;
constexpr int BN_TOT_SLOTS = 8;      // = DODA_STATS_SLOTS (spconv_common.hpp)

// ta / tb: the totals of the producers of the columns [0, ca) and [ca, c) (tb null: one producer) — a channel concatenation.
struct TotArgs {
    const double *ta = nullptr, *tb = nullptr;
    int ca = 0, m = 0;
    float eps = 0.f, momentum = 0.f;
    float *rm = nullptr, *rv = nullptr;           // forward: running statistics or null
    long long *nbt = nullptr;
    float *out_a = nullptr, *out_b = nullptr;     // forward: save_mean, save_invstd; backward: dgamma, dbeta
    int accum = 0;                                // backward: dgamma / dbeta are ADDED to (a second backward pass of one optimizer step)
};

#if defined(__HIPCC__)
__device__ __forceinline__ void tot_sums(const TotArgs &t, int c, int ch, double &s1, double &s2) {
    const bool first = ch < t.ca;
    const double *src = first ? t.ta : t.tb;
    const int cw = first ? t.ca : c - t.ca, cc = first ? ch : ch - t.ca;
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int k = 0; k < BN_TOT_SLOTS; ++k) {      // (layout: spconv_common.hpp stats_emit — a 128-byte line per four channels)
        s1 += src[((size_t)(k * 2 + 0) * (cw / 4) + cc / 4) * 16 + (cc & 3)];
        s2 += src[((size_t)(k * 2 + 1) * (cw / 4) + cc / 4) * 16 + (cc & 3)];
    }
}

// Forward, channel `ch`: batch mean / 1 / sqrt(biased variance + eps) from the totals; `publish` (one workgroup of the launch):
// save_mean / save_invstd, the running statistics (momentum, unbiased variance) and num_batches_tracked.
__device__ __forceinline__ void tot_fwd_channel(const TotArgs &t, int c, int ch, bool publish, float &mu, float &is) {
    double s1, s2;
    tot_sums(t, c, ch, s1, s2);
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
}

// Backward, channel `ch`: dx = ca * (dz - cb - xhat * cd); `publish`: dbeta = sum dz, dgamma = sum dz * xhat.
__device__ __forceinline__ void tot_bwd_channel(const TotArgs &t, int c, int ch, bool publish, float invstd, float gamma,
                                                float &ca, float &cb, float &cd) {
    double s1, s2;
    tot_sums(t, c, ch, s1, s2);
    ca = gamma * invstd;
    cb = (float)(s1 / t.m);
    cd = (float)(s2 / t.m);
    if (publish) {
        if (t.accum) {
            t.out_b[ch] += (float)s1;
            t.out_a[ch] += (float)s2;
        } else {
            t.out_b[ch] = (float)s1;                // dbeta
            t.out_a[ch] = (float)s2;                // dgamma
        }
    }
}

// One element of the sweeps, in the order the standalone kernels always used (-ffp-contract=off: every operation rounds):
//   forward  y  = [relu]((x - mu) * is * ga + be)
//   backward dx = ca * ([yv > 0] dz - cb - xh * cd),  xh = (x - mu) * is, yv = xh * ga + be
__device__ __forceinline__ float bn_fwd_elem(float x, float mu, float is, float ga, float be) { return (x - mu) * is * ga + be; }
__device__ __forceinline__ float bn_bwd_elem(float x, float dz, float mu, float is, float ga, float be, float ca, float cb, float cd,
                                             int relu) {
    const float xh = (x - mu) * is;
    if (relu) {
        const float yv = xh * ga + be;
        dz = yv > 0.f ? dz : 0.f;
    }
    return ca * (dz - cb - xh * cd);
}
#endif
