// Sparse-convolution arithmetic with bf16 feature storage (BASELINE config 2: bf16 training).
// Same output-stationary structure as spconv_f32.hip; features are bf16 in HBM (half the
// feature bytes), weights arrive fp32 and are rounded to bf16 by a tiny pre-pack kernel that
// lays them out in MFMA B-fragment order (one 8-byte load per lane per fragment), products are
// accumulated in fp32 by v_mfma_f32_16x16x16_bf16, outputs are rounded to bf16 (RNE) once.
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// packed[o][cc][nb][lane] = 4 bf16: B_o[cc*16 + 4g + e][nb*16 + i], e = 0..3, lane = 16g + i
template <int WL>
__global__ __launch_bounds__(256) void pack_weights(const float *__restrict__ w, int K, int kc,
                                                    int nc, int n_chunk, int NB,
                                                    s16x4 *__restrict__ packed) {
    const long long total = (long long)K * n_chunk * NB * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int nb = (int)(r % NB); r /= NB;
    const int cc = (int)(r % n_chunk);
    const int o = (int)(r / n_chunk);
    const int i = lane & 15, g = lane >> 4;
    const int col = nb * 16 + i;
    s16x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cc * 16 + 4 * g + q;
        float f = 0.f;
        if (c < kc && col < nc) {
            if (WL == 0) f = w[((long long)o * kc + c) * nc + col];
            else {
                const int oo = (WL == 2) ? (K - 1 - o) : o;
                f = w[((long long)oo * nc + col) * kc + c];
            }
        }
        v[q] = (short)f2bf(f);
    }
    packed[e] = v;
}

template <int NB, int S>
__global__ __launch_bounds__(256) void conv_gather_bf16(const unsigned short *__restrict__ x,
                                                        int kc, const s16x4 *__restrict__ wp,
                                                        int nc, const int32_t *__restrict__ tbl,
                                                        int ld, int K, int n_out,
                                                        unsigned short *__restrict__ y) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long long t0 = ((long long)blockIdx.x * 4 + wid) * (16 * S);
    if (t0 >= n_out) return;

    f32x4 acc[S][NB];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_chunk = (kc + 15) / 16;
    const bool vec_ok = (kc & 3) == 0;  // 8-byte aligned channel quads

    for (int o = 0; o < K; ++o) {
        int idx[S];
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const long long t = t0 + s * 16 + i;
            idx[s] = t < n_out ? tbl[(long long)o * ld + t] : -1;
            any |= (__ballot(idx[s] >= 0) != 0ull);
        }
        if (!any) continue;
        for (int cc = 0; cc < n_chunk; ++cc) {
            s16x4 a[S];
            const int c0 = cc * 16 + 4 * g;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                a[s] = (s16x4){0, 0, 0, 0};
                if (idx[s] >= 0) {
                    const unsigned short *row = x + (long long)idx[s] * kc;
                    if (vec_ok) {
                        if (c0 < kc) a[s] = *reinterpret_cast<const s16x4 *>(row + c0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (c0 + q < kc) a[s][q] = (short)row[c0 + q];
                    }
                }
            }
            const s16x4 *wrow = wp + (((long long)o * n_chunk + cc) * NB) * 64 + lane;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const s16x4 b = wrow[nb * 64];
#pragma unroll
                for (int s = 0; s < S; ++s)
                    acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[s], b, acc[s][nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long t = t0 + s * 16 + 4 * g + r;
            if (t < n_out) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int col = nb * 16 + i;
                    if (col < nc) y[t * nc + col] = f2bf(acc[s][nb][r]);
                }
            }
        }
}

// weight gradient, rows as the MFMA k dimension (4 rows per lane group)
template <int OG, int TA, int TB>
__global__ __launch_bounds__(256) void wgrad_bf16(const unsigned short *__restrict__ a, int ca,
                                                  const unsigned short *__restrict__ b, int cb,
                                                  const int32_t *__restrict__ tbl, int ld, int K,
                                                  int n_rows, int rows_per_chunk, int n_ogb,
                                                  int n_tag, float *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    int yy = blockIdx.y;
    const int ogb = yy % n_ogb; yy /= n_ogb;
    const int tag = yy % n_tag;
    const int tbg = yy / n_tag;
    const int o0 = (ogb * 4 + wid) * OG;
    const int ta0 = tag * TA, tb0 = tbg * TB;

    f32x4 acc[OG][TA][TB];
#pragma unroll
    for (int oo = 0; oo < OG; ++oo)
#pragma unroll
        for (int x_ = 0; x_ < TA; ++x_)
#pragma unroll
            for (int y_ = 0; y_ < TB; ++y_) acc[oo][x_][y_] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long r_begin = (long long)blockIdx.x * rows_per_chunk;
    long long r_end = r_begin + rows_per_chunk;
    if (r_end > n_rows) r_end = n_rows;

    if (o0 < K) {
        for (long long r0 = r_begin; r0 < r_end; r0 += 16) {
            // lane group g owns rows r0 + 4g + q, q = 0..3 (the 4 k-slots of its fragment)
            s16x4 bv[TB];
#pragma unroll
            for (int y_ = 0; y_ < TB; ++y_) {
                const int col = (tb0 + y_) * 16 + i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long row = r0 + 4 * g + q;
                    bv[y_][q] = (row < r_end && col < cb) ? (short)b[row * cb + col] : (short)0;
                }
            }
#pragma unroll
            for (int oo = 0; oo < OG; ++oo) {
                const int o = o0 + oo;
                if (o < K) {
                    int idx[4];
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long row = r0 + 4 * g + q;
                        idx[q] = row < r_end ? tbl[(long long)o * ld + row] : -1;
                        any |= idx[q] >= 0;
                    }
                    if (__ballot(any) == 0ull) continue;
#pragma unroll
                    for (int x_ = 0; x_ < TA; ++x_) {
                        const int ci = (ta0 + x_) * 16 + i;
                        s16x4 av;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            av[q] = (idx[q] >= 0 && ci < ca) ? (short)a[(long long)idx[q] * ca + ci] : (short)0;
#pragma unroll
                        for (int y_ = 0; y_ < TB; ++y_)
                            acc[oo][x_][y_] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(
                                av, bv[y_], acc[oo][x_][y_], 0, 0, 0);
                    }
                }
            }
        }
    }
    float *out = partial + (long long)blockIdx.x * K * ca * cb;
#pragma unroll
    for (int oo = 0; oo < OG; ++oo) {
        const int o = o0 + oo;
        if (o < K) {
#pragma unroll
            for (int x_ = 0; x_ < TA; ++x_)
#pragma unroll
                for (int y_ = 0; y_ < TB; ++y_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = (ta0 + x_) * 16 + 4 * g + r, co = (tb0 + y_) * 16 + i;
                        if (ci < ca && co < cb)
                            out[((long long)o * ca + ci) * cb + co] = acc[oo][x_][y_][r];
                    }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_bf(const float *__restrict__ partial, int R,
                                                       long long n_elem, float *__restrict__ dw) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += partial[(long long)r * n_elem + e];
    dw[e] = s;
}

struct WgradPlan {
    int OG, TA, TB, n_ogb, n_tag, n_tbg, R, rows_per_chunk;
};

WgradPlan plan_wgrad(int K, int ca, int cb, int n_rows) {
    WgradPlan p;
    const int ta = (ca + 15) / 16, tb = (cb + 15) / 16;
    p.TA = (ta % 2 == 0) ? 2 : 1;
    p.TB = (tb % 2 == 0) ? 2 : 1;
    p.OG = (p.TA * p.TB == 4) ? 1 : 2;
    p.n_ogb = div_up(K, 4 * p.OG);
    p.n_tag = ta / p.TA;
    p.n_tbg = tb / p.TB;
    const int gy = p.n_ogb * p.n_tag * p.n_tbg;
    int R = 1024 / gy;
    if (R < 1) R = 1;
    if (R > 128) R = 128;
    const int max_r = div_up(n_rows > 0 ? n_rows : 1, 64);
    if (R > max_r) R = max_r;
    p.rows_per_chunk = div_up(div_up(n_rows > 0 ? n_rows : 1, R), 16) * 16;
    p.R = div_up(n_rows > 0 ? n_rows : 1, p.rows_per_chunk);
    return p;
}

template <int NB, int S>
int launch_gather_bf16(const unsigned short *x, int kc, const s16x4 *wp, int nc,
                       const int32_t *tbl, int ld, int K, int n_out, unsigned short *y,
                       hipStream_t s) {
    hipLaunchKernelGGL((conv_gather_bf16<NB, S>), dim3(div_up(n_out, 4 * 16 * S)), dim3(256), 0, s,
                       x, kc, wp, nc, tbl, ld, K, n_out, y);
    return doda_check_launch();
}
}  // namespace

extern "C" size_t doda_spconv_gather_bf16_workspace_bytes(int32_t K, int32_t kc, int32_t nc) {
    if (K <= 0 || kc <= 0 || nc <= 0) return 0;
    return align_up((size_t)K * ((kc + 15) / 16) * ((nc + 15) / 16) * 64 * 8, 256);
}

extern "C" int doda_spconv_gather_bf16(const uint16_t *x, int32_t kc, const float *w, int32_t nc,
                                       const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out,
                                       uint16_t *y, int32_t w_layout, void *ws, size_t ws_bytes,
                                       doda_stream_t stream) {
    if (kc <= 0 || nc <= 0 || K <= 0 || n_out < 0 || ld < n_out) return DODA_ERR_INVALID;
    if (w_layout < 0 || w_layout > 2) return DODA_ERR_INVALID;
    if (n_out == 0) return DODA_OK;
    if (!x || !w || !tbl || !y || !ws) return DODA_ERR_INVALID;
    if (nc > 256) return DODA_ERR_UNSUPPORTED;
    if (ws_bytes < doda_spconv_gather_bf16_workspace_bytes(K, kc, nc)) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const int n_chunk = (kc + 15) / 16, nb = (nc + 15) / 16;
    s16x4 *wp = (s16x4 *)ws;
    const long long total = (long long)K * n_chunk * nb * 64;
    const dim3 pg(div_up(total, 256)), pb(256);
    if (w_layout == 0) hipLaunchKernelGGL((pack_weights<0>), pg, pb, 0, s, w, K, kc, nc, n_chunk, nb, wp);
    else if (w_layout == 1) hipLaunchKernelGGL((pack_weights<1>), pg, pb, 0, s, w, K, kc, nc, n_chunk, nb, wp);
    else hipLaunchKernelGGL((pack_weights<2>), pg, pb, 0, s, w, K, kc, nc, n_chunk, nb, wp);
#define DODA_G(NB, S) return launch_gather_bf16<NB, S>(x, kc, wp, nc, tbl, ld, K, n_out, y, s)
    switch (nb) {
        case 1: DODA_G(1, 4);
        case 2: DODA_G(2, 4);
        case 3: DODA_G(3, 2);
        case 4: DODA_G(4, 2);
        case 5: DODA_G(5, 2);
        case 6: DODA_G(6, 2);
        case 7: DODA_G(7, 2);
        case 8: DODA_G(8, 2);
        case 9: DODA_G(9, 1);
        case 10: DODA_G(10, 1);
        case 11: DODA_G(11, 1);
        case 12: DODA_G(12, 1);
        case 13: DODA_G(13, 1);
        case 14: DODA_G(14, 1);
        case 15: DODA_G(15, 1);
        case 16: DODA_G(16, 1);
    }
#undef DODA_G
    return DODA_ERR_UNSUPPORTED;
}

extern "C" int doda_spconv_wgrad_bf16(const uint16_t *a, int32_t ca, const uint16_t *b, int32_t cb,
                                      const int32_t *tbl, int32_t ld, int32_t K, int32_t n_rows,
                                      float *dw, void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (ca <= 0 || cb <= 0 || K <= 0 || n_rows < 0 || ld < n_rows || !dw) return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    const long long n_elem = (long long)K * ca * cb;
    if (n_rows == 0) {
        hipMemsetAsync(dw, 0, (size_t)n_elem * 4, s);
        return DODA_OK;
    }
    if (!a || !b || !tbl || !ws) return DODA_ERR_INVALID;
    const WgradPlan p = plan_wgrad(K, ca, cb, n_rows);
    if (ws_bytes < (size_t)p.R * n_elem * 4) return DODA_ERR_WORKSPACE;
    float *partial = (float *)ws;
    const dim3 grid(p.R, p.n_ogb * p.n_tag * p.n_tbg), block(256);
#define DODA_W(OG, TA, TB)                                                                         \
    hipLaunchKernelGGL((wgrad_bf16<OG, TA, TB>), grid, block, 0, s, a, ca, b, cb, tbl, ld, K,    \
                       n_rows, p.rows_per_chunk, p.n_ogb, p.n_tag, partial)
    if (p.OG == 2 && p.TA == 1 && p.TB == 1) DODA_W(2, 1, 1);
    else if (p.OG == 2 && p.TA == 2 && p.TB == 1) DODA_W(2, 2, 1);
    else if (p.OG == 2 && p.TA == 1 && p.TB == 2) DODA_W(2, 1, 2);
    else if (p.OG == 1 && p.TA == 2 && p.TB == 2) DODA_W(1, 2, 2);
    else return DODA_ERR_UNSUPPORTED;
#undef DODA_W
    int st = doda_check_launch();
    if (st != DODA_OK) return st;
    hipLaunchKernelGGL(wgrad_reduce_bf, dim3(div_up(n_elem, 256)), dim3(256), 0, s, partial, p.R,
                       n_elem, dw);
    return doda_check_launch();
}
