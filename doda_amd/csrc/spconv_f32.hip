// Sparse-convolution arithmetic, fp32: gather -> MFMA -> single store per output row.
//
// Replaces spconv v1.2's indice_conv / indice_conv_backward (per offset: sparse_gather kernel,
// cuBLAS mm, sparse_scatter_add kernel; reference call sites model/unet_block.py:26,29,48,70,78).
// Design (MI355X-first, not a translation):
//   * OUTPUT-STATIONARY.  A wave owns S subtiles of 16 consecutive output rows; for every kernel
//     offset o it gathers the rows tbl[o][t] of x straight into the A-operand layout of
//     v_mfma_f32_16x16x4_f32 (lane (i,g) = row i, 16-byte channel quad g: one coalesced-per-row
//     dwordx4 per lane feeds four MFMA k-steps), multiplies by W[o] and accumulates in registers.
//     Every output row is written exactly once: no gather/scatter buffers in HBM, no atomics,
//     bit-deterministic.  Offsets whose 16 rows are all absent are skipped wave-uniformly.
//   * The same kernel serves forward (w_layout 0), SubM data-grad (w_layout 2: W[K-1-o]^T on the
//     same table — the SubM table is its own transpose under offset mirroring), and the
//     data-grads of down2 / inverse convolutions (w_layout 1 with the child / parent tables).
//   * fp32 MFMA is an exact k-ordered fmaf chain (guide §3), so results differ from a CPU GEMM
//     only by summation order.
// Bound: HBM for the table + feature traffic (B_f = 4(M_in*Cin + M_out*Cout) + 4*K*Cin*Cout +
// 4*K*M_out for the table), fp32-MFMA for the dense 27-offset contraction (157 TF peak).
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// B-operand element B_o[c][col] for the three weight layouts.
template <int WL>
__device__ __forceinline__ float load_b(const float *__restrict__ w, int o, int K, int kc, int nc,
                                        int c, int col) {
    if (c >= kc || col >= nc) return 0.f;
    if (WL == 0) return w[((long long)o * kc + c) * nc + col];
    const int oo = (WL == 2) ? (K - 1 - o) : o;
    return w[((long long)oo * nc + col) * kc + c];
}

template <int NB, int S, int WL>
__global__ __launch_bounds__(256) void conv_gather_f32(const float *__restrict__ x, int kc,
                                                       const float *__restrict__ w, int nc,
                                                       const int32_t *__restrict__ tbl, int ld,
                                                       int K, int n_out, float *__restrict__ y) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long long t0 = ((long long)blockIdx.x * 4 + wid) * (16 * S);
    if (t0 >= n_out) return;

    f32x4 acc[S][NB];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool small_k = kc <= 4;           // input conv (Cin = 3): one k-step, channel = g
    const bool vec_ok = (kc & 3) == 0;      // rows are 16-byte aligned
    const int n_chunk = small_k ? 1 : (kc + 15) / 16;

    for (int o = 0; o < K; ++o) {
        int idx[S];
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const long long t = t0 + s * 16 + i;
            idx[s] = t < n_out ? tbl[(long long)o * ld + t] : -1;
            any |= (__ballot(idx[s] >= 0) != 0ull);
        }
        if (!any) continue;  // wave-uniform

        for (int cc = 0; cc < n_chunk; ++cc) {
            float a[S][4];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                a[s][0] = a[s][1] = a[s][2] = a[s][3] = 0.f;
                if (idx[s] >= 0) {
                    const float *row = x + (long long)idx[s] * kc;
                    if (small_k) {
                        if (g < kc) a[s][0] = row[g];
                    } else if (vec_ok) {
                        const int c0 = cc * 16 + 4 * g;
                        if (c0 < kc) {
                            const float4 v = *reinterpret_cast<const float4 *>(row + c0);
                            a[s][0] = v.x; a[s][1] = v.y; a[s][2] = v.z; a[s][3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int st = 0; st < 4; ++st) {
                            const int c = cc * 16 + 4 * g + st;
                            if (c < kc) a[s][st] = row[c];
                        }
                    }
                }
            }
            const int n_step = small_k ? 1 : 4;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (st < n_step) {
                    const int c = small_k ? g : cc * 16 + 4 * g + st;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float b = load_b<WL>(w, o, K, kc, nc, c, nb * 16 + i);
#pragma unroll
                        for (int s = 0; s < S; ++s)
                            acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][st], b,
                                                                              acc[s][nb], 0, 0, 0);
                    }
                }
            }
        }
    }

    // D layout: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long t = t0 + s * 16 + 4 * g + r;
            if (t < n_out) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int col = nb * 16 + i;
                    if (col < nc) y[t * nc + col] = acc[s][nb][r];
                }
            }
        }
}

template <int NB, int S>
int launch_gather(const float *x, int kc, const float *w, int nc, const int32_t *tbl, int ld,
                  int K, int n_out, float *y, int wl, hipStream_t s) {
    const int rows_per_block = 4 * 16 * S;
    const dim3 grid(div_up(n_out, rows_per_block)), block(256);
    switch (wl) {
        case 0: hipLaunchKernelGGL((conv_gather_f32<NB, S, 0>), grid, block, 0, s, x, kc, w, nc, tbl, ld, K, n_out, y); break;
        case 1: hipLaunchKernelGGL((conv_gather_f32<NB, S, 1>), grid, block, 0, s, x, kc, w, nc, tbl, ld, K, n_out, y); break;
        case 2: hipLaunchKernelGGL((conv_gather_f32<NB, S, 2>), grid, block, 0, s, x, kc, w, nc, tbl, ld, K, n_out, y); break;
        default: return DODA_ERR_INVALID;
    }
    return doda_check_launch();
}

// ---- weight gradient ------------------------------------------------------------------------
// dw[o][ci][co] = sum_t a[tbl[o][t]][ci] * b[t][co].  MFMA with the ROW index as the k
// dimension: A[i = ci][k = row], B[k = row][j = co]; a wave holds OG offsets x TA x TB 16x16
// accumulator tiles and streams a chunk of rows.  Partials [R][K][ca][cb] are reduced by a
// second kernel in fixed order (deterministic; no float atomics).
template <int OG, int TA, int TB>
__global__ __launch_bounds__(256) void wgrad_f32(const float *__restrict__ a, int ca,
                                                 const float *__restrict__ b, int cb,
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
            float bv[4][TB];
            long long rows[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                rows[st] = r0 + 4 * st + g;
                const bool ok = rows[st] < r_end;
#pragma unroll
                for (int y_ = 0; y_ < TB; ++y_) {
                    const int col = (tb0 + y_) * 16 + i;
                    bv[st][y_] = (ok && col < cb) ? b[rows[st] * cb + col] : 0.f;
                }
            }
#pragma unroll
            for (int oo = 0; oo < OG; ++oo) {
                const int o = o0 + oo;
                if (o < K) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int idx = rows[st] < r_end ? tbl[(long long)o * ld + rows[st]] : -1;
                        if (__ballot(idx >= 0) == 0ull) continue;  // wave-uniform
#pragma unroll
                        for (int x_ = 0; x_ < TA; ++x_) {
                            const int ci = (ta0 + x_) * 16 + i;
                            const float av = (idx >= 0 && ci < ca) ? a[(long long)idx * ca + ci] : 0.f;
#pragma unroll
                            for (int y_ = 0; y_ < TB; ++y_)
                                acc[oo][x_][y_] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    av, bv[st][y_], acc[oo][x_][y_], 0, 0, 0);
                        }
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

__global__ __launch_bounds__(256) void wgrad_reduce(const float *__restrict__ partial, int R,
                                                    long long n_elem, float *__restrict__ dw) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += partial[(long long)r * n_elem + e];
    dw[e] = s;
}

struct WgradPlan {
    int OG, TA, TB;       // per-wave accumulator tiling
    int n_ogb, n_tag, n_tbg;
    int R, rows_per_chunk;
};

WgradPlan plan_wgrad(int K, int ca, int cb, int n_rows) {
    WgradPlan p;
    const int ta = (ca + 15) / 16, tb = (cb + 15) / 16;
    p.TA = (ta % 2 == 0) ? 2 : 1;
    p.TB = (tb % 2 == 0) ? 2 : 1;
    p.OG = (p.TA * p.TB == 1) ? 2 : 1;
    if (p.TA * p.TB == 2) p.OG = 2;
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
}  // namespace

extern "C" int doda_spconv_gather_f32(const float *x, int32_t kc, const float *w, int32_t nc,
                                      const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out,
                                      float *y, int32_t w_layout, doda_stream_t stream) {
    if (kc <= 0 || nc <= 0 || K <= 0 || n_out < 0 || ld < n_out) return DODA_ERR_INVALID;
    if (n_out == 0) return DODA_OK;
    if (!x || !w || !tbl || !y) return DODA_ERR_INVALID;
    if (nc > 256) return DODA_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const int nb = (nc + 15) / 16;
#define DODA_G(NB, S) return launch_gather<NB, S>(x, kc, w, nc, tbl, ld, K, n_out, y, w_layout, s)
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

extern "C" size_t doda_spconv_wgrad_workspace_bytes(int32_t K, int32_t ca, int32_t cb,
                                                    int32_t n_rows) {
    if (K <= 0 || ca <= 0 || cb <= 0) return 0;
    const WgradPlan p = plan_wgrad(K, ca, cb, n_rows);
    return align_up((size_t)p.R * K * ca * cb * 4, 256);
}

extern "C" int doda_spconv_wgrad_f32(const float *a, int32_t ca, const float *b, int32_t cb,
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
    hipLaunchKernelGGL((wgrad_f32<OG, TA, TB>), grid, block, 0, s, a, ca, b, cb, tbl, ld, K,     \
                       n_rows, p.rows_per_chunk, p.n_ogb, p.n_tag, partial)
    if (p.OG == 2 && p.TA == 1 && p.TB == 1) DODA_W(2, 1, 1);
    else if (p.OG == 2 && p.TA == 2 && p.TB == 1) DODA_W(2, 2, 1);
    else if (p.OG == 2 && p.TA == 1 && p.TB == 2) DODA_W(2, 1, 2);
    else if (p.OG == 1 && p.TA == 2 && p.TB == 2) DODA_W(1, 2, 2);
    else return DODA_ERR_UNSUPPORTED;
#undef DODA_W
    int st = doda_check_launch();
    if (st != DODA_OK) return st;
    hipLaunchKernelGGL(wgrad_reduce, dim3(div_up(n_elem, 256)), dim3(256), 0, s, partial, p.R,
                       n_elem, dw);
    return doda_check_launch();
}
