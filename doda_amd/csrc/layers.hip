// Per-layer backend of the U-Net op list (ABI 11: doda_layers_run; ABI 12: every level — GEMM ops carry their table's tilebook).
//
// The deep levels of DODA's U-Net (reference model/unet_block.py:55-100 UBlock: blocks -> strided conv -> UBlock -> inverse conv ->
// concatenation -> blocks_tail, ResidualBlocks of model/unet_block.py:9-37 inside) are described by the caller as a list of
// doda_cx_op (include/doda_hip.h).  Every op is a WHOLE-CHIP launch of the kernels the module-by-module path uses anyway (doda_spconv_gather_ex, the BatchNorm sweeps over fp64 totals), issued back to back
// from C++ with nothing in between — and a BatchNorm whose consumer is the next convolution of the list is folded into that
// convolution's gather (doda_conv_prologue):
//     BNFWD ; GEMM(x = its output)                 -> one launch   (forward:  BatchNorm1d -> ReLU -> conv)
//     BNBWD ; GEMM(x = its output)                 -> one launch   (backward: the BatchNorm's input gradient feeds the previous conv's
//                                                                   data gradient; the skip gradient rides along as `add`)
// wherever the BatchNorm's rows are few enough for its own sweep to be a launch-floor kernel (DODA_PRE_FWD_ROWS, default 16384 /
// DODA_PRE_BWD_ROWS, default 0 = the backward BatchNorm keeps its own 3.3 us launch: measured, see g_bwd_rows below).
// The folded and the unfolded form of an op give the same bits (bn_totals.hpp), so the fusion is a schedule, not a numerics change.
// (Round 5 walked the same list inside ONE persistent launch on one XCD: 1/8 of the chip's matrix rate and loads in flight lost to
// the whole-chip kernels at the bench size and, once those were issued from here, at the host floor too — removed in ABI 11.)
#include "common.hpp"
#include "spconv_common.hpp"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

inline long long env_ll(const char *name, long long dflt) {
    const char *e = getenv(name);
    return e && *e ? atoll(e) : dflt;
}

// ---- standalone BatchNorm ops in the general (strided, split-output) form -------------------------------------------------
// y[:, 0:c_split) -> y, y[:, c_split:c) -> y2: the two halves of a concatenation's gradient as two dense tensors.
template <int ESZ, int KIND>
__global__ __launch_bounds__(256) void lay_bn(const void *__restrict__ x_, unsigned x_ld, int c, const PreArgs pre, void *__restrict__ y_,
                                              unsigned y_ld, void *__restrict__ y2_, unsigned y2_ld, int c_split) {
    typedef typename std::conditional<ESZ == 2, unsigned short, float>::type elem;
    __shared__ __attribute__((aligned(16))) float co[PreForm<ESZ, KIND>::NVEC][PRE_MAX_C];
    const PreRaw raw = pre_request<KIND>(pre, c);
    pre_finish<ESZ, KIND>(pre, c, raw, co);
    doda_sync();
    // a thread keeps ONE 16-byte column piece for the whole sweep (the block uses the largest multiple of the pieces per row among
    // its 256 threads): the per-channel coefficients are read from LDS once, no division per piece, four rows in flight per thread
    constexpr int NV = 16 / ESZ, U = 4;
    const int ppr = c / NV, rpb = 256 / ppr;
    const int tr = (int)threadIdx.x / ppr, c0 = ((int)threadIdx.x - tr * ppr) * NV;
    if (tr >= rpb) return;
    PreCo<ESZ, KIND> cv;
    pre_load_co<ESZ, KIND>(co, c0, cv);
    const elem *x = (const elem *)x_;
    const bool left = c0 < c_split;
    elem *const yo = left ? (elem *)y_ + c0 : (elem *)y2_ + (c0 - c_split);
    const size_t yl = left ? y_ld : y2_ld;
    const long long rows = pre.rows, step = (long long)gridDim.x * rpb;
    for (long long r0 = (long long)blockIdx.x * rpb + tr; r0 < rows; r0 += step * U) {
        u32x4 xv[U], uv[U], av[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long r = r0 + k * step < rows ? r0 + k * step : r0;   // (past the end: the first row again, not stored)
            xv[k] = *reinterpret_cast<const u32x4 *>(x + r * x_ld + c0);
            uv[k] = av[k] = xv[k];
            if constexpr (KIND >= 2) uv[k] = *reinterpret_cast<const u32x4 *>((const elem *)pre.aux + r * pre.aux_ld + c0);
            if constexpr (KIND >= 3) av[k] = *reinterpret_cast<const u32x4 *>((const elem *)pre.add + r * pre.add_ld + c0);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long r = r0 + k * step;
            const u32x4 o = pre_piece<ESZ, KIND>(xv[k], uv[k], av[k], cv, pre.relu, ~0u);
            if (r < rows) *reinterpret_cast<u32x4 *>(yo + r * yl) = o;
        }
    }
}

// (sum x, sum x^2) of a [rows, c] matrix into fp64 totals (layout: spconv_common.hpp stats_emit).  One thread per (row lane,
// 4-channel group); fp32 partial sums over at most a few hundred rows per thread, fp64 atomics per workgroup.
template <int ESZ>
__global__ __launch_bounds__(256) void lay_stats(const void *__restrict__ x_, unsigned x_ld, int rows, int c, double *__restrict__ tot) {
    typedef typename std::conditional<ESZ == 2, unsigned short, float>::type elem;
    const elem *x = (const elem *)x_;
    const int nf = c / 4, rpb = 256 / nf > 0 ? 256 / nf : 1;
    const int f = threadIdx.x % nf, rl = threadIdx.x / nf;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    __shared__ float red[2][256][4];
    if (rl < rpb && f < nf) {
        for (long long r = (long long)blockIdx.x * rpb + rl; r < rows; r += (long long)gridDim.x * rpb) {
            f32x4 v;
            if constexpr (ESZ == 2) {
                const u32x2 p = *reinterpret_cast<const u32x2 *>(x + r * x_ld + f * 4);
                v = (f32x4){__uint_as_float(p[0] << 16), __uint_as_float(p[0] & 0xffff0000u), __uint_as_float(p[1] << 16),
                            __uint_as_float(p[1] & 0xffff0000u)};
            } else {
                v = *reinterpret_cast<const f32x4 *>(x + r * x_ld + f * 4);
            }
            s1 += v;
            s2 += v * v;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[0][threadIdx.x][q] = s1[q]; red[1][threadIdx.x][q] = s2[q]; }
    doda_sync();
    if (rl == 0 && f < nf) {
        for (int k = 1; k < rpb; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) { s1[q] += red[0][k * nf + f][q]; s2[q] += red[1][k * nf + f][q]; }
        const size_t slot = (size_t)(blockIdx.x & (DODA_STATS_SLOTS - 1));
        double *t1 = tot + ((slot * 2 + 0) * (size_t)nf + (size_t)f) * 16, *t2 = tot + ((slot * 2 + 1) * (size_t)nf + (size_t)f) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) { unsafeAtomicAdd(t1 + q, (double)s1[q]); unsafeAtomicAdd(t2 + q, (double)s2[q]); }
    }
}

inline bool chan_ok(int c, int esz) { return c > 0 && c <= PRE_MAX_C && c % (16 / esz) == 0; }
inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// PreArgs of a BatchNorm op of the list (BNFWD: kind 1; BNBWD: kind 2 / 3)
bool pre_of(const doda_cx_op &o, int esz, PreArgs *p) {
    *p = PreArgs{};
    if (o.kind == DODA_CX_BNFWD) {
        const bool training = (o.flags & DODA_CX_F_TRAINING) != 0;
        p->kind = 1;
        p->relu = (o.flags & DODA_CX_F_RELU) ? 1 : 0;
        p->rows = o.rows;
        if (training) {
            const int split = o.c_split > 0 && o.c_split < o.c_in ? o.c_split : o.c_in;
            if (!o.stats || (split < o.c_in && !o.stats_b) || !o.mean || !o.invstd || split % 4) return false;
            p->tot.ta = (const double *)o.stats;
            p->tot.tb = split < o.c_in ? (const double *)o.stats_b : nullptr;
            p->tot.ca = split;
            p->tot.rm = o.running_mean; p->tot.rv = o.running_var; p->tot.nbt = (long long *)o.nbt;
            p->tot.out_a = o.mean; p->tot.out_b = o.invstd;
        } else {
            if (!o.running_mean || !o.running_var) return false;
            p->tot.rm = o.running_mean; p->tot.rv = o.running_var;
            p->tot.ca = o.c_in;
        }
        p->tot.m = o.rows; p->tot.eps = o.eps; p->tot.momentum = o.momentum;
        p->gamma = o.gamma; p->beta = o.beta;
        p->mean = o.mean; p->invstd = o.invstd;
        p->side = o.y; p->side_ld = (unsigned)o.y_ld;
        return o.gamma && o.beta && o.y;
    }
    if (o.kind == DODA_CX_BNBWD) {
        p->kind = o.res ? 3 : 2;
        p->relu = (o.flags & DODA_CX_F_RELU) ? 1 : 0;
        p->rows = o.rows;
        p->tot.ta = (const double *)o.stats; p->tot.ca = o.c_in; p->tot.m = o.rows;
        p->tot.out_a = o.dgamma; p->tot.out_b = o.dbeta; p->tot.accum = (o.flags & DODA_CX_F_ACCUM) ? 1 : 0;
        p->gamma = o.gamma; p->beta = o.beta; p->mean = o.mean; p->invstd = o.invstd;
        p->side = o.y; p->side_ld = (unsigned)o.y_ld;
        p->aux = o.aux; p->aux_ld = (unsigned)o.aux_ld;
        p->add = o.res; p->add_ld = (unsigned)o.res_ld;
        return o.stats && o.gamma && o.beta && o.mean && o.invstd && o.aux && o.y && o.dgamma && o.dbeta;
    }
    return false;
}

template <int KIND>
int launch_bn(const doda_cx_op &o, int esz, const PreArgs &p, hipStream_t s) {
    const int c = o.c_in;
    const int ppr = c / (16 / esz), rpb = 256 / ppr;            // (c <= PRE_MAX_C = 256: ppr <= 64)
    long long grid = ((long long)o.rows + rpb - 1) / rpb;       // one row per thread, then four
    static const long long cap = env_ll("DODA_LAY_BN_GRID", 2048);
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const int split = (o.kind == DODA_CX_BNBWD && o.c_split > 0 && o.c_split < c) ? o.c_split : c;
    if (esz == 2)
        hipLaunchKernelGGL((lay_bn<2, KIND>), dim3((unsigned)grid), dim3(256), 0, s, o.x, (unsigned)o.x_ld, c, p, o.y, (unsigned)o.y_ld, o.y2,
                           (unsigned)o.y2_ld, split);
    else
        hipLaunchKernelGGL((lay_bn<4, KIND>), dim3((unsigned)grid), dim3(256), 0, s, o.x, (unsigned)o.x_ld, c, p, o.y, (unsigned)o.y_ld, o.y2,
                           (unsigned)o.y2_ld, split);
    return doda_check_launch();
}

// rows from which a dense BatchNorm op takes the register-resident sweeps of bn.hip instead of lay_bn
inline long long tuned_rows(int esz) {
    static const long long bf = env_ll("DODA_LAY_TUNED_ROWS", 32768);
    return esz == 4 ? 4096 : bf;
}

int run_bn(const doda_cx_op &o, int esz, hipStream_t s) {
    PreArgs p;
    if (!pre_of(o, esz, &p)) return DODA_ERR_INVALID;
    const int va = 16 / esz;
    if (!chan_ok(o.c_in, esz) || !o.x || o.x_ld % va || o.y_ld % va || !al16(o.x) || !al16(o.y)) return DODA_ERR_UNSUPPORTED;
    if (o.kind == DODA_CX_BNFWD) {
        if (o.y_ld < o.c_in || o.x_ld < o.c_in) return DODA_ERR_INVALID;
        // dense training-mode sweeps of many rows take the tuned kernels of bn.hip (registers hold the channel vectors).  fp32: the
        // same operation order as lay_bn / the folded gather, so from 4096 rows; bf16: bn.hip rounds in a different order than the
        // fused-multiply-add form of pre_piece, so only above any fold limit, where "folded == unfolded" has nothing to compare
        // (ABI 12, bf16: above the rows any fold limit reaches — the finest levels, where the sweep is an HBM-bound kernel)
        if ((o.flags & DODA_CX_F_TRAINING) && o.x_ld == o.c_in && o.y_ld == o.c_in && o.rows >= tuned_rows(esz))
            return doda_bn_relu_fwd_totals(o.x, o.rows, o.c_in, esz, p.tot.ta, p.tot.tb, p.tot.ca, o.eps, o.momentum, o.gamma, o.beta,
                                           o.running_mean, o.running_var, o.nbt, p.relu, o.y, o.mean, o.invstd, (doda_stream_t)s);
        return launch_bn<1>(o, esz, p, s);
    }
    const int split = (o.c_split > 0 && o.c_split < o.c_in) ? o.c_split : o.c_in;
    if (split % va || (split < o.c_in && (!o.y2 || o.y2_ld % va || !al16(o.y2))) || o.aux_ld % va || !al16(o.aux) ||
        (o.res && (o.res_ld % va || !al16(o.res))))
        return DODA_ERR_UNSUPPORTED;
    if (split == o.c_in && o.x_ld == o.c_in && o.y_ld == o.c_in && o.aux_ld == o.c_in && !(o.flags & DODA_CX_F_ACCUM) && o.rows >= tuned_rows(esz))
        return doda_bn_relu_bwd_totals(o.aux, o.x, o.rows, o.c_in, esz, p.tot.ta, o.mean, o.invstd, o.gamma, o.beta, p.relu, o.res,
                                       o.res ? o.res_ld : 0, o.y, o.dgamma, o.dbeta, (doda_stream_t)s);
    return p.kind == 3 ? launch_bn<3>(o, esz, p, s) : launch_bn<2>(o, esz, p, s);
}

int run_stats(const doda_cx_op &o, int esz, hipStream_t s) {
    if (!o.x || !o.stats || o.c_in % 4 || o.c_in <= 0 || o.c_in > 1024 || o.x_ld % 4 || o.x_ld < o.c_in) return DODA_ERR_INVALID;
    const int nf = o.c_in / 4, rpb = 256 / nf > 0 ? 256 / nf : 1;
    if (nf > 256) return DODA_ERR_UNSUPPORTED;
    long long grid = ((long long)o.rows + (long long)rpb * 8 - 1) / ((long long)rpb * 8);   // ~8 rows per thread
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    if (esz == 2) hipLaunchKernelGGL((lay_stats<2>), dim3((unsigned)grid), dim3(256), 0, s, o.x, (unsigned)o.x_ld, o.rows, o.c_in, (double *)o.stats);
    else hipLaunchKernelGGL((lay_stats<4>), dim3((unsigned)grid), dim3(256), 0, s, o.x, (unsigned)o.x_ld, o.rows, o.c_in, (double *)o.stats);
    return doda_check_launch();
}

// the convolution of the list, optionally with a BatchNorm op folded into its gather
int run_gemm(const doda_cx_op &o, int esz, const doda_cx_op *bn, hipStream_t s) {
    if (!o.x || !o.w || !o.y || !o.tbl) return DODA_ERR_INVALID;
    doda_conv_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    int32_t stats_rows = 0;
    ep.residual = o.res;
    ep.residual_ld = o.res ? o.res_ld : 0;
    ep.x_ld = o.x_ld;
    ep.y_ld = o.y_ld;
    ep.tilebook = o.tilebook;
    ep.tilebook_rows = o.tilebook ? o.rows : 0;
    if (o.stats) {
        ep.stats = (float *)o.stats;             // (non-NULL selects the statistics epilogue; the sums go to the totals)
        ep.stats_totals = (double *)o.stats;
        ep.stats_rows_h = &stats_rows;
        if (o.aux) {
            ep.bn_x = o.aux; ep.bn_x_ld = o.aux_ld;
            ep.bn_mean = o.mean; ep.bn_invstd = o.invstd; ep.bn_gamma = o.gamma; ep.bn_beta = o.beta;
            ep.bn_relu = (o.flags & DODA_CX_F_RELU) ? 1 : 0;
        }
    }
    doda_conv_prologue q;
    const void *x = o.x;
    if (bn) {
        memset(&q, 0, sizeof(q));
        const bool fwd = bn->kind == DODA_CX_BNFWD;
        q.kind = fwd ? 1 : (bn->res ? 3 : 2);
        q.relu = (bn->flags & DODA_CX_F_RELU) ? 1 : 0;
        q.rows = bn->rows;
        q.eps = bn->eps; q.momentum = bn->momentum;
        q.gamma = bn->gamma; q.beta = bn->beta;
        q.mean = bn->mean; q.invstd = bn->invstd;
        q.side = bn->y; q.side_ld = bn->y_ld;
        if (fwd) {
            if (bn->flags & DODA_CX_F_TRAINING) {
                const int split = bn->c_split > 0 && bn->c_split < bn->c_in ? bn->c_split : bn->c_in;
                q.totals = (const double *)bn->stats;
                q.totals_b = split < bn->c_in ? (const double *)bn->stats_b : nullptr;
                q.c_a = split;
                q.num_batches_tracked = bn->nbt;
            }
            q.running_mean = bn->running_mean; q.running_var = bn->running_var;
        } else {
            q.totals = (const double *)bn->stats;
            q.aux = bn->aux; q.aux_ld = bn->aux_ld;
            q.add = bn->res; q.add_ld = bn->res ? bn->res_ld : 0;
            q.dgamma = bn->dgamma; q.dbeta = bn->dbeta;
            q.accumulate = (bn->flags & DODA_CX_F_ACCUM) ? 1 : 0;
        }
        ep.prologue = &q;
        x = bn->x;                 // the conv gathers the BatchNorm's INPUT rows
        ep.x_ld = bn->x_ld;
    }
    return doda_spconv_gather_ex(x, o.rows_in, o.c_in, esz, (const float *)o.w, o.c_out, o.tbl, o.tbl_ld, o.K, o.rows, o.y, 0, 0x100,
                                 nullptr, 0, &ep, (doda_stream_t)s);
}

// may ops[i] (a BatchNorm op) ride in the gather of ops[i + 1]?
bool foldable(const doda_cx_op &b, const doda_cx_op &g, int esz, long long max_rows) {
    if (g.kind != DODA_CX_GEMM || g.x != b.y || g.x_ld != b.y_ld || g.c_in != b.c_in || g.rows_in != b.rows || b.rows > max_rows ||
        b.y_ld != b.c_in)   // (the side output is the weight gradient's dense operand)
        return false;
    if (b.kind == DODA_CX_BNBWD && b.c_split > 0 && b.c_split < b.c_in) return false;   // two outputs: its own launch
    if (esz == 2 && b.c_in < 32) return false;
    return chan_ok(b.c_in, esz);
}

// Defaults from measurements on MI355X (tools/prebench_prof.sh, tools/layers_ab.py): at 1900 rows x 80 channels a standalone
// totals sweep (lay_bn) is a 3.3 us kernel + ~2 us boundary; folding costs the conv +5.3 us forward (break-even on the GPU, one launch
// less on the host: 4.88 -> 4.81 ms per bench step) and +8.3 us backward (two gathered operands, ~60 vector instructions per piece:
// 4.81 -> 4.89 ms) — so the forward folds, the backward does not (DODA_PRE_BWD_ROWS=4096 to fold it on a host-bound box).
long long g_fwd_rows = env_ll("DODA_PRE_FWD_ROWS", 16384), g_bwd_rows = env_ll("DODA_PRE_BWD_ROWS", 0);

}  // namespace

namespace doda_layers {
long long fwd_rows() { return g_fwd_rows; }
long long bwd_rows() { return g_bwd_rows; }
void set_fwd_rows(long long v) { g_fwd_rows = v < 0 ? 0 : v; }
void set_bwd_rows(long long v) { g_bwd_rows = v < 0 ? 0 : v; }
}  // namespace doda_layers

extern "C" int doda_layers_run(const doda_cx_op *ops_h, int32_t n_ops, int32_t elem_bytes, int32_t *n_launches_h, doda_stream_t stream) {
    if (n_launches_h) *n_launches_h = 0;
    if (n_ops == 0) return DODA_OK;
    if (n_ops < 0 || !ops_h || (elem_bytes != 2 && elem_bytes != 4)) return DODA_ERR_INVALID;
    const long long fwd_rows = g_fwd_rows, bwd_rows = g_bwd_rows;
    hipStream_t s = as_stream(stream);
    int launches = 0;
    for (int k = 0; k < n_ops; ++k) {
        const doda_cx_op &o = ops_h[k];
        if (o.rows < 0 || o.n_part != 0) return DODA_ERR_INVALID;   // (n_part != 0: a list built for the executor's partial rows)
        if (o.rows == 0) continue;
        int st = DODA_OK;
        switch (o.kind) {
        case DODA_CX_GEMM:
            st = run_gemm(o, elem_bytes, nullptr, s);
            break;
        case DODA_CX_BNFWD:
        case DODA_CX_BNBWD: {
            const long long lim = o.kind == DODA_CX_BNFWD ? fwd_rows : bwd_rows;
            if (k + 1 < n_ops && foldable(o, ops_h[k + 1], elem_bytes, lim)) {
                st = run_gemm(ops_h[k + 1], elem_bytes, &o, s);
                if (st == DODA_OK) { ++k; break; }
                if (st != DODA_ERR_UNSUPPORTED) break;      // (UNSUPPORTED: a shape the folded kernels do not take — two launches)
            }
            st = run_bn(o, elem_bytes, s);
            break;
        }
        case DODA_CX_STATS:
            st = run_stats(o, elem_bytes, s);
            break;
        default:
            return DODA_ERR_INVALID;
        }
        if (st != DODA_OK) return st;
        ++launches;
    }
    if (n_launches_h) *n_launches_h = launches;
    return DODA_OK;
}
