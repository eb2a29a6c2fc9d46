// Sparse-convolution weight gradient over PAIR LISTS:
//     dw[o][ci][co] (+)= sum_{p < n_o} a[in_o[p]][ci] * b[out_o[p]][co]
// (spconv v1.2 indice_conv_backward's per-offset `Xg^T . dYg` GEMMs over indice_pairs /
// indice_pair_num; reference call sites model/unet_block.py:26,29,48,70,78.)
//
// Why a second formulation next to spconv_wgrad.hip (dense gather table + LDS transpose):
//   * the dense table walks K * M (offset, row) slots of which ~36 % hold a pair at level 1 (and 1/8
//     for the strided convolutions); the pair lists hold the P present pairs only: 2.8x fewer gathers,
//     MFMAs and bytes, and the table (4*K*M bytes) is not read at all;
//   * the contraction runs over ROWS, so both MFMA operands need rows as their k dimension while
//     memory holds channels contiguously.  The dense kernel goes through LDS (16-byte stores +
//     ds_read_b64_tr_b16; PMC: bank conflicts 70 % of LDS-active cycles, the bound of that kernel).
//     Here the transposition is done by the matrix core itself: a gathered 16-byte row slice is the
//     natural A operand of v_mfma_f32_16x16x32_bf16 (lane = pair, registers = channels); multiplied
//     by a ONE-HOT B operand it comes back in D layout with lane = channel, registers = pairs —
//     exactly the k-order the contraction wants.  Products with 1.0 and sums with zeros are exact,
//     and the fp32 -> bf16 repack keeps the upper 16 bits of values that are bf16 already.
//     No LDS, no barriers in the loop, every wave independent.
// Per 32 pairs and 16x16 channel tile: 2 index reads (shared by two steps), 2 row gathers (16 B per
// lane), 4 transposing MFMAs, 1 contraction MFMA.  MFMA is still far from a bound.
// Deterministic: per-chunk partials, fixed-order reduce, no float atomics.
#include "wgrad_pairs.hpp"
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WAVE_PAIRS = 512;            // pairs per wave: 8 double-steps of 64
constexpr int BLK_PAIRS = 4 * WAVE_PAIRS;  // pairs per block (one offset, one chunk)
constexpr unsigned OOB = 0x80000000u;      // absent pair: beyond any buffer (operands are < 1 GB)
constexpr unsigned CH_OOB = 0x40000000u;   // channel block past the channel count (OOB + CH_OOB does not wrap)

struct PJob {            // one layer inside a kernel-variant group
    const void *a, *b;
    const int32_t *pin, *pout, *pnum;
    float *partial;      // [K][n_chunk][ca][cb]
    unsigned a_bytes, b_bytes;
    int ca, cb, ld, K, n_rows, n_chunk, n_tag, n_tbg, blk_end, pad;
};
struct RJob {            // dw[o][q] (+)= sum_{c < chunks(o)} partial[o][c][q]
    const float4 *partial;
    float4 *dw;
    const int32_t *pnum;
    int n_quad, n_chunk, K, ld, accumulate, blk_end;
};

template <class J>
__device__ __forceinline__ int find_job(const J *jobs, int n_jobs, int blk) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (blk < jobs[mid].blk_end) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ bf16x8 pack_hi16(const f32x4 &d0, const f32x4 &d1) {
    // the values are bf16-exact: keep the upper halves.  k-slot q of the lane: q < 4 -> d0[q], else d1[q-4]
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(__float_as_uint(d0[1]), __float_as_uint(d0[0]), 0x07060302u);
    r[1] = __builtin_amdgcn_perm(__float_as_uint(d0[3]), __float_as_uint(d0[2]), 0x07060302u);
    r[2] = __builtin_amdgcn_perm(__float_as_uint(d1[1]), __float_as_uint(d1[0]), 0x07060302u);
    r[3] = __builtin_amdgcn_perm(__float_as_uint(d1[3]), __float_as_uint(d1[2]), 0x07060302u);
    return __builtin_bit_cast(bf16x8, r);
}

// TA x TB 16-channel blocks of (a, b) per wave.  Block = (offset o, chunk of BLK_PAIRS pairs,
// channel tile); its four waves take WAVE_PAIRS pairs each and add up through LDS at the end.
template <int TA, int TB>
__global__ __launch_bounds__(256) void wgrad_pairs_kernel(const PJob *__restrict__ jobs, int n_jobs) {
    const int jn = find_job(jobs, n_jobs, (int)blockIdx.x);
    const PJob d = jobs[jn];
    int lb = (int)blockIdx.x - (jn == 0 ? 0 : jobs[jn - 1].blk_end);
    const int tbg = lb % d.n_tbg; lb /= d.n_tbg;
    const int tag = lb % d.n_tag; lb /= d.n_tag;
    const int chunk = lb % d.n_chunk;
    const int o = lb / d.n_chunk;
    // no counts: every list is full (identity list of a 1x1 conv).  readfirstlane: the value sizes a
    // buffer descriptor, which must be provably wave-uniform (else hipcc wraps each load in a waterfall loop)
    const int n_o = __builtin_amdgcn_readfirstlane(d.pnum ? d.pnum[o] : d.ld);
    const int c0 = chunk * BLK_PAIRS;
    if (c0 >= n_o) return;   // block-uniform: the list of this offset ends before the chunk

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int p_begin = c0 + wid * WAVE_PAIRS;
    const int p_end = (p_begin + WAVE_PAIRS < n_o) ? p_begin + WAVE_PAIRS : n_o;

    // one-hot B operands: P[G][k = (g, q)][j = i] = 1 iff the lane group carries pair group G
    // (g >> 1 == G), the half-row of channel j (g & 1 == j >> 3) and q == j & 7
    bf16x8 P[2];
#pragma unroll
    for (int G = 0; G < 2; ++G) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const bool mine = (g >> 1) == G && (i >> 3) == (g & 1);
        const int q = i & 7;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            v[w] = (mine && (q >> 1) == w) ? ((q & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
        P[G] = __builtin_bit_cast(bf16x8, v);
    }

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)d.a, 0, d.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)d.b, 0, d.b_bytes, 0x00020000);
    const int32_t *pi = d.pin + (long long)o * d.ld, *po = d.pout + (long long)o * d.ld;
    const unsigned rb_a = (unsigned)d.ca * 2u, rb_b = (unsigned)d.cb * 2u;
    const unsigned half = (unsigned)(g & 1) * 16u;
    // channel blocks of this tile; a block past the channel count reads zeros
    unsigned ch_a[TA], ch_b[TB];
#pragma unroll
    for (int xa = 0; xa < TA; ++xa) {
        const int blk = tag * TA + xa;
        ch_a[xa] = blk * 16 < d.ca ? (unsigned)blk * 32u + half : CH_OOB;
    }
#pragma unroll
    for (int yb = 0; yb < TB; ++yb) {
        const int blk = tbg * TB + yb;
        ch_b[yb] = blk * 16 < d.cb ? (unsigned)blk * 32u + half : CH_OOB;
    }

    f32x4 acc[TA][TB];
#pragma unroll
    for (int xa = 0; xa < TA; ++xa)
#pragma unroll
        for (int yb = 0; yb < TB; ++yb) acc[xa][yb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // index registers of one double-step: lane l holds the pair p + l.  Raw buffer loads over this
    // offset's lists (num_records = n_o entries): a lane past the end reads 0 and is masked when the
    // rows are requested.  No branch anywhere in the loop: with control flow inside it hipcc falls
    // back to s_waitcnt vmcnt(0) in front of every MFMA group and nothing stays in flight.
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void *)pi, 0, (unsigned)n_o * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void *)po, 0, (unsigned)n_o * 4u, 0x00020000);
    auto load_idx = [&](int p, int &in_l, int &out_l) {
        const unsigned voff = (unsigned)(p + lane) * 4u;
        in_l = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_i, voff, 0, 0);
        out_l = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_o, voff, 0, 0);
    };
    // row slices of sub-step h (32 pairs) of the double-step whose indices are (in_l, out_l)
    auto load_rows = [&](int p, int h, int in_l, int out_l, u32x4 (&xr)[TA], u32x4 (&yr)[TB]) {
        const int src = 32 * h + 16 * (g >> 1) + i;
        const int in_s = __shfl(in_l, src, 64), out_s = __shfl(out_l, src, 64);
        const bool ok = p + src < p_end;
        const unsigned va = ok ? (unsigned)in_s * rb_a : OOB, vb = ok ? (unsigned)out_s * rb_b : OOB;
#pragma unroll
        for (int xa = 0; xa < TA; ++xa) xr[xa] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, va + ch_a[xa], 0, 0);
#pragma unroll
        for (int yb = 0; yb < TB; ++yb) yr[yb] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vb + ch_b[yb], 0, 0);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const u32x4 (&xr)[TA], const u32x4 (&yr)[TB]) {
        bf16x8 at[TA], bt[TB];
#pragma unroll
        for (int xa = 0; xa < TA; ++xa) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, xr[xa]);
            const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, P[0], zero, 0, 0, 0);
            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, P[1], zero, 0, 0, 0);
            at[xa] = pack_hi16(d0, d1);
        }
#pragma unroll
        for (int yb = 0; yb < TB; ++yb) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, yr[yb]);
            const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, P[0], zero, 0, 0, 0);
            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, P[1], zero, 0, 0, 0);
            bt[yb] = pack_hi16(d0, d1);
        }
#pragma unroll
        for (int xa = 0; xa < TA; ++xa)
#pragma unroll
            for (int yb = 0; yb < TB; ++yb)
                acc[xa][yb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[xa], bt[yb], acc[xa][yb], 0, 0, 0);
    };

    if (p_begin < p_end) {
        // Software pipeline over double-steps of 64 pairs, program order per iteration k:
        //   indices(k+2)  ->  rows(k+1) (needs indices(k+1), issued one iteration ago)  ->  MFMAs(k).
        // vmcnt retires in order, so requesting the indices FIRST means neither wait drains the queue:
        // rows(k+1) and indices(k+2) stay in flight under the MFMAs of k.  The scheduling barriers keep
        // hipcc from hoisting the MFMAs above the loads.  Loads past p_end are issued all the same
        // (index 0 / out-of-range row offset -> zeros).
        // Two iterations are written out with the buffer roles swapped: a register copy of a buffer
        // whose loads are in flight would wait for them.
        int in_c, out_c, in_n, out_n, in_f, out_f;
        u32x4 xa0[TA], ya0[TB], xa1[TA], ya1[TB], xb0[TA], yb0[TB], xb1[TA], yb1[TB];
        load_idx(p_begin, in_c, out_c);
        load_idx(p_begin + 64, in_n, out_n);
        load_rows(p_begin, 0, in_c, out_c, xa0, ya0);
        load_rows(p_begin, 1, in_c, out_c, xa1, ya1);
        for (int p = p_begin; p < p_end; p += 128) {
            load_idx(p + 128, in_f, out_f);
            __builtin_amdgcn_sched_barrier(0);
            load_rows(p + 64, 0, in_n, out_n, xb0, yb0);
            load_rows(p + 64, 1, in_n, out_n, xb1, yb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(xa0, ya0);
            compute(xa1, ya1);
            __builtin_amdgcn_sched_barrier(0);
            load_idx(p + 192, in_n, out_n);
            __builtin_amdgcn_sched_barrier(0);
            load_rows(p + 128, 0, in_f, out_f, xa0, ya0);
            load_rows(p + 128, 1, in_f, out_f, xa1, ya1);
            __builtin_amdgcn_sched_barrier(0);
            compute(xb0, yb0);
            compute(xb1, yb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the block's four waves add up (fixed order), wave 0 writes the chunk's partial ----
    __shared__ f32x4 part[3][TA][TB][64];
    if (wid > 0) {
#pragma unroll
        for (int xa = 0; xa < TA; ++xa)
#pragma unroll
            for (int yb = 0; yb < TB; ++yb) part[wid - 1][xa][yb][lane] = acc[xa][yb];
    }
    __syncthreads();
    if (wid > 0) return;
    float *out = d.partial + ((long long)o * d.n_chunk + chunk) * d.ca * d.cb;
#pragma unroll
    for (int xa = 0; xa < TA; ++xa)
#pragma unroll
        for (int yb = 0; yb < TB; ++yb) {
            f32x4 t = acc[xa][yb];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4 v = part[w][xa][yb][lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] += v[r];
            }
            // D[i = ci][j = co]: lane (co = lane & 15, g) holds ci = 4g + r
            const int co = (tbg * TB + yb) * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = (tag * TA + xa) * 16 + 4 * g + r;
                if (ci < d.ca && co < d.cb) out[(long long)ci * d.cb + co] = t[r];
            }
        }
}

// dw[o][q] (+)= sum over the chunks the offset's list reaches.  16 quads x 16 chunk lanes per block,
// lane r sums chunks r, r+16, ... and the lane sums are added in ascending r: fixed order.
__global__ __launch_bounds__(256) void wgrad_pairs_reduce(const RJob *__restrict__ jobs, int n_jobs) {
    __shared__ float4 part[16][16];
    const int jn = find_job(jobs, n_jobs, (int)blockIdx.x);
    const RJob d = jobs[jn];
    int lb = (int)blockIdx.x - (jn == 0 ? 0 : jobs[jn - 1].blk_end);
    const int qb = (d.n_quad + 15) / 16;
    const int o = lb / qb;
    const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int q = (lb - o * qb) * 16 + el;
    const int n_o = d.pnum ? d.pnum[o] : d.ld;
    const int nch = (n_o + BLK_PAIRS - 1) / BLK_PAIRS;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d.n_quad)
        for (int c = rl; c < nch; c += 16) {
            const float4 v = d.partial[((long long)o * d.n_chunk + c) * d.n_quad + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[rl][el] = s;
    __syncthreads();
    if (rl == 0 && q < d.n_quad) {
        float4 t = part[0][el];
#pragma unroll 4
        for (int r = 1; r < 16; ++r) {
            const float4 v = part[r][el];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        float4 *dst = d.dw + (long long)o * d.n_quad + q;
        if (d.accumulate) {
            const float4 old = *dst;
            t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
        }
        *dst = t;
    }
}

struct Geo { int ta, tb, n_tag, n_tbg, n_chunk; };

Geo make_geo(const doda_wgrad_job &j) {
    Geo g;
    const int na = j.ca / 16, nb = j.cb / 16;
    g.ta = na >= 2 ? 2 : 1;
    g.tb = nb >= 2 ? 2 : 1;
    g.n_tag = div_up(na, g.ta);
    g.n_tbg = div_up(nb, g.tb);
    g.n_chunk = div_up(j.pair_ld > 0 ? j.pair_ld : 1, BLK_PAIRS);
    return g;
}

template <int TA, int TB>
void launch_variant(int blocks, const PJob *jobs_dev, int n, hipStream_t s) {
    hipLaunchKernelGGL((wgrad_pairs_kernel<TA, TB>), dim3(blocks), dim3(256), 0, s, jobs_dev, n);
}

}  // namespace

namespace doda_pairs {

bool eligible(const doda_wgrad_job &j) {
    if (j.elem_bytes != 2 || j.ca <= 0 || j.cb <= 0 || (j.ca % 16) || (j.cb % 16) || j.K <= 0 || j.n_rows <= 0)
        return false;
    if (!j.a || !j.b || !j.dw) return false;
    if (!j.pair_in || !j.pair_out || j.pair_ld <= 0 || j.n_a <= 0) return false;
    const long long n_a = j.n_a;
    if ((unsigned long long)n_a * j.ca * 2ull >= 0x3fffffffull) return false;
    if ((unsigned long long)j.n_rows * j.cb * 2ull >= 0x3fffffffull) return false;
    if (((uintptr_t)j.a % 16) || ((uintptr_t)j.b % 16) || ((uintptr_t)j.dw % 16)) return false;
    const Geo g = make_geo(j);
    if ((long long)j.K * g.n_chunk * g.n_tag * g.n_tbg > 0x3fffffff) return false;
    return true;
}

size_t partial_bytes(const doda_wgrad_job &j) {
    const Geo g = make_geo(j);
    return align_up((size_t)j.K * g.n_chunk * j.ca * j.cb * 4, 256);
}

size_t desc_bytes_per_job() { return sizeof(PJob) + sizeof(RJob); }

int prepare(const doda_wgrad_job *jobs, const int *which, int n, char *ws_base, size_t *ws_off, Prepared *out) {
    out->desc.clear();
    out->groups.clear();
    std::vector<PJob> pj;
    std::vector<RJob> rj;
    std::vector<size_t> offs(n);
    for (int k = 0; k < n; ++k) {
        offs[k] = *ws_off;
        *ws_off += partial_bytes(jobs[which[k]]);
    }
    for (int ta = 1; ta <= 2; ++ta)
        for (int tb = 1; tb <= 2; ++tb) {
            Prepared::Group grp{ta, tb, (int)pj.size(), 0, 0};
            for (int k = 0; k < n; ++k) {
                const doda_wgrad_job &j = jobs[which[k]];
                const Geo g = make_geo(j);
                if (g.ta != ta || g.tb != tb) continue;
                PJob d;
                memset(&d, 0, sizeof(d));
                d.a = j.a; d.b = j.b;
                d.pin = j.pair_in; d.pout = j.pair_out; d.pnum = j.pair_num;
                d.partial = (float *)(ws_base + offs[k]);
                d.a_bytes = (unsigned)((size_t)j.n_a * j.ca * 2);
                d.b_bytes = (unsigned)((size_t)j.n_rows * j.cb * 2);
                d.ca = j.ca; d.cb = j.cb; d.ld = j.pair_ld; d.K = j.K; d.n_rows = j.n_rows;
                d.n_chunk = g.n_chunk; d.n_tag = g.n_tag; d.n_tbg = g.n_tbg;
                grp.blocks += j.K * g.n_chunk * g.n_tag * g.n_tbg;
                d.blk_end = grp.blocks;
                pj.push_back(d);
                ++grp.count;
            }
            if (grp.count) out->groups.push_back(grp);
        }
    int r_blocks = 0;
    for (int k = 0; k < n; ++k) {
        const doda_wgrad_job &j = jobs[which[k]];
        const Geo g = make_geo(j);
        RJob d;
        memset(&d, 0, sizeof(d));
        d.partial = (const float4 *)(ws_base + offs[k]);
        d.dw = (float4 *)j.dw;
        d.pnum = j.pair_num;
        d.n_quad = j.ca * j.cb / 4;
        d.n_chunk = g.n_chunk; d.K = j.K; d.ld = j.pair_ld;
        d.accumulate = (j.flags & DODA_WGRAD_ACCUMULATE) ? 1 : 0;
        r_blocks += j.K * div_up(d.n_quad, 16);
        d.blk_end = r_blocks;
        rj.push_back(d);
    }
    out->reduce_off = pj.size() * sizeof(PJob);
    out->n_reduce = (int)rj.size();
    out->reduce_blocks = r_blocks;
    out->desc.resize(pj.size() * sizeof(PJob) + rj.size() * sizeof(RJob));
    if (!pj.empty()) memcpy(out->desc.data(), pj.data(), pj.size() * sizeof(PJob));
    if (!rj.empty()) memcpy(out->desc.data() + out->reduce_off, rj.data(), rj.size() * sizeof(RJob));
    return DODA_OK;
}

int launch(const Prepared &p, const void *desc_dev, hipStream_t s) {
    const PJob *pj = (const PJob *)desc_dev;
    for (const Prepared::Group &g : p.groups) {
        if (g.ta == 1 && g.tb == 1) launch_variant<1, 1>(g.blocks, pj + g.first, g.count, s);
        else if (g.ta == 2 && g.tb == 1) launch_variant<2, 1>(g.blocks, pj + g.first, g.count, s);
        else if (g.ta == 1 && g.tb == 2) launch_variant<1, 2>(g.blocks, pj + g.first, g.count, s);
        else launch_variant<2, 2>(g.blocks, pj + g.first, g.count, s);
    }
    int st = doda_check_launch();
    if (st != DODA_OK) return st;
    if (p.n_reduce > 0) {
        hipLaunchKernelGGL(wgrad_pairs_reduce, dim3(p.reduce_blocks), dim3(256), 0, s,
                           (const RJob *)((const char *)desc_dev + p.reduce_off), p.n_reduce);
        st = doda_check_launch();
    }
    return st;
}

}  // namespace doda_pairs
