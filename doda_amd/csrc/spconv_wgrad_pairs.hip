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

constexpr int SEG_TILE = 256;              // rows per tile of the export's segment prefix (rulebook.hip PAIR_TILE)
constexpr int RANGE_TILES = 8;             // a block's row range: 2048 rows of the lists' `in` side
constexpr int RANGE_ROWS = SEG_TILE * RANGE_TILES;
constexpr unsigned OOB = 0x80000000u;      // absent pair: beyond any buffer (operands are < 1 GB)
constexpr unsigned CH_OOB = 0x40000000u;   // channel block past the channel count (OOB + CH_OOB does not wrap)
constexpr int MAX_K = 28;

// One layer inside a kernel-variant group.  Work item = (row range of the lists' `in` side, group of 4
// offsets, channel tile); the block's four waves take one offset each and walk that offset's pairs
// whose `in` row lies in the range: the segment [seg[o][t0], seg[o][t0 + RANGE_TILES]) of list o, read
// from the per-256-row-tile prefix the list export leaves behind.  All offsets of a row range touch the
// same neighbourhood of rows, and consecutive work items run on one XCD (xcd_work_item), so the rows
// are fetched from HBM once and served from L2 to the other offsets.  (Blocks split by OFFSET instead
// re-fetched every row into every XCD's L2: measured 63 us per level-1 layer at ~6 TB/s of fabric reads.)
struct PJob {
    const void *a, *b;
    const int32_t *pin, *pout, *pnum, *seg;   // seg: [K][seg_nt] exclusive prefix per tile, or NULL (identity lists)
    float *partial;      // [n_range][K][ca][cb]
    unsigned a_bytes, b_bytes;
    int ca, cb, ld, K, n_rows, n_range, n_og, n_tag, n_tbg, seg_nt, blk_end, pad;
};
struct RJob {            // dw[q] (+)= sum_r partial[r][q], q over K*ca*cb/4
    const float4 *partial;
    float4 *dw;
    long long n_quad;
    int R, accumulate, blk_end, pad;
};
// block -> job: the inclusive block prefixes travel in the kernel arguments (scalar cache), not in a
// dependent chain of global loads
constexpr int MAX_GROUP = 48;
struct Ends { int n; int end[MAX_GROUP]; };

__device__ __forceinline__ int find_end(const Ends &e, int blk) {
    int j = 0;
#pragma unroll 1
    for (int k = 0; k < e.n - 1; ++k) j += blk >= e.end[k] ? 1 : 0;
    return j;
}

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

// TA x TB 16-channel blocks of (a, b) per wave.
template <int TA, int TB>
__global__ __launch_bounds__(256) void wgrad_pairs_kernel(const PJob *__restrict__ jobs, const Ends ends) {
    const int jn = find_end(ends, (int)blockIdx.x);
    const PJob &d = jobs[jn];
    const int first = jn == 0 ? 0 : ends.end[jn - 1], n_items = ends.end[jn] - first;
    int lb = xcd_work_item((int)blockIdx.x - first, n_items);   // contiguous items per XCD
    const int tbg = lb % d.n_tbg; lb /= d.n_tbg;
    const int tag = lb % d.n_tag; lb /= d.n_tag;
    const int og = lb % d.n_og;
    const int range = lb / d.n_og;

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int o = og * 4 + wid;
    if (o >= d.K) return;     // (no barrier in this kernel)
    // the wave's segment of list o.  readfirstlane: the values size buffer descriptors, which must be
    // provably wave-uniform (else hipcc wraps each load in a waterfall loop)
    int p_begin, p_end;
    if (d.seg) {
        const int t0 = range * RANGE_TILES, t1 = t0 + RANGE_TILES;
        const int n_o = d.pnum[o];
        p_begin = d.seg[(long long)o * d.seg_nt + t0];
        p_end = t1 < d.seg_nt ? d.seg[(long long)o * d.seg_nt + t1] : n_o;
    } else {                  // identity lists (1x1 convolution): pair p = (p, p)
        p_begin = range * RANGE_ROWS;
        p_end = p_begin + RANGE_ROWS < d.ld ? p_begin + RANGE_ROWS : d.ld;
    }
    p_begin = __builtin_amdgcn_readfirstlane(p_begin);
    p_end = __builtin_amdgcn_readfirstlane(p_end);

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

    // index registers of one double-step: lane l holds the pair p + l.  Raw buffer loads over the list
    // up to p_end: a lane past the end reads 0 and is masked when the rows are requested.  No branch
    // anywhere in the loop: with control flow inside it hipcc falls back to s_waitcnt vmcnt(0) in front
    // of every MFMA group and nothing stays in flight.
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void *)pi, 0, (unsigned)p_end * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void *)po, 0, (unsigned)p_end * 4u, 0x00020000);
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

    // ---- the wave owns (range, offset): it writes that partial itself (zeros for an empty segment) ----
    float *out = d.partial + ((long long)range * d.K + o) * d.ca * d.cb;
#pragma unroll
    for (int xa = 0; xa < TA; ++xa)
#pragma unroll
        for (int yb = 0; yb < TB; ++yb) {
            // D[i = ci][j = co]: lane (co = lane & 15, g) holds ci = 4g + r
            const int co = (tbg * TB + yb) * 16 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = (tag * TA + xa) * 16 + 4 * g + r;
                if (ci < d.ca && co < d.cb) out[(long long)ci * d.cb + co] = acc[xa][yb][r];
            }
        }
}

// dw[q] (+)= sum_r partial[r][q]: 16 quads x 16 range lanes per block, lane r sums ranges r, r+16, ...
// and the lane sums are added in ascending r: fixed order.
__global__ __launch_bounds__(256) void wgrad_pairs_reduce(const RJob *__restrict__ jobs, int n_jobs) {
    __shared__ float4 part[16][16];
    const int jn = find_job(jobs, n_jobs, (int)blockIdx.x);
    const RJob d = jobs[jn];
    const int first = jn == 0 ? 0 : jobs[jn - 1].blk_end;
    const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long long q = (long long)((int)blockIdx.x - first) * 16 + el;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d.n_quad)
        for (int r = rl; r < d.R; r += 16) {
            const float4 v = d.partial[(long long)r * d.n_quad + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[rl][el] = s;
    doda_sync();
    if (rl == 0 && q < d.n_quad) {
        float4 t = part[0][el];
#pragma unroll 4
        for (int r = 1; r < 16; ++r) {
            const float4 v = part[r][el];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        if (d.accumulate) {
            const float4 old = d.dw[q];
            t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
        }
        d.dw[q] = t;
    }
}

struct Geo {
    int ta, tb, n_tag, n_tbg, n_og, n_range;
    bool direct;   // one range and no accumulation: the waves write dw themselves
};

Geo make_geo(const doda_wgrad_job &j) {
    Geo g;
    const int na = j.ca / 16, nb = j.cb / 16;
    g.ta = na >= 2 ? 2 : 1;
    g.tb = nb >= 2 ? 2 : 1;
    g.n_tag = div_up(na, g.ta);
    g.n_tbg = div_up(nb, g.tb);
    g.n_og = div_up(j.K, 4);
    // rows of the lists' `in` side: the segment prefix covers pair_seg_nt tiles; identity lists: pair_ld pairs
    const long long rows = j.pair_seg ? (long long)j.pair_seg_nt * SEG_TILE : (long long)j.pair_ld;
    g.n_range = div_up(rows > 0 ? rows : 1, RANGE_ROWS);
    g.direct = g.n_range == 1 && !(j.flags & DODA_WGRAD_ACCUMULATE);
    return g;
}

template <int TA, int TB>
void launch_variant(int blocks, const PJob *jobs_dev, const Ends &ends, hipStream_t s) {
    hipLaunchKernelGGL((wgrad_pairs_kernel<TA, TB>), dim3(blocks), dim3(256), 0, s, jobs_dev, ends);
}

}  // namespace

namespace doda_pairs {

bool eligible(const doda_wgrad_job &j) {
    if (j.elem_bytes != 2 || j.ca <= 0 || j.cb <= 0 || (j.ca % 16) || (j.cb % 16) || j.K <= 0 || j.n_rows <= 0)
        return false;
    if (!j.a || !j.b || !j.dw || j.K > MAX_K) return false;
    if (!j.pair_in || !j.pair_out || j.pair_ld <= 0 || j.n_a <= 0) return false;
    // real lists come with their counts and segment prefix; the identity lists of a 1x1 conv with neither
    if (j.pair_num ? (!j.pair_seg || j.pair_seg_nt <= 0) : (j.pair_seg != nullptr || j.K != 1)) return false;
    if ((unsigned long long)j.n_a * j.ca * 2ull >= 0x3fffffffull) return false;
    if ((unsigned long long)j.n_rows * j.cb * 2ull >= 0x3fffffffull) return false;
    if (((uintptr_t)j.a % 16) || ((uintptr_t)j.b % 16) || ((uintptr_t)j.dw % 16)) return false;
    const Geo g = make_geo(j);
    if ((long long)g.n_range * g.n_og * g.n_tag * g.n_tbg > 0x3fffffff) return false;
    return true;
}

size_t partial_bytes(const doda_wgrad_job &j) {
    const Geo g = make_geo(j);
    if (g.direct) return 0;
    return align_up((size_t)g.n_range * j.K * j.ca * j.cb * 4, 256);
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
            auto flush = [&]() {
                if (grp.count) out->groups.push_back(grp);
                grp = Prepared::Group{ta, tb, (int)pj.size(), 0, 0};
            };
            for (int k = 0; k < n; ++k) {
                const doda_wgrad_job &j = jobs[which[k]];
                const Geo g = make_geo(j);
                if (g.ta != ta || g.tb != tb) continue;
                if (grp.count == MAX_GROUP) flush();
                PJob d;
                memset(&d, 0, sizeof(d));
                d.a = j.a; d.b = j.b;
                d.pin = j.pair_in; d.pout = j.pair_out; d.pnum = j.pair_num; d.seg = j.pair_seg;
                d.partial = g.direct ? j.dw : (float *)(ws_base + offs[k]);
                d.a_bytes = (unsigned)((size_t)j.n_a * j.ca * 2);
                d.b_bytes = (unsigned)((size_t)j.n_rows * j.cb * 2);
                d.ca = j.ca; d.cb = j.cb; d.ld = j.pair_ld; d.K = j.K; d.n_rows = j.n_rows;
                d.n_range = g.n_range; d.n_og = g.n_og; d.n_tag = g.n_tag; d.n_tbg = g.n_tbg;
                d.seg_nt = j.pair_seg_nt;
                grp.blocks += g.n_range * g.n_og * g.n_tag * g.n_tbg;
                d.blk_end = grp.blocks;
                pj.push_back(d);
                ++grp.count;
            }
            flush();
        }
    int r_blocks = 0;
    for (int k = 0; k < n; ++k) {
        const doda_wgrad_job &j = jobs[which[k]];
        const Geo g = make_geo(j);
        if (g.direct) continue;
        RJob d;
        memset(&d, 0, sizeof(d));
        d.partial = (const float4 *)(ws_base + offs[k]);
        d.dw = (float4 *)j.dw;
        d.n_quad = (long long)j.K * j.ca * j.cb / 4;
        d.R = g.n_range;
        d.accumulate = (j.flags & DODA_WGRAD_ACCUMULATE) ? 1 : 0;
        r_blocks += div_up(d.n_quad, 16);
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
    const PJob *pj_h = (const PJob *)p.desc.data();
    for (const Prepared::Group &g : p.groups) {
        Ends ends;
        memset(&ends, 0, sizeof(ends));
        ends.n = g.count;
        for (int k = 0; k < g.count; ++k) ends.end[k] = pj_h[g.first + k].blk_end;
        if (g.ta == 1 && g.tb == 1) launch_variant<1, 1>(g.blocks, pj + g.first, ends, s);
        else if (g.ta == 2 && g.tb == 1) launch_variant<2, 1>(g.blocks, pj + g.first, ends, s);
        else if (g.ta == 1 && g.tb == 2) launch_variant<1, 2>(g.blocks, pj + g.first, ends, s);
        else launch_variant<2, 2>(g.blocks, pj + g.first, ends, s);
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
