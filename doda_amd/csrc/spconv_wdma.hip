// wgrad_dma16: weight gradient of the SubM 16 -> 16 bf16 layers over the rulebook's TILEBOOK
//     dw[o][ci][co] (+)= sum_t x[tbl[o][t]][ci] * dy[t][co]
// (spconv v1.2 indice_conv_backward's per-offset `Xg^T . dYg`; reference call sites model/unet_block.py:26,29 through
// autograd), for ALL layers of a rulebook in one launch.
//
// The pair-list kernel (spconv_wgrad_pairs.hip) gathers one 32-byte x row and one 32-byte dy row PER PAIR from
// global memory: 12.4 pairs per voxel x 64 B = 0.8 KB of texture-path traffic per voxel against 137 B of operands
// (PMC: GRBM_TA_BUSY 93 %); cold it runs at 57 us per level-1 layer.  Here a tile of 256 output rows t is staged
// ONCE in LDS — the ~2.5 x 256 distinct x rows its table entries reference (LDS-DMA through the tilebook's list, as the
// round-3 conv_dma16 experiment staged them; pieces past the list's end are not issued) and its 256 dy rows — and
// every (offset, 32-row k-step) is served from there.  BOTH operands reach MFMA k-order through ds_read_b64_tr_b16:
//   * dy (dense: rows t0 .. t0+255) with contiguous addresses (conflict-free);
//   * the gathered x rows with PER-LANE addresses built from the tile's local indices: lane 4q + c of a 16-lane group
//     points at chunk c (4 channels) of the staged row that output row q's index selects (an absent neighbour: the
//     shared zero row) and receives channel i of the four rows — ~2.5-way bank conflicts on randomly placed rows.
// Per (offset, k-step): one 8-byte local-index read per two steps, two transposed reads, one MFMA.
// (Round 3's first form transposed the gathered 16-byte row slices on the matrix core — two one-hot MFMAs + four v_perm
// per step, as the pair kernel does.  PMC showed it ISSUE bound: MFMA pipe 30 % busy, ~19 VALU per step, LDS only 33 %
// busy, 34 % issue stalls; the transposed gather trades that for LDS conflicts the LDS had room for: 39 -> 31.5 us cold
// per level-1 layer.  profiles/r03_pmc_wgrad_dma16.txt.)
// ONE persistent 16-wave workgroup per CU with two tile buffers: the DMA of tile j+1 is issued between the first steps
// of tile j, each wave at a different step (the waves leave the barrier together, and sixteen waves queueing on the CU's
// texture path at the same instant stall each other instead of overlapping with the matrix work).
// The 54 units (offset, half of the tile's k-steps) are dealt to the 16 waves; accumulators stay in registers across
// all tiles a workgroup has of a 16 x 16 channel block; per (workgroup, block it touches) one partial [27][16][16] is written
// (block-major chunks: see the schedule in the kernel), summed in a fixed order by wgrad_dma_reduce: deterministic.
#include "common.hpp"
#include "tilebook.hpp"
#include "spconv_common.hpp"
#include "wgrad_pairs.hpp"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace {

constexpr int WD_WAVES = 16;
constexpr int WD_ROWS_BYTES = (TB_UMAX + 1) * 32;     // slot 0: the shared zero row
constexpr int WD_LIDX_BYTES = TB_LIDX_BYTES;          // ten planes x 256 packed words = 10240 = ten 1 KB DMA pieces (round 4; was 16 pieces)
constexpr int WD_DY_BYTES = TB_T * 32;
constexpr int WD_UNITS_BYTES = 2 * TB_K * 256 * 4;    // the per-block exchange of the units' accumulators re-uses a tile buffer
constexpr int WD_BUF_BYTES = (WD_ROWS_BYTES + WD_LIDX_BYTES + WD_DY_BYTES) > WD_UNITS_BYTES ? (WD_ROWS_BYTES + WD_LIDX_BYTES + WD_DY_BYTES)
                                                                                           : WD_UNITS_BYTES;
static_assert(WD_LIDX_BYTES % 1024 == 0, "whole DMA pieces");
constexpr int WD_UNITS = 2 * TB_K;                    // (offset, half of the k-steps)
constexpr int WD_MAX_UNITS = (WD_UNITS + WD_WAVES - 1) / WD_WAVES;   // 4
constexpr int WD_MAX_JOBS = 16;
static_assert(WD_BUF_BYTES % 16 == 0 && 2 * WD_BUF_BYTES + 64 <= 160 * 1024, "two tile buffers per CU");
static_assert(WD_UNITS * 256 * 4 <= WD_BUF_BYTES, "the final exchange re-uses ONE tile buffer (the other may be a DMA target)");

// One 16 x 16 channel block of a layer: x / dy point at the block's first channel, rows lie x_stride / dy_stride BYTES apart
// (32 for a 16-channel tensor; 64 for a 16-channel half of a 32-channel one: round 4 — a 32 -> 16 layer is two blocks over
// the halves of x, each staged through the same 32-byte LDS rows).  part: [groups][27][256] of this block.
struct WdJob { const void *x, *dy; float *part; unsigned x_stride, dy_stride; int first_rank, pad; };   // first_rank: see the schedule
struct WdJobs { int n; WdJob j[WD_MAX_JOBS]; };

__device__ __forceinline__ u32x4 wd_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}
// 16 bytes per lane from buffer offset `voff` straight into LDS at lds_base + lane * 16 (lds_base wave-uniform)
__device__ __forceinline__ void wd_dma16(unsigned lds_base, unsigned voff, const u32x4 &rs) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void wd_aload128(u32x4 &dst, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
// TAG: call sites on the two sides of a branch carry different tags — identical asm statements at the head of both
// successors are hoisted above the branch, and the values then cross it through register copies that hipcc places
// BEFORE the hand-written s_waitcnt (it does not know the registers are still in flight)
template <int TAG = 0>
__device__ __forceinline__ s16x4 wd_tr_b64(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 ; %2" : "=v"(v) : "v"(addr), "n"(TAG) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 wd_pack_hi16(const f32x4 &d0, const f32x4 &d1) {
    // the values are bf16-exact: keep the upper halves.  k-slot q of the lane: q < 4 -> d0[q], else d1[q-4]
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(__float_as_uint(d0[1]), __float_as_uint(d0[0]), 0x07060302u);
    r[1] = __builtin_amdgcn_perm(__float_as_uint(d0[3]), __float_as_uint(d0[2]), 0x07060302u);
    r[2] = __builtin_amdgcn_perm(__float_as_uint(d1[1]), __float_as_uint(d1[0]), 0x07060302u);
    r[3] = __builtin_amdgcn_perm(__float_as_uint(d1[3]), __float_as_uint(d1[2]), 0x07060302u);
    return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(1024) void wgrad_dma16(const WdJobs jobs, const int32_t *__restrict__ tbl,
                                                    int ld, int n, const TileBookView tb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WD_BUF_BYTES];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const u32x4 rs_ul = wd_rsrc(tb.ulist, (unsigned)tb.nt * (unsigned)TB_UMAX * 4u);
    const u32x4 rs_li = wd_rsrc(tb.lidx, (unsigned)tb.nt * (unsigned)TB_LIDX_BYTES);

    // Persistent schedule (round 4: BLOCK-major).  The (block, tile) items of the launch, block-major, are cut into gridDim.x
    // contiguous chunks; workgroup rank r = (XCD, slot) takes chunk r, so an XCD owns one contiguous stretch of the list
    // (adjacent tiles share most of their neighbour rows: L2 reuse) and a workgroup sees at most two or three blocks: it
    // writes ONE partial per block it touches — (gridDim.x + blocks) partials per launch.  Round 3 let every workgroup
    // walk every block of the launch over its own strided tiles: blocks x gridDim.x partials (76 MB written and read
    // again at level 1 with 11 blocks, a flush of four barriers per block and workgroup).
    const int G = gridDim.x, nt = tb.nt;
    const int rank = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    const long long N = (long long)jobs.n * nt;
    const int start = (int)((long long)rank * N / G), n_items = (int)((long long)(rank + 1) * N / G) - start;
    if (n_items == 0) return;     // (more workgroups than items: no partial slot was counted for this one)

    if (tid < 4) reinterpret_cast<u32x4 *>(smem + (tid >> 1) * WD_BUF_BYTES)[tid & 1] = (u32x4){0u, 0u, 0u, 0u};   // zero rows

    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    const unsigned src_half = (unsigned)(lane & 1) * 16u;      // rows land linearly: slot l at byte 32 l (read by transposed gathers)
    // Staging of one item by 16 waves.  Wave w moves two of the 32 row pieces — pieces (kb*8 + w8) and ((kb+1)*8 + w8)
    // with w8 = w & 7, kb = 2 (w >> 3): in the list's storage order (tb_upos) their entries are one 8-byte load per
    // lane —, one of the 16 index-strip pieces, and (waves 0..7) one of the 8 dy pieces.
    const int w8 = wid & 7, kb = (wid >> 3) * 2;
    struct Where { int job; unsigned tile; bool ok; };
    auto where = [&](int item) {     // (one scalar division per item, not per piece)
        Where q;
        q.ok = item < n_items;
        const int gid = start + (q.ok ? item : 0);
        q.job = gid / nt;
        q.tile = (unsigned)(gid - q.job * nt);
        return q;
    };
    auto issue_list = [&](const Where &q, u32x2 &rid) {
        asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(rid)
                     : "v"(q.ok ? q.tile * (unsigned)(TB_UMAX * 4) + (unsigned)((w8 * 32 + (lane >> 1)) * 4 + kb) * 4u : OOB), "s"(rs_ul)
                     : "memory");
    };
    auto issue_rows = [&](const Where &q, int item, int k, const u32x2 &rid) {      // k = 0, 1
        const unsigned buf = smem_base + (unsigned)(item & 1) * (unsigned)WD_BUF_BYTES;
        const unsigned xs = jobs.j[q.job].x_stride;
        const u32x4 rs_x = wd_rsrc(jobs.j[q.job].x, (unsigned)n * xs - (xs - 32u));
        // The list is sorted with the absent entries (negative) at its end, and a piece covers 32 consecutive entries: a
        // piece whose FIRST entry is absent stages nothing any local index points at — not issued (40 % of the pieces at
        // ~600 distinct rows per tile; every wait in this kernel is vmcnt(0), so counts may differ between waves).  Inside
        // the last used piece an absent entry's row offset is out of range and lands as zeros.
        if (!q.ok || __builtin_amdgcn_readfirstlane((int)rid[k]) < 0) return;
        wd_dma16(buf + 32u + (unsigned)(((kb + k) * 8 + w8) * 1024), rid[k] * xs + src_half, rs_x);
    };
    auto issue_strip = [&](const Where &q, int item) {
        const unsigned buf = smem_base + (unsigned)(item & 1) * (unsigned)WD_BUF_BYTES;
        if (wid < WD_LIDX_BYTES / 1024)      // (wave-uniform: ten pieces)
            wd_dma16(buf + (unsigned)WD_ROWS_BYTES + (unsigned)(wid * 1024),
                     q.ok ? q.tile * (unsigned)TB_LIDX_BYTES + (unsigned)(wid * 64 + lane) * 16u : OOB, rs_li);
    };
    auto issue_dy = [&](const Where &q, int item) {
        if (wid < 8) {      // (wave-uniform; every wait in this kernel is vmcnt(0), so waves need not issue equal counts)
            const unsigned buf = smem_base + (unsigned)(item & 1) * (unsigned)WD_BUF_BYTES;
            const unsigned ds = jobs.j[q.job].dy_stride;
            const u32x4 rs_dy = wd_rsrc(jobs.j[q.job].dy, (unsigned)n * ds - (ds - 32u));
            // rows past n: out of range -> zeros
            wd_dma16(buf + (unsigned)(WD_ROWS_BYTES + WD_LIDX_BYTES) + (unsigned)(wid * 1024),
                     q.ok ? (q.tile * (unsigned)TB_T + (unsigned)(wid * 32 + (lane >> 1))) * ds + (unsigned)(lane & 1) * 16u : OOB, rs_dy);
        }
    };

    f32x4 acc[WD_MAX_UNITS];
#pragma unroll
    for (int m = 0; m < WD_MAX_UNITS; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // prologue: list of items 0 and 1, DMA of item 0
    u32x2 la, lb;
    {
        const Where q0 = where(0), q1 = where(1);
        issue_list(q0, la);
        issue_list(q1, lb);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(la), "+v"(lb) : : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // zero rows written
        issue_rows(q0, 0, 0, la);
        issue_rows(q0, 0, 1, la);
        issue_strip(q0, 0);
        issue_dy(q0, 0);
    }

    // body(item, lnext = list(item+1) [landed], lnew <- list(item+2))
    auto body = [&](int item, bool no_list, const u32x2 &lnext, u32x2 &lnew) {
        const Where qc = where(item), q1 = where(item + 1), q2 = where(item + 2);
        const int t0 = (int)qc.tile * TB_T;
        const unsigned char *buf = smem + (item & 1) * WD_BUF_BYTES;
        const unsigned rows_base = smem_base + (unsigned)(item & 1) * (unsigned)WD_BUF_BYTES;
        const unsigned *lidx_s = reinterpret_cast<const unsigned *>(buf + WD_ROWS_BYTES);   // ten planes of packed local indices (tilebook.hpp)
        const unsigned dy_base = rows_base + (unsigned)(WD_ROWS_BYTES + WD_LIDX_BYTES);
        const unsigned short *xg = reinterpret_cast<const unsigned short *>(jobs.j[qc.job].x);
        issue_list(q2, lnew);        // first thing: it has the whole tile to land (lnew held list(item): dead)
        // ---- dy fragments of the wave's four k-steps (every unit of wave w covers the half h = w & 1 of the tile):
        // channel i of rows 32 ks + 8 g + 0..7 ----
        const int q4 = i >> 2, c4 = i & 3, hw = wid & 1;
        bf16x8 bt[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned a = dy_base + (unsigned)((32 * (4 * hw + kk) + 8 * g + q4) * 32 + c4 * 8);
            const s16x4 lo4 = wd_tr_b64(a), hi4 = wd_tr_b64(a + 4u * 32u);
            bt[kk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bt[0]), "+v"(bt[1]), "+v"(bt[2]), "+v"(bt[3]) : : "memory");
        // (overflow path below) lane (i, g) reads the half g & 1 of the row of output row 32 ks + 8 (i >> 2) + 4 (g >> 1) + (i & 3)
        const int rl = 8 * (i >> 2) + 4 * (g >> 1) + (i & 3);
        const unsigned hsel = (unsigned)(g & 1);
        if (!no_list) {
            // The gathered operand comes out of LDS already in MFMA k-order: ds_read_b64_tr_b16 with per-lane addresses is a
            // gather — lane 4 q + c of a 16-lane group points at chunk c (4 channels) of the staged row that the local index
            // of output row 32 ks + 8 g + q selects and receives channel i of the rows q = 0..3 (second read: q + 4).
            // Round 3 first used the matrix core for this (two one-hot MFMAs + four v_perm per k-step): PMC showed that
            // form ISSUE-bound (per SIMD 30 % MFMA busy + ~19 VALU per step, LDS only 33 % busy: profiles/
            // r03_pmc_wgrad_dma16.txt); the transposed gather costs ~2.5-way bank conflicts on an LDS that has the room.
            // Per k-step: one 8-byte index read (two k-steps x lo/hi halves share two of them), two transposed reads, one
            // MFMA.  Addresses of all the wave's steps first, reads three steps ahead of the MFMAs.
            // Two units (eight steps) at a time: their sixteen row addresses live in registers, the next pair's are computed
            // when these are spent (the wave budget is 128 VGPRs at 16 waves per workgroup).
            const unsigned lane_off = rows_base + (unsigned)c4 * 8u;
            const bool tail_unit = wid + WD_WAVES * (WD_MAX_UNITS - 1) < WD_UNITS;   // the waves that own a unit 48 .. 53
#pragma unroll
            for (int mp = 0; mp < WD_MAX_UNITS / 2; ++mp) {
                unsigned ra[8], rb[8];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int unit = wid + WD_WAVES * (2 * mp + mm);
                    const int o = unit < WD_UNITS ? unit >> 1 : 0;      // (a unit past the end: any valid strip, result dropped)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) {
                        // k-steps ks = 4 hw + 2 kp (+1): the lane's operand rows are 32 ks + 8 g + q4 (+4) of the tile; their local
                        // indices for offset o: bits tb_lshift(o) .. +9 of their word in plane tb_lplane(o) (tilebook.hpp)
                        // (plane tb_lplane(o), word tb_lpos(row): row r + 32 is two subtiles = 2 words on, row r + 4 — bit 2 of
                        // the row, bit 3 of its slot — 32 words)
                        const unsigned *sp = lidx_s + tb_lplane(o) * TB_T + tb_lpos((2 * hw + kp) * 64 + 8 * g + q4);
                        const unsigned shb = tb_lshift(o);
                        ra[mm * 4 + 2 * kp] = (__builtin_amdgcn_ubfe(sp[0], shb, 10u) << 5) + lane_off;
                        ra[mm * 4 + 2 * kp + 1] = (__builtin_amdgcn_ubfe(sp[2], shb, 10u) << 5) + lane_off;
                        rb[mm * 4 + 2 * kp] = (__builtin_amdgcn_ubfe(sp[32], shb, 10u) << 5) + lane_off;
                        rb[mm * 4 + 2 * kp + 1] = (__builtin_amdgcn_ubfe(sp[34], shb, 10u) << 5) + lane_off;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // (the whole pipeline of a pair — first reads included — sits on ONE side of the branch below: a value of an
                // inline-asm read that crossed it would be copied before its wait)
                auto run_pair = [&](auto tagc) {
                    constexpr int TAG = decltype(tagc)::value;
                    constexpr int last = TAG == 1 ? 4 : 8;     // steps the wave runs in this pair
                    s16x4 xl[3], xh[3];
                    xl[0] = wd_tr_b64<TAG>(ra[0]); xh[0] = wd_tr_b64<TAG>(rb[0]);
                    xl[1] = wd_tr_b64<TAG>(ra[1]); xh[1] = wd_tr_b64<TAG>(rb[1]);
#pragma unroll
                    for (int st = 0; st < last; ++st) {
                        const int gs = mp * 8 + st, m = gs >> 2, kk = gs & 3;
                        // the next item's DMA: in the FIRST steps of the tile (two buffers: it needs the rest of the tile to
                        // land), odd and even waves on alternating steps
                        if (gs == (wid & 1)) issue_rows(q1, item + 1, 0, lnext);
                        if (gs == 2 + (wid & 1)) issue_rows(q1, item + 1, 1, lnext);
                        if (gs == 4 + (wid & 1)) issue_strip(q1, item + 1);
                        if (gs == 6 + (wid & 1)) issue_dy(q1, item + 1);
                        if (st + 2 < last) { xl[(st + 2) % 3] = wd_tr_b64<TAG>(ra[st + 2]); xh[(st + 2) % 3] = wd_tr_b64<TAG>(rb[st + 2]); }
                        // LDS returns in order: everything but the reads of the steps after this one has landed
                        const int newer = st + 2 < last ? 4 : 2 * (last - 1 - st);
                        if (newer == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xl[st % 3]), "+v"(xh[st % 3]) : : "memory");
                        else if (newer == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xl[st % 3]), "+v"(xh[st % 3]) : : "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xl[st % 3]), "+v"(xh[st % 3]) : : "memory");
                        const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xl[st % 3], xh[st % 3], 0, 1, 2, 3, 4, 5, 6, 7));
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bt[kk], acc[m], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (mp < WD_MAX_UNITS / 2 - 1 || tail_unit) run_pair(std::integral_constant<int, 0>{});
                else run_pair(std::integral_constant<int, 1>{});      // the last pair of a wave without a fourth unit: four steps
            }
        } else {
            // a tile without a list (more than TB_LMAX distinct rows; none at 2 cm): slices through the dense table
            issue_rows(q1, item + 1, 0, lnext);
            issue_rows(q1, item + 1, 1, lnext);
            issue_strip(q1, item + 1);
            issue_dy(q1, item + 1);
            // here the row slices (a 16-byte slice is the natural A operand: lane = row, registers = channels) reach k-order
            // through the matrix core, as in the pair kernel: multiplied by a one-hot B operand they come back with
            // lane = channel, registers = rows.  P[G][k = (g, q)][j = i] = 1 iff g >> 1 == G, g & 1 == j >> 3, q == j & 7
            bf16x8 P[2];
#pragma unroll
            for (int G = 0; G < 2; ++G) {
                u32x4 v = {0u, 0u, 0u, 0u};
                const bool mine = (g >> 1) == G && (i >> 3) == (g & 1);
                const int q = i & 7;
#pragma unroll
                for (int w = 0; w < 4; ++w) v[w] = (mine && (q >> 1) == w) ? ((q & 1) ? 0x3F800000u : 0x00003F80u) : 0u;
                P[G] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll 1
            for (int m = 0; m < WD_MAX_UNITS; ++m) {
                const int unit = wid + WD_WAVES * m;
                if (unit >= WD_UNITS) break;
                const int o = unit >> 1;
                f32x4 part = {0.f, 0.f, 0.f, 0.f};
                // all four table entries, then all four slices: two round trips per unit
                int gi[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int r = 32 * (4 * hw + kk) + rl;
                    gi[kk] = t0 + r < n ? tbl[(size_t)o * ld + (size_t)(t0 + r)] : -1;
                }
                u32x4 v[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    v[kk] = (u32x4){0u, 0u, 0u, 0u};
                    if (gi[kk] >= 0) v[kk] = *reinterpret_cast<const u32x4 *>(xg + (size_t)gi[kk] * (jobs.j[qc.job].x_stride >> 1) + hsel * 8);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, v[kk]);
                    const f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, P[0], zero, 0, 0, 0);
                    const f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, P[1], zero, 0, 0, 0);
                    part = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd_pack_hi16(e0, e1), bt[kk], part, 0, 0, 0);
                }
#pragma unroll
                for (int mm = 0; mm < WD_MAX_UNITS; ++mm)
                    if (mm == m) acc[mm] += part;      // (static indices: acc[] stays in registers)
            }
        }
    };

    // the layer's partial: units -> LDS, halves added, one [27][256] block per workgroup
    // (exchange through the buffer the last tile was read from: the other one is the target of the next item's DMA)
    auto flush = [&](int job, int bufsel, int last_gid) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading the buffer
        float *ex = reinterpret_cast<float *>(smem + bufsel * WD_BUF_BYTES);
#pragma unroll
        for (int m = 0; m < WD_MAX_UNITS; ++m) {
            const int unit = wid + WD_WAVES * m;
            if (unit < WD_UNITS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ex[unit * 256 + (4 * g + r) * 16 + i] = acc[m][r];
            }
            acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // partial slot inside the block: consecutive ranks share a block when every rank has items (N >= G); with fewer items
        // than workgroups every non-empty rank holds exactly one tile and the tile's index is the slot
        const int pslot = N >= G ? rank - jobs.j[job].first_rank : last_gid - job * nt;
        float *dst = jobs.j[job].part + (size_t)pslot * (TB_K * 256);
        for (int e = tid; e < TB_K * 256; e += 1024) {
            const int o = e >> 8, c = e & 255;
            dst[e] = ex[(2 * o) * 256 + c] + ex[(2 * o + 1) * 256 + c];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < 2) reinterpret_cast<u32x4 *>(smem + bufsel * WD_BUF_BYTES)[tid] = (u32x4){0u, 0u, 0u, 0u};   // its zero row again
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    bool nl_cur = __builtin_amdgcn_readfirstlane((int)la[0]) == -2;      // tile 0 has no list
    for (int item = 0; item < n_items; item += 2) {
        // everything issued so far has landed: DMA(item), list(item+1)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(lb) : : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const bool nl_next = __builtin_amdgcn_readfirstlane((int)lb[0]) == -2;
            body(item, nl_cur, lb, la);
            nl_cur = nl_next;
        }
        if (item + 1 == n_items || (start + item + 1) % nt == 0) flush((start + item) / nt, item & 1, start + item);   // the block's last tile here
        if (item + 1 >= n_items) break;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(la) : : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const bool nl_next = __builtin_amdgcn_readfirstlane((int)la[0]) == -2;
            body(item + 1, nl_cur, la, lb);
            nl_cur = nl_next;
        }
        if (item + 2 == n_items || (start + item + 2) % nt == 0) flush((start + item + 1) / nt, (item + 1) & 1, start + item + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// dw[job][e] (+)= sum over workgroups of part[job][wg][e], fixed order.  One workgroup per 32 outputs and layer; its 16
// lane groups take every 16th partial, eight loads in flight each, combined in a fixed tree through LDS.
// e = (o, ci, co) of the 16 x 16 block lands at dw[o * ldo + ci * ldc + co] (dw already points at the block's corner)
struct WdRJob { const float *part; float *dw; int accumulate, ldo, ldc, n_part; };   // n_part: partials of this block
struct WdRJobs { WdRJob j[WD_MAX_JOBS]; };

__global__ __launch_bounds__(512) void wgrad_dma_reduce(const WdRJobs jobs) {
    __shared__ float red[16][32];
    const WdRJob d = jobs.j[blockIdx.y];
    const int n_part = d.n_part;
    const int jx = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + jx;
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.f;
    int b = p;
    for (; b + 7 * 16 < n_part; b += 8 * 16) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += d.part[(size_t)(b + q * 16) * (TB_K * 256) + e];
    }
    for (int q = 0; b < n_part; b += 16, ++q) a[q & 7] += d.part[(size_t)b * (TB_K * 256) + e];
    red[p][jx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    doda_sync();
    if (p == 0) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][jx];
        float *dst = d.dw + (size_t)(e >> 8) * d.ldo + (size_t)((e >> 4) & 15) * d.ldc + (e & 15);
        *dst = d.accumulate ? *dst + v : v;
    }
}

// (Round 4 built the fp32 counterpart — wgrad_tile_f32: x rows, index strip and dy rows staged in LDS, the next tile held in
// registers during the multiply, rows as the k dimension of v_mfma_f32_16x16x4_f32, k-steps without a present neighbour
// skipped — parity-green against the oracle and SLOWER than the gather-table kernel: 132-143 us against 110 us per level-1
// layer (8 layers per call).  The fp32 matrix rate (1/16 of bf16) puts 53 us of MFMA issue under a dense layer, 4-byte LDS
// operand reads (one per lane and k-step: 4.6 k LDS instructions per tile) another ~45 us, and sixteen waves leaving a
// barrier together run the two phases one after the other.  Removed)
bool g_use_wdma = !(getenv("DODA_NO_WDMA") && getenv("DODA_NO_WDMA")[0] == '1');

}  // namespace

namespace doda_wdma {

bool enabled() { return g_use_wdma; }
void set_enabled(bool on) { g_use_wdma = on; }

int groups_for(int n_rows) {
    const int nt = (n_rows + TB_T - 1) / TB_T;
    int groups = (nt + 7) / 8 * 8;
    return groups > 256 ? 256 : groups;
}
size_t partial_bytes(int n_rows) { return (size_t)groups_for(n_rows) * TB_K * 256 * sizeof(float); }
int max_jobs() { return WD_MAX_JOBS; }

// blocks[k]: one 16 x 16 channel block of a layer (all over the same table / tilebook, n_rows rows); part: n x partial_bytes
int launch(const Block *blocks, int n_blocks, const int32_t *tbl, int ld, int n_rows, const void *tilebook, void *part,
           hipStream_t s) {
    const TileBookView tb = tilebook_view(const_cast<void *>(tilebook), n_rows);
    const int groups = groups_for(n_rows), nt = tb.nt;
    size_t used = 0;      // partial slots handed out so far (bytes)
    for (int first = 0; first < n_blocks; first += WD_MAX_JOBS) {
        const int nj = n_blocks - first < WD_MAX_JOBS ? n_blocks - first : WD_MAX_JOBS;
        WdJobs jobs;
        WdRJobs rj;
        ::memset(&jobs, 0, sizeof(jobs));
        ::memset(&rj, 0, sizeof(rj));
        jobs.n = nj;
        // which ranks touch which block (the kernel's chunking, restated): first rank and count per block
        int first_rank[WD_MAX_JOBS], count[WD_MAX_JOBS];
        for (int k = 0; k < nj; ++k) { first_rank[k] = -1; count[k] = 0; }
        const long long N = (long long)nj * nt;
        if (N >= groups) {
            for (int r = 0; r < groups; ++r) {
                const long long st = (long long)r * N / groups, en = (long long)(r + 1) * N / groups;
                for (long long b = st / nt; b <= (en - 1) / nt; ++b) {
                    if (first_rank[b] < 0) first_rank[b] = r;
                    ++count[b];
                }
            }
        } else {      // fewer items than workgroups: one tile per non-empty rank, slot = tile index
            for (int k = 0; k < nj; ++k) { first_rank[k] = 0; count[k] = nt; }
        }
        for (int k = 0; k < nj; ++k) {
            const Block &b = blocks[first + k];
            float *p = (float *)((char *)part + used);
            used += (size_t)count[k] * TB_K * 256 * sizeof(float);
            jobs.j[k] = WdJob{b.x, b.dy, p, (unsigned)b.x_stride, (unsigned)b.dy_stride, first_rank[k], 0};
            rj.j[k] = WdRJob{p, b.dw, b.accumulate, b.ldo, b.ldc, count[k]};
        }
        hipLaunchKernelGGL(wgrad_dma16, dim3(groups), dim3(1024), 0, s, jobs, tbl, ld, n_rows, tb);
        hipLaunchKernelGGL(wgrad_dma_reduce, dim3(TB_K * 256 / 32, nj), dim3(512), 0, s, rj);
    }
    return doda_check_launch();
}

}  // namespace doda_wdma

