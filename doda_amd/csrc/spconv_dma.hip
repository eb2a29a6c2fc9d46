// conv_dma16: the SubM 16 -> 16 bf16 tile kernel (forward and data gradient of DODA's level-1 block convolutions,
// reference call sites model/unet_block.py:26,29 through spconv's indice_conv / indice_conv_backward) as a
// PERSISTENT ONE-WORKGROUP-PER-CU PIPELINE fed by LDS-DMA.
//
// conv_tile (spconv_tile.hip) keeps three 4-wave workgroups per CU, each with one tile's neighbourhood in LDS and
// the loads of at most one more tile in flight; measured cold (operands from HBM, as inside the training step) a
// tile iteration takes ~9 us of which < 1 us is arithmetic — the CU simply has too few bytes in flight
// (Little's law: 8 TB/s / 256 CUs x ~2 us loaded latency = ~60 KB per CU; a tile is ~45 KB).  Here ONE 8-wave
// workgroup owns the CU's LDS as THREE tile buffers (3 x 48 KB) and keeps the staging loads of TWO tiles ahead of
// the one it multiplies in flight, written straight into LDS by `buffer_load_dwordx4 ... lds` (no staging
// registers, no ds_write pass):
//
//   iteration j:   a  request list(j+3), epilogue operands(j+1)                      -> registers (asm loads)
//                  b  s_waitcnt vmcnt(N): DMA(j), list(j+2), operands(j) have landed; DMA(j+1) stays in flight
//                  c  s_barrier  (everybody's pieces of tile j are in; everybody is done with tile j-1's buffer)
//                  d  DMA(j+2): rows of list(j+2) + the index strip -> buffer (j+2) % 3
//                  e  multiply tile j out of buffer j % 3 (weights held in 56 registers: no vector-memory
//                     instruction in the loop), epilogue, stores
//
// vmcnt retires in order, so every wait is a compile-time constant as long as each wave issues the same number
// of vector-memory instructions per iteration: lists / DMA pieces past the end read out-of-range offsets (zeros),
// nobody is masked out of EXEC.  hipcc knows nothing of the asm loads and DMA (it would drain them at every LDS
// read and barrier), so all waits and barriers inside the loop are written by hand (guide: "Pipelining across
// barriers").  Same arithmetic as conv_tile: pairs of offsets (2u, 2u+1) per v_mfma_f32_16x16x32_bf16, fp32
// accumulation, one bf16 rounding at the store; same epilogue options (residual add, BatchNorm statistics in
// forward and backward form, one partial row per tile).
#include "common.hpp"
#include "tilebook.hpp"
#include "spconv_common.hpp"
#include <stdlib.h>

namespace {

constexpr int DM_WAVES = 8;
constexpr int DM_CAP = 1024;                          // distinct rows staged per tile (list entries 0 .. 1023)
constexpr int DM_ROWS_BYTES = (DM_CAP + 1) * 32;      // slot 0: the shared zero row
constexpr int DM_LIDX_BYTES = 16384;                  // 27 x 256 x 2 = 13824, rounded up to whole 1 KB DMA pieces
constexpr int DM_BUF_BYTES = DM_ROWS_BYTES + DM_LIDX_BYTES;
constexpr int DM_NBUF = 3;
constexpr int DM_ROW_PIECES = DM_CAP * 2 / 64 / DM_WAVES;     // DMA instructions per wave and tile: rows (4)
constexpr int DM_LIDX_PIECES = DM_LIDX_BYTES / 1024 / DM_WAVES;   // index strip (2)
constexpr int DM_NG = DM_ROW_PIECES + DM_LIDX_PIECES;
static_assert(DM_ROW_PIECES * 64 * DM_WAVES == DM_CAP * 2 && DM_LIDX_PIECES * 1024 * DM_WAVES == DM_LIDX_BYTES, "uniform DMA counts");
static_assert(DM_BUF_BYTES % 16 == 0, "buffers stay 16-byte aligned");

__device__ __forceinline__ u32x4 dm_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}

// 16 bytes per lane from buffer offset `voff` straight into LDS at lds_base + lane * 16 (lds_base wave-uniform).
// M0 carries the LDS base; it is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ void dma16(unsigned lds_base, unsigned voff, const u32x4 &rs) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void aload64(u32x2 &dst, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}

// counted stores are asm as well: hipcc merged two identical placeholder stores into one, which shifts every
// hand-counted wait by one
__device__ __forceinline__ void astore64(const u32x2 &v, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void astore128(const u32x4 &v, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 2" : : "v"(v), "v"(voff), "s"(rs) : "memory");   // (data registers are read after issue)
}

// phase time stamps of one workgroup (DODA_DMA_DBG bit 7; tools/dmastamps.py): [wave 0 | wave 5][iteration][phase]
__device__ unsigned long long g_dm_stamps[2 * 16 * 8];
__device__ __forceinline__ void dm_stamp(int dbg, int wid, int j, int phase) {
    if ((dbg & 128) && blockIdx.x == 8 && (wid == 0 || wid == 5) && j < 16 && (threadIdx.x & 63) == 0)
        g_dm_stamps[((wid ? 1 : 0) * 16 + j) * 8 + phase] = __builtin_amdgcn_s_memtime();
}

struct DmList { u32x4 rid; };                 // list entries of this lane's four row pieces (one 16-byte load)
struct DmEpi { u32x2 res[2], bnx[2]; };       // epilogue operands: four bf16 channels of two rows

__device__ __forceinline__ void aload128(u32x4 &dst, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}

// wait until at most N vector-memory operations are outstanding (no register operands: see dm_tie)
template <int N>
__device__ __forceinline__ void dm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// tie the registers the landed loads wrote to this point: their uses must not be scheduled above the wait that
// precedes this statement (volatile asm statements keep their order)
__device__ __forceinline__ void dm_tie(DmList &l, DmEpi &e) {
    asm volatile("" : "+v"(l.rid), "+v"(e.res[0]), "+v"(e.res[1]), "+v"(e.bnx[0]), "+v"(e.bnx[1]) : : "memory");
}

__device__ __forceinline__ void dm_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 dm_unpack(const u32x2 &v) {
    return (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                   __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}

// Vector-memory instructions of one wave and iteration i, in issue order (all counted by vmcnt, which retires in order):
//     D(i+2)  DM_NG  DMA pieces of tile i+2 (rows of list(i+2), index strip), spread over the multiply loop of tile i
//     LE_i    1+N_E  list(i+4) and the epilogue operands of tile i+2            -> registers
//     S(i)    N_S    stores of tile i
// At the top of iteration i the wave needs D(i), list(i+2) and operands(i) — all issued in iteration i-2 — and may
// leave everything younger in flight: S(i-2), D(i+1), LE_{i-1}, S(i-1)  ->  s_waitcnt vmcnt(2 N_S + DM_NG + 1 + N_E).
template <bool STATS>
__global__ __launch_bounds__(512) void conv_dma16(const void *__restrict__ x, unsigned x_bytes,
                                                  const void *__restrict__ wp, unsigned wp_bytes,
                                                  const int32_t *__restrict__ tbl, int ld, int n_out,
                                                  const TileBookView tb, void *__restrict__ y, unsigned y_bytes,
                                                  const void *__restrict__ res, const EpiArgs ep, const int dbg) {
    constexpr int NU = (TB_K + 1) / 2;
    constexpr int N_E = STATS ? 4 : 2;                 // epilogue operand loads per wave and tile
    constexpr int N_S = STATS ? 4 : 2;                 // stores per wave and tile
    constexpr int N_WAIT = 2 * N_S + DM_NG + 1 + N_E;
    __shared__ __attribute__((aligned(16))) unsigned char smem[DM_NBUF * DM_BUF_BYTES];
    __shared__ f32x4 sred[STATS ? DM_WAVES : 1][2][4];

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const u32x4 rs_x = dm_rsrc(x, x_bytes);
    const u32x4 rs_ul = dm_rsrc(tb.ulist, (unsigned)tb.nt * (unsigned)TB_UMAX * 4u);
    const u32x4 rs_li = dm_rsrc(tb.lidx, (unsigned)tb.nt * (unsigned)(TB_K * TB_T * 2));
    const u32x4 rs_res = dm_rsrc(res, res ? y_bytes : 0u);                 // absent operand: every offset out of range
    const u32x4 rs_bnx = dm_rsrc(ep.bn_x, (STATS && ep.bn_x) ? y_bytes : 0u);
    const u32x4 rs_y = dm_rsrc(y, y_bytes);
    const u32x4 rs_st = dm_rsrc(ep.stats, STATS ? (unsigned)tb.nt * 2u * 16u * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)TB_K * (unsigned)ld * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xb = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);

    // persistent schedule: XCD (blockIdx & 7) owns one contiguous range of tiles, its L workgroups stride it
    const int L = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = tb.nt >> 3, rn = tb.nt & 7;
    const int lo = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    const int cnt = qn + (xcd < rn ? 1 : 0);
    const int nt_w = slot < cnt ? (cnt - slot + L - 1) / L : 0;      // tiles of this workgroup
    if (nt_w == 0) return;
    auto tile_of = [&](int j) { return lo + slot + j * L; };

    // weights: pair-packed fragments [o][32 slots] x 16 B, unit u = offsets (2u, 2u + 1); offset 27 lies past the
    // buffer (zeros).  Held in registers for the workgroup's whole life.
    u32x4 wr[NU];
    {
        const unsigned lane_w = (unsigned)(g >> 1) * 512u + (unsigned)((g & 1) * 16 + i) * 16u;
#pragma unroll
        for (int u = 0; u < NU; ++u) wr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)u * 1024u + lane_w, 0, 0);
    }
    // per-channel BatchNorm vectors of the data-grad statistics (this lane's four channels): loaded ONCE — a
    // compiler-tracked load inside the loop would make hipcc wait vmcnt(0) there and drain the pipeline
    f32x4 bn_mu = {0.f, 0.f, 0.f, 0.f}, bn_is = bn_mu, bn_ga = bn_mu, bn_be = bn_mu;
    if constexpr (STATS) {
        if (ep.bn_x) {
            bn_mu = *reinterpret_cast<const f32x4 *>(ep.bn_mean + 4 * g);
            bn_is = *reinterpret_cast<const f32x4 *>(ep.bn_invstd + 4 * g);
            bn_ga = *reinterpret_cast<const f32x4 *>(ep.bn_gamma + 4 * g);
            bn_be = *reinterpret_cast<const f32x4 *>(ep.bn_beta + 4 * g);
        }
    }
    // tie every compiler-tracked load to this point: hipcc places its wait HERE, before the hand-counted region
#pragma unroll
    for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(wr[u]));
    asm volatile("" : "+v"(bn_mu), "+v"(bn_is), "+v"(bn_ga), "+v"(bn_be));
    // the zero row of every buffer
    if (tid < 2 * DM_NBUF) reinterpret_cast<u32x4 *>(smem + (tid >> 1) * DM_BUF_BYTES)[tid & 1] = (u32x4){0u, 0u, 0u, 0u};

    const unsigned smem_base = (unsigned)(uintptr_t)smem;
    // Row piece k of this lane: list entry e = (k*8 + wid)*32 + (lane >> 1), stored so that the lane's four entries
    // are one 16-byte load (tb_upos).  Its half of the row is swizzled with bit 3 of e (= bit 4 of the lane): the
    // 16 lanes of an LDS read phase ask for the same half of 16 different rows, which un-swizzled 32-byte rows put
    // on HALF of the banks (>= 2-way conflicts by construction).
    const unsigned src_half = (unsigned)((lane & 1) ^ ((lane >> 4) & 1)) * 16u;
    auto issue_list = [&](int j, DmList &l) {
        const bool ok = j < nt_w && !(dbg & 8);
        aload128(l.rid, ok ? (unsigned)tile_of(j) * (unsigned)(TB_UMAX * 4) + (unsigned)(wid * 32 + (lane >> 1)) * 16u : OOB, rs_ul);
    };
    auto issue_epi = [&](int j, DmEpi &e) {
        const bool ok = j < nt_w;
        const unsigned row0 = (unsigned)tile_of(j) * (unsigned)TB_T + (unsigned)wid * 32u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned t = row0 + (unsigned)(s * 16 + i);
            const unsigned voff = (ok && t < (unsigned)n_out) ? (t * 16u + 4u * (unsigned)g) * 2u : OOB;
            aload64(e.res[s], voff, rs_res);
            if constexpr (STATS) aload64(e.bnx[s], voff, rs_bnx);
        }
    };
    // DMA piece n of tile j into buffer j % 3: n < 4 rows of the list, n >= 4 the index strip
    auto issue_piece = [&](int j, int n, const DmList &l) {
        const bool ok = j < nt_w;
        const unsigned buf = smem_base + (unsigned)(j % DM_NBUF) * (unsigned)DM_BUF_BYTES;
        if (n < DM_ROW_PIECES) {
            // an absent entry is negative: its row offset is out of range and lands as zeros
            const unsigned voff = (ok && !(dbg & 2)) ? l.rid[n] * 32u + src_half : OOB;
            dma16(buf + 32u + (unsigned)((n * DM_WAVES + wid) * 1024), voff, rs_x);
        } else {
            const int k = n - DM_ROW_PIECES;
            const unsigned p = (unsigned)((k * DM_WAVES + wid) * 64 + lane);
            dma16(buf + (unsigned)DM_ROWS_BYTES + (unsigned)((k * DM_WAVES + wid) * 1024),
                  (ok && !(dbg & 4)) ? (unsigned)tile_of(j) * (unsigned)(TB_K * TB_T * 2) + p * 16u : OOB, rs_li);
        }
    };

    // ---- iteration j: multiply tile j out of buffer j % 3 while the DMA of tile j+2 (list lcur) is issued between
    // the units; then list(j+4) -> lnew, operands(j+2) -> enew; epilogue of tile j with operands ecur; stores ----
    auto iteration = [&](int j, bool overflow, const DmList &lcur, DmList &lnew, const DmEpi &ecur, DmEpi &enew) {
        const int tile = tile_of(j), t0 = tile * TB_T, row0 = t0 + wid * 32;
        const unsigned char *rows_s = smem + (j % DM_NBUF) * DM_BUF_BYTES;
        const unsigned short *lidx_s = reinterpret_cast<const unsigned short *>(rows_s + DM_ROWS_BYTES);
        const unsigned half = (unsigned)(g & 1);
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (dbg & 1) {
#pragma unroll
            for (int n = 0; n < DM_NG; ++n) issue_piece(j + 2, n, lcur);
        } else if (!overflow) {
            // the lane's two local indices of unit u (subtiles 2 (wid & 1), 2 (wid & 1) + 1 of its 64-row group)
            const unsigned short *my = lidx_s + (wid >> 1) * 64 + i * 4 + (wid & 1) * 2;
            auto loadl = [&](int u) {
                const int osel = 2 * u + (g >> 1);
                unsigned v = 0u;   // offset 27 of the last pair: the zero row
                if (osel < TB_K) v = *reinterpret_cast<const unsigned *>(my + osel * TB_T);
                return v;
            };
            // local index l (1 + list entry): byte l*32 + (half ^ bit 3 of the entry) * 16
            auto addr = [&](unsigned l) { return l * 32u + ((half ^ (((l - 1u) >> 3) & 1u)) << 4); };
            auto fetch = [&](unsigned l, u32x4 (&xa)[2]) {
                xa[0] = *reinterpret_cast<const u32x4 *>(rows_s + addr(l & 0xffffu));
                xa[1] = *reinterpret_cast<const u32x4 *>(rows_s + addr(l >> 16));
            };
            unsigned lr[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) lr[u] = loadl(u);
            u32x4 xa[4][2];
            fetch(lr[0], xa[0]);
            fetch(lr[1], xa[1]);
            fetch(lr[2], xa[2]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 3 < NU) fetch(lr[u + 3], xa[(u + 3) & 3]);
                // one DMA piece of tile j+2 every other unit: the texture path works on them under the LDS reads
                // and MFMAs of this tile instead of in a phase of its own
                if ((u & 1) == 0 && u / 2 < DM_NG) issue_piece(j + 2, u / 2, lcur);
                __builtin_amdgcn_sched_barrier(0);
                mma_bf16_k32(acc[0], wr[u], xa[u & 3][0]);
                mma_bf16_k32(acc[1], wr[u], xa[u & 3][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int n = 0; n < DM_NG; ++n) issue_piece(j + 2, n, lcur);
            // overflow tile (more distinct rows than the list holds): operands gathered from global memory through
            // the dense table.  Rare (none at 2 cm); the compiler's own waits drain the pipeline here, which is only slow.
#pragma unroll
            for (int u = 0; u < NU; ++u) {   // (unrolled: a run-time index into wr[] would move the fragments to scratch)
                const int osel = 2 * u + (g >> 1);
                u32x4 xa[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int t = row0 + s * 16 + i;
                    const unsigned voff = (osel < TB_K && t < n_out) ? ((unsigned)osel * (unsigned)ld + (unsigned)t) * 4u : OOB;
                    const unsigned go = __builtin_amdgcn_raw_buffer_load_b32(rs_t, voff, 0, 0);
                    const bool present = osel < TB_K && t < n_out && (int)go >= 0;
                    xa[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_xb, present ? go * 32u + half * 16u : OOB, 0, 0);
                }
                mma_bf16_k32(acc[0], wr[u], xa[0]);
                mma_bf16_k32(acc[1], wr[u], xa[1]);
            }
        }
        dm_stamp(dbg, wid, j, 3);
        issue_list(j + 4, lnew);
        issue_epi(j + 2, enew);
        dm_stamp(dbg, wid, j, 4);
        if (dbg & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (ablation: no overlap across iterations)
        // ---- epilogue: lane (i, g) holds output channels 4g .. 4g+3 of rows row0 + 16 s + i ----
        const unsigned col = 4u * (unsigned)g;
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned t = (unsigned)(row0 + s * 16 + i);
            const unsigned voff = (t < (unsigned)n_out && !(dbg & 16)) ? (t * 16u + col) * 2u : OOB;
            f32x4 a = acc[s];
            a += dm_unpack(ecur.res[s]);            // (no residual: the load was out of range: zeros)
            u32x2 packed_out;
            packed_out[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
            packed_out[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
            if constexpr (STATS) {
                f32x4 v = dm_unpack(packed_out);    // y as stored
                if (t >= (unsigned)n_out) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ep.bn_x) {
                    const f32x4 xr = dm_unpack(ecur.bnx[s]);
                    const f32x4 xh = (xr - bn_mu) * bn_is;
                    if (ep.bn_relu) {
                        const f32x4 yv = xh * bn_ga + bn_be;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = yv[q] > 0.f ? v[q] : 0.f;
                    }
                    st1 += v;
                    st2 += v * xh;
                } else {
                    st1 += v;
                    st2 += v * v;
                }
            }
            astore64(packed_out, voff, rs_y);
        }
        if constexpr (STATS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { st1[q] = row_sum16(st1[q]); st2[q] = row_sum16(st2[q]); }
            if (i == 15) { sred[wid][0][g] = st1; sred[wid][1][g] = st2; }
            dm_barrier();
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
            if (wid == 0 && i == 15) {
                a1 = ((sred[0][0][g] + sred[1][0][g]) + (sred[2][0][g] + sred[3][0][g])) +
                     ((sred[4][0][g] + sred[5][0][g]) + (sred[6][0][g] + sred[7][0][g]));
                a2 = ((sred[0][1][g] + sred[1][1][g]) + (sred[2][1][g] + sred[3][1][g])) +
                     ((sred[4][1][g] + sred[5][1][g]) + (sred[6][1][g] + sred[7][1][g]));
            }
            // every wave issues the two stores (constant vmcnt bookkeeping); only wave 0's lanes 15, 31, 47, 63 land
            const unsigned so = (wid == 0 && i == 15) ? ((unsigned)tile * 32u + col) * 4u : OOB;
            astore128(__builtin_bit_cast(u32x4, a1), so, rs_st);
            astore128(__builtin_bit_cast(u32x4, a2), so + (so == OOB ? 0u : 64u), rs_st);
        }
        dm_stamp(dbg, wid, j, 5);
    };
    auto dummy_stores = [&]() {
#pragma unroll
        for (int k = 0; k < N_S; ++k) astore64((u32x2){0u, 0u}, OOB, rs_y);
    };
    // a tile without a list (more than TB_UMAX distinct rows) carries -2 in every entry
    auto no_list = [&](const DmList &l) { return __builtin_amdgcn_readfirstlane((int)l.rid[0]) == -2 && !(dbg & 64); };

    // ---- prologue: the queue is given the shape of two past iterations (with stores that land nowhere), so that
    // ONE wait constant serves every iteration ----
    DmList la, lb, lc;
    DmEpi ea, eb, ec;
    if constexpr (!STATS) ea.bnx[0] = ea.bnx[1] = eb.bnx[0] = eb.bnx[1] = ec.bnx[0] = ec.bnx[1] = (u32x2){0u, 0u};
    ec.res[0] = ec.res[1] = (u32x2){0u, 0u};
    if constexpr (STATS) ec.bnx[0] = ec.bnx[1] = (u32x2){0u, 0u};
    issue_list(0, la);
    issue_list(1, lb);
    dm_wait<0>();
    dm_tie(la, ec);
    dm_tie(lb, ec);
    dm_barrier();                 // the zero rows are written
    bool ov0 = no_list(la), ov1 = no_list(lb);
#pragma unroll
    for (int n = 0; n < DM_NG; ++n) issue_piece(0, n, la);      // "iteration -2": D(0), list(2), operands(0), stores
    issue_list(2, lc);
    issue_epi(0, ea);
    dummy_stores();
#pragma unroll
    for (int n = 0; n < DM_NG; ++n) issue_piece(1, n, lb);      // "iteration -1": D(1), list(3), operands(1), stores
    issue_list(3, la);
    issue_epi(1, eb);
    dummy_stores();
    // Register sets rotate with period three (lists: the DMA of tile j+2 reads list(j+2) during iteration j, list(j+3)
    // is landing, list(j+4) is requested; operands likewise), written out three times so that no register holding a
    // load in flight is ever copied.
    for (int j = 0; j < nt_w; j += 3) {
        dm_stamp(dbg, wid, j, 0);
        dm_wait<N_WAIT>();
        dm_stamp(dbg, wid, j, 1);
        dm_tie(lc, ea);                                  // list(j+2), operands(j)
        dm_barrier();
        dm_stamp(dbg, wid, j, 2);
        { const bool ov2 = no_list(lc); iteration(j, ov0, lc, lb, ea, ec); ov0 = ov1; ov1 = ov2; }
        if (j + 1 >= nt_w) break;
        dm_stamp(dbg, wid, j + 1, 0);
        dm_wait<N_WAIT>();
        dm_stamp(dbg, wid, j + 1, 1);
        dm_tie(la, eb);                                  // list(j+3), operands(j+1)
        dm_barrier();
        dm_stamp(dbg, wid, j + 1, 2);
        { const bool ov2 = no_list(la); iteration(j + 1, ov0, la, lc, eb, ea); ov0 = ov1; ov1 = ov2; }
        if (j + 2 >= nt_w) break;
        dm_stamp(dbg, wid, j + 2, 0);
        dm_wait<N_WAIT>();
        dm_stamp(dbg, wid, j + 2, 1);
        dm_tie(lb, ec);                                  // list(j+4), operands(j+2)
        dm_barrier();
        dm_stamp(dbg, wid, j + 2, 2);
        { const bool ov2 = no_list(lb); iteration(j + 2, ov0, lb, la, ec, eb); ov0 = ov1; ov1 = ov2; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // loads still in flight target registers and LDS of this workgroup
}

// OFF by default: measured on MI355X (gpurun_out / DESIGN.md §9) the pipeline equals conv_tile on the plain kernel (29 us cold)
// and loses with the statistics epilogue (43 against 33 us); doda_spconv_set_dma_kernel(1) / DODA_DMA=1 switch it on
bool g_use_dma = getenv("DODA_DMA") && getenv("DODA_DMA")[0] == '1';

}  // namespace

namespace doda_dma {

bool enabled() { return g_use_dma; }
void set_enabled(bool on) { g_use_dma = on; }

// bf16 16 -> 16 over a tilebook; *n_part receives the number of statistics rows (one per tile)
int launch_conv16(const void *x, unsigned xb, const void *wp, unsigned wpb, const int32_t *tbl, int ld, int n_out,
                  const void *tilebook, void *y, unsigned yb, const void *res, const EpiArgs &ep, int *n_part, hipStream_t s) {
    const TileBookView tb = tilebook_view(const_cast<void *>(tilebook), n_out);
    int groups = (tb.nt + 7) / 8 * 8;
    if (groups > 256) groups = 256;       // one workgroup per CU
    if (n_part) *n_part = tb.nt;
    static const int dbg = getenv("DODA_DMA_DBG") ? atoi(getenv("DODA_DMA_DBG")) : 0;   // ablation switches (measurements only)
    if (ep.stats)
        hipLaunchKernelGGL((conv_dma16<true>), dim3(groups), dim3(512), 0, s, x, xb, wp, wpb, tbl, ld, n_out, tb, y, yb, res, ep, dbg);
    else
        hipLaunchKernelGGL((conv_dma16<false>), dim3(groups), dim3(512), 0, s, x, xb, wp, wpb, tbl, ld, n_out, tb, y, yb, res, ep, dbg);
    return doda_check_launch();
}

}  // namespace doda_dma

extern "C" void doda_spconv_set_dma_kernel(int32_t on) { doda_dma::set_enabled(on != 0); }

// measurement aid: the phase time stamps DODA_DMA_DBG=128 makes workgroup 8 of conv_dma16 record (2 x 16 x 8 uint64)
extern "C" int doda_debug_dma_stamps(unsigned long long *dst_h) {
    return hipMemcpyFromSymbol(dst_h, HIP_SYMBOL(g_dm_stamps), sizeof(unsigned long long) * 2 * 16 * 8) == hipSuccess ? DODA_OK
                                                                                                                   : DODA_ERR_LAUNCH;
}
