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
constexpr int DM_NL = DM_ROW_PIECES + 1;              // list registers + the tile's count
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
__device__ __forceinline__ void aload32(unsigned &dst, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void aload64(u32x2 &dst, unsigned voff, const u32x4 &rs) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rs) : "memory");
}

struct DmList { unsigned rid[DM_ROW_PIECES]; unsigned ucount; };   // list entries of this lane's DMA pieces
struct DmEpi { u32x2 res[2], bnx[2]; };                            // epilogue operands: four bf16 channels of two rows

// wait until at most N vector-memory operations are outstanding, and tie the registers the landed loads wrote to
// this point (their uses must not be scheduled above the wait)
template <int N>
__device__ __forceinline__ void dm_wait(DmList &l, DmEpi &e) {
    asm volatile("s_waitcnt vmcnt(%9)"
                 : "+v"(l.rid[0]), "+v"(l.rid[1]), "+v"(l.rid[2]), "+v"(l.rid[3]), "+v"(l.ucount),
                   "+v"(e.res[0]), "+v"(e.res[1]), "+v"(e.bnx[0]), "+v"(e.bnx[1])
                 : "n"(N) : "memory");
}
static_assert(DM_ROW_PIECES == 4, "dm_wait names four list registers");

__device__ __forceinline__ void dm_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x4 dm_unpack(const u32x2 &v) {
    return (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                   __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}

template <bool STATS>
__global__ __launch_bounds__(512) void conv_dma16(const void *__restrict__ x, unsigned x_bytes,
                                                  const void *__restrict__ wp, unsigned wp_bytes,
                                                  const int32_t *__restrict__ tbl, int ld, int n_out,
                                                  const TileBookView tb, void *__restrict__ y, unsigned y_bytes,
                                                  const void *__restrict__ res, const EpiArgs ep) {
    constexpr int NU = (TB_K + 1) / 2;
    constexpr int N_E = STATS ? 4 : 2;                 // epilogue operand loads per wave and tile
    constexpr int N_S = STATS ? 4 : 2;                 // stores per wave and tile
    constexpr int N_LE = DM_NL + N_E;
    __shared__ __attribute__((aligned(16))) unsigned char smem[DM_NBUF * DM_BUF_BYTES];
    __shared__ f32x4 sred[STATS ? DM_WAVES : 1][2][4];

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const u32x4 rs_x = dm_rsrc(x, x_bytes);
    const u32x4 rs_ul = dm_rsrc(tb.ulist, (unsigned)tb.nt * (unsigned)TB_UMAX * 4u);
    const u32x4 rs_li = dm_rsrc(tb.lidx, (unsigned)tb.nt * (unsigned)(TB_K * TB_T * 2));
    const u32x4 rs_uc = dm_rsrc(tb.ucount, (unsigned)tb.nt * 4u);
    const u32x4 rs_res = dm_rsrc(res, res ? y_bytes : 0u);                 // absent operand: every offset out of range
    const u32x4 rs_bnx = dm_rsrc(ep.bn_x, (STATS && ep.bn_x) ? y_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void *)ep.stats, 0,
                                                                           STATS ? (unsigned)tb.nt * 2u * 16u * 4u : 0u, 0x00020000);
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
    // this lane's DMA pieces: row piece k -> list entry ((k * 8 + wid) * 64 + lane) >> 1, half lane & 1
    auto issue_list = [&](int j, DmList &l) {
        const bool ok = j < nt_w;
        const unsigned t = (unsigned)tile_of(j);
#pragma unroll
        for (int k = 0; k < DM_ROW_PIECES; ++k) {
            const unsigned e = (unsigned)(((k * DM_WAVES + wid) * 64 + lane) >> 1);
            aload32(l.rid[k], ok ? (t * (unsigned)TB_UMAX + e) * 4u : OOB, rs_ul);
        }
        aload32(l.ucount, ok ? t * 4u : OOB, rs_uc);
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
    auto issue_dma = [&](int j, const DmList &l) {
        const bool ok = j < nt_w;
        const unsigned buf = smem_base + (unsigned)(j % DM_NBUF) * (unsigned)DM_BUF_BYTES;
#pragma unroll
        for (int k = 0; k < DM_ROW_PIECES; ++k) {
            // an absent entry is -1: its row offset is out of range and lands as zeros
            const unsigned voff = ok ? l.rid[k] * 32u + (unsigned)(lane & 1) * 16u : OOB;
            dma16(buf + 32u + (unsigned)((k * DM_WAVES + wid) * 1024), voff, rs_x);
        }
        const unsigned t = (unsigned)tile_of(j);
#pragma unroll
        for (int k = 0; k < DM_LIDX_PIECES; ++k) {
            const unsigned p = (unsigned)((k * DM_WAVES + wid) * 64 + lane);
            dma16(buf + (unsigned)DM_ROWS_BYTES + (unsigned)((k * DM_WAVES + wid) * 1024),
                  ok ? t * (unsigned)(TB_K * TB_T * 2) + p * 16u : OOB, rs_li);
        }
    };

    // ---- one tile: multiply out of buffer j % 3, epilogue with operands `e`, stores ----
    auto compute = [&](int j, unsigned ucount, const DmEpi &e) {
        const int tile = tile_of(j), t0 = tile * TB_T, row0 = t0 + wid * 32;
        const unsigned char *rows_s = smem + (j % DM_NBUF) * DM_BUF_BYTES;
        const unsigned short *lidx_s = reinterpret_cast<const unsigned short *>(rows_s + DM_ROWS_BYTES);
        const unsigned half = (unsigned)(g & 1) * 16u;
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (ucount <= (unsigned)DM_CAP) {
            // the lane's two local indices of unit u (subtiles 2 (wid & 1), 2 (wid & 1) + 1 of its 64-row group)
            const unsigned short *my = lidx_s + (wid >> 1) * 64 + i * 4 + (wid & 1) * 2;
            auto loadl = [&](int u) {
                const int osel = 2 * u + (g >> 1);
                unsigned v = 0u;   // offset 27 of the last pair: the zero row
                if (osel < TB_K) v = *reinterpret_cast<const unsigned *>(my + osel * TB_T);
                return v;
            };
            auto fetch = [&](unsigned l, u32x4 (&xa)[2]) {
                xa[0] = *reinterpret_cast<const u32x4 *>(rows_s + (l & 0xffffu) * 32u + half);
                xa[1] = *reinterpret_cast<const u32x4 *>(rows_s + (l >> 16) * 32u + half);
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
                __builtin_amdgcn_sched_barrier(0);
                mma_bf16_k32(acc[0], wr[u], xa[u & 3][0]);
                mma_bf16_k32(acc[1], wr[u], xa[u & 3][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // overflow tile (more distinct rows than the buffers hold): operands gathered from global memory through
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
                    xa[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_xb, present ? go * 32u + half : OOB, 0, 0);
                }
                mma_bf16_k32(acc[0], wr[u], xa[0]);
                mma_bf16_k32(acc[1], wr[u], xa[1]);
            }
        }
        // ---- epilogue: lane (i, g) holds output channels 4g .. 4g+3 of rows row0 + 16 s + i ----
        const unsigned col = 4u * (unsigned)g;
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned t = (unsigned)(row0 + s * 16 + i);
            const unsigned voff = t < (unsigned)n_out ? (t * 16u + col) * 2u : OOB;
            f32x4 a = acc[s];
            a += dm_unpack(e.res[s]);               // (no residual: the load was out of range: zeros)
            u32x2 packed_out;
            packed_out[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
            packed_out[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
            if constexpr (STATS) {
                f32x4 v = dm_unpack(packed_out);    // y as stored
                if (t >= (unsigned)n_out) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ep.bn_x) {
                    const f32x4 xr = dm_unpack(e.bnx[s]);
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
            __builtin_amdgcn_raw_buffer_store_b64(packed_out, rs_y, voff, 0, 0);
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
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a1), rs_st, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2), rs_st, so + (so == OOB ? 0u : 64u), 0, 0);
        }
    };

    // ---- prologue: lists of tiles 0 .. 2 and operands of tile 0, then the DMA of tiles 0 and 1 ----
    DmList la, lb, lc;
    DmEpi ea, eb;
    eb.res[0] = eb.res[1] = eb.bnx[0] = eb.bnx[1] = (u32x2){0u, 0u};
    if constexpr (!STATS) ea.bnx[0] = ea.bnx[1] = (u32x2){0u, 0u};
    issue_list(0, la);
    issue_list(1, lb);
    issue_list(2, lc);
    issue_epi(0, ea);
    dm_wait<0>(la, ea);
    dm_wait<0>(lb, eb);
    dm_wait<0>(lc, eb);
    dm_barrier();                 // the zero rows are written
    issue_dma(0, la);
    issue_dma(1, lb);
    // N_S stores that land nowhere: the queue of iteration 0 then has the shape of every later one (.. DMA(j+1),
    // stores(j-1), list(j+3) ..) and ONE wait constant serves all iterations — two wait statements on a branch made
    // hipcc merge their register operands through copies placed BEFORE the wait, i.e. copies of loads in flight
#pragma unroll
    for (int k = 0; k < N_S; ++k) __builtin_amdgcn_raw_buffer_store_b64((u32x2){0u, 0u}, rs_y, OOB, 0, 0);
    // From here on: list registers rotate a <- c, b <- a, c <- b every iteration, written out three times so that no
    // register holding a load in flight is ever copied (a copy would have to wait for it); the epilogue operands
    // alternate between two sets the same way.
    // step(j, lnext, lnew, ecur, enew):  lnext = list(j+2) (landed below), lnew receives list(j+3)
    auto step = [&](int j, DmList &lnext, DmList &lnew, DmEpi &ecur, DmEpi &enew) {
        issue_list(j + 3, lnew);                 // a
        issue_epi(j + 1, enew);
        dm_wait<DM_NG + N_S + N_LE>(lnext, ecur);                  // b
        dm_barrier();                            // c
        issue_dma(j + 2, lnext);                 // d
    };
    // the count of tile j travels with list(j): keep the three most recent ones in scalars
    unsigned uc0 = (unsigned)__builtin_amdgcn_readfirstlane((int)la.ucount);
    unsigned uc1 = (unsigned)__builtin_amdgcn_readfirstlane((int)lb.ucount);
    for (int j = 0; j < nt_w; j += 6) {
        // iteration j: list(j+2) = lc, new list -> la (list(j) = la is dead: its DMA was issued)
        step(j, lc, la, ea, eb);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)lc.ucount); compute(j, uc0, ea); uc0 = uc1; uc1 = uc2; }
        if (j + 1 >= nt_w) break;
        step(j + 1, la, lb, eb, ea);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)la.ucount); compute(j + 1, uc0, eb); uc0 = uc1; uc1 = uc2; }
        if (j + 2 >= nt_w) break;
        step(j + 2, lb, lc, ea, eb);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)lb.ucount); compute(j + 2, uc0, ea); uc0 = uc1; uc1 = uc2; }
        if (j + 3 >= nt_w) break;
        step(j + 3, lc, la, eb, ea);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)lc.ucount); compute(j + 3, uc0, eb); uc0 = uc1; uc1 = uc2; }
        if (j + 4 >= nt_w) break;
        step(j + 4, la, lb, ea, eb);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)la.ucount); compute(j + 4, uc0, ea); uc0 = uc1; uc1 = uc2; }
        if (j + 5 >= nt_w) break;
        step(j + 5, lb, lc, eb, ea);
        { const unsigned uc2 = (unsigned)__builtin_amdgcn_readfirstlane((int)lb.ucount); compute(j + 5, uc0, eb); uc0 = uc1; uc1 = uc2; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // loads still in flight target registers and LDS of this workgroup
}

bool g_use_dma = true;

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
    if (ep.stats)
        hipLaunchKernelGGL((conv_dma16<true>), dim3(groups), dim3(512), 0, s, x, xb, wp, wpb, tbl, ld, n_out, tb, y, yb, res, ep);
    else
        hipLaunchKernelGGL((conv_dma16<false>), dim3(groups), dim3(512), 0, s, x, xb, wp, wpb, tbl, ld, n_out, tb, y, yb, res, ep);
    return doda_check_launch();
}

}  // namespace doda_dma

extern "C" void doda_spconv_set_dma_kernel(int32_t on) { doda_dma::set_enabled(on != 0); }
