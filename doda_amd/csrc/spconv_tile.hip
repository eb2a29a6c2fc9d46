// LDS-staged sparse-convolution kernel over a tilebook (tilebook.hpp / tilebook.hip): conv_tile (forward and data
// gradient of the bf16 16- / 32-channel and fp32 16-channel SubM layers).  Dispatched from spconv_gather.hip
// (run_gather) through doda_tile::launch_conv_tile;
// replaces spconv v1.2's indice_conv / indice_conv_backward data path (reference call sites
// model/unet_block.py:26,29,48) for the layers whose rulebook carries a tilebook.
#include "common.hpp"
#include "tilebook.hpp"
#include "spconv_common.hpp"

#ifdef DODA_TILE_STAMPS
// Measurement build only (tools/tilestamps.py compiles its own copy of the library with -DDODA_TILE_STAMPS): wave 0 of every
// workgroup records the 100 MHz wall clock at six points of each of its first eight tiles.
__device__ unsigned long long doda_tile_stamp_buf[768 * 8 * 8];
extern "C" int doda_debug_tile_stamps(void *host_out, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(doda_tile_stamp_buf), bytes);
}
#define TILE_STAMP(k)                                                                                      \
    do {                                                                                                   \
        if (tid0 == 0 && stamp_it < 8) doda_tile_stamp_buf[(blockIdx.x * 8 + stamp_it) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define TILE_STAMP(k) do { } while (0)
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// LDS-staged tile kernel (bf16, 16 or 32 input channels, K = 27 SubM: the layers of the two finest U-Net
// levels and the north-star gate).  conv_fast is paced by the texture path: 28 gather instructions per 32
// output rows at 37 % lane use, each costing >= 16 cycles of the CU's L1 whatever its EXEC mask (DESIGN.md
// §4).  Here a workgroup owns a tile of TB_T = 256 consecutive output rows whose neighbourhood the tilebook
// (tilebook.hpp) lists as ~2.2 x 256 DISTINCT input rows:
//   phase A  every distinct row is loaded once, consecutive lanes covering consecutive list entries (runs
//            of consecutive rows -> whole 128-byte lines), and parked in LDS next to the tile's local-index
//            strip (ten planes of packed 10-bit local indices, tilebook.hpp);
//   phase B  per unit (a pair of offsets x 16 channels, or one offset x 32 channels) the lane's four local indices (one per
//            16-row subtile) come out of a 16-byte LDS read that serves three units, four 16-byte LDS reads fetch
//            the operand rows (an absent neighbour is the shared zero row: same address in every lane, a
//            broadcast), four MFMAs accumulate.  The only vector-memory instruction in the loop is the
//            streamed weight fragment (L1-resident).
// Vector-memory instructions per wave and tile (16 channels): 8 rows + 2 list + 3 strip + 14 weights + 4 epilogue
// operands + 4 stores = 35 against 96 per 64 rows.  A tile whose neighbourhood exceeds the kernel's capacity (1023 rows of 32
// bytes — 10-bit local indices —, 960 of 64 bytes: 0.2 % of the level-1 tiles of a 2 cm scene, most of a 1 cm scene's — such batches keep plain tables at that level) takes
// the same loop with the operands gathered from global memory through the dense table.
// Same arithmetic as conv_fast up to the order in which offsets are paired (fixed (2u, 2u+1) here, pairs of
// ACTIVE offsets there): fp32 accumulation, one bf16 rounding at the store.
// ---------------------------------------------------------------------------------------------
// Store epilogue of the tile kernels: conv_fast's epilogue for bf16 features, one channel block, four waves
// holding four row ranges of the workgroup's tile (residual add, single bf16 rounding, BatchNorm statistics
// as one partial row per workgroup — see EpiArgs).  The two operands it reads from global memory — the
// residual rows and the BatchNorm input rows — are requested by epi_prefetch BEFORE the multiply phase, so
// the store does not wait for another memory round trip at the end of a tile's dependency chain.
template <bool OUT32> struct EpiPre { u32x4 res[4], bnx[4]; };   // four channels of a row: fp32 ...
template <> struct EpiPre<false> { u32x2 res[4], bnx[4]; };        // ... or bf16

// ALWAYS: issue every load whatever the call's operands (an absent operand: out-of-range offset, zeros, no traffic) — the
// pipelined kernel needs the same instruction sequence on every path (conv_tile16)
template <int S, bool OUT32, bool STATS, bool ALWAYS = false>
__device__ __forceinline__ void epi_prefetch(EpiPre<OUT32> &pre, int row0, int i, int g, int nb0, int nc, int n_out,
                                             unsigned y_bytes, const void *__restrict__ res, const EpiArgs &ep) {
    constexpr unsigned OSZ = OUT32 ? 4u : 2u;
    const unsigned col = (unsigned)(nb0 * 16 + 4 * g);
    // (ABI 12: the residual and the BatchNorm input may be column slices of wider matrices — row strides res_ld / bnx_ld, 0 = dense;
    // a dense operand keeps the caller's y_bytes bound)
    const unsigned rl = ep.res_ld ? ep.res_ld : (unsigned)nc, bl = ep.bnx_ld ? ep.bnx_ld : (unsigned)nc;
    const unsigned dense_bytes = (unsigned)n_out * (unsigned)nc * OSZ;
    const unsigned r_bytes = ep.res_ld ? ((unsigned)(n_out - 1) * rl + (unsigned)nc) * OSZ : (ep.y_ld ? dense_bytes : y_bytes);
    const unsigned b_bytes = ep.bnx_ld ? ((unsigned)(n_out - 1) * bl + (unsigned)nc) * OSZ : (ep.y_ld ? dense_bytes : y_bytes);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)res, 0, r_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)ep.bn_x, 0, b_bytes, 0x00020000);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned t = (unsigned)(row0 + s * 16 + i);
        const bool in_range = t < (unsigned)n_out && col < (unsigned)nc;
        const unsigned voff = in_range ? (t * rl + col) * OSZ : OOB;
        const unsigned voff_b = in_range ? (t * bl + col) * OSZ : OOB;
        if constexpr (ALWAYS) {
            const unsigned vr = res ? voff : OOB, vb = ep.bn_x ? voff_b : OOB;
            if constexpr (OUT32) pre.res[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vr, 0, 0);
            else pre.res[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, vr, 0, 0);
            if constexpr (STATS) {
                if constexpr (OUT32) pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vb, 0, 0);
                else pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_b, vb, 0, 0);
            }
            continue;
        }
        if (res) {
            if constexpr (OUT32) pre.res[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, voff, 0, 0);
            else pre.res[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, voff, 0, 0);
        }
        if (STATS && ep.bn_x) {
            if constexpr (OUT32) pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, 0, 0);
            else pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_b, voff_b, 0, 0);
        }
    }
}

__device__ __forceinline__ f32x4 epi_unpack(const u32x4 &v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ f32x4 epi_unpack(const u32x2 &v) {
    return (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                   __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}

// Statistics (round 4): every lane keeps its own running (sum, sum of squares / sum dz * xhat) of the four channels it
// stores, across ALL tiles of the persistent workgroup (lst[0], lst[1]); the reduction over the 16 rows of a lane group
// (DPP), the four waves (LDS) and the store of the workgroup's ONE partial row happen once, at the end of the kernel
// (stats_flush).  Round 3 reduced per tile: 32 DPP operations, an LDS exchange and two barriers in every tile's
// epilogue — on the tile loop's critical path.  Fixed order whatever the timing: deterministic.
template <int S, bool OUT32, bool STATS, bool BNLDS = false, int B = 0, int NBA = 1>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[S][NBA], const EpiPre<OUT32> &pre, int row0, int i, int g,
                                              int nb0, int nc, int n_out, __amdgpu_buffer_rsrc_t rs_y,
                                              const void *__restrict__ res, const EpiArgs &ep, f32x4 (&lst)[2],
                                              const f32x4 (*bnv_lds)[4] = nullptr) {
    constexpr unsigned OSZ = OUT32 ? 4u : 2u;
    const unsigned col = (unsigned)(nb0 * 16 + 4 * g);
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    // the BatchNorm vectors of the lane's four channels: ONCE per tile (inside the row loop the stores to y between them kept
    // hipcc from hoisting the loads: sixteen L1 round trips per lane and tile in the data-gradient epilogue)
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = mu, be = mu;
    if constexpr (STATS) {
        if (BNLDS && ep.bn_x) {   // conv_tile16: the vectors sit in LDS (a global load here would queue behind the
            mu = bnv_lds[0][g];      // next tile's prefetched rows)
            is = bnv_lds[1][g];
            ga = bnv_lds[2][g];
            be = bnv_lds[3][g];
        } else if (!BNLDS && ep.bn_x) {
            const unsigned cc = col < (unsigned)nc ? col : 0u;
            mu = *reinterpret_cast<const f32x4 *>(ep.bn_mean + cc);
            is = *reinterpret_cast<const f32x4 *>(ep.bn_invstd + cc);
            if (ep.bn_relu) {
                ga = *reinterpret_cast<const f32x4 *>(ep.bn_gamma + cc);
                be = *reinterpret_cast<const f32x4 *>(ep.bn_beta + cc);
            }
        }
    }
    const unsigned yl = ep.y_ld ? ep.y_ld : (unsigned)nc;   // (ABI 12: y may be a column slice of a wider matrix)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned t = (unsigned)(row0 + s * 16 + i);
        const unsigned voff = (t < (unsigned)n_out && col < (unsigned)nc) ? (t * yl + col) * OSZ : OOB;
        f32x4 a = acc[s][B];
        if (res) a += epi_unpack(pre.res[s]);
        u32x2 packed_out = {0u, 0u};
        if constexpr (!OUT32) {
            packed_out[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
            packed_out[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
        }
        if constexpr (STATS) {
            f32x4 v = a;
            if constexpr (!OUT32) v = epi_unpack(packed_out);
            if (ep.bn_x) {
                const f32x4 xr = epi_unpack(pre.bnx[s]);
                const f32x4 xh = (xr - mu) * is;
                if (ep.bn_relu) {
                    const f32x4 yv = xh * ga + be;
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
        if constexpr (OUT32)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rs_y, voff, 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b64(packed_out, rs_y, voff, 0, 0);
    }
    if constexpr (STATS) {
        lst[0] += st1;
        lst[1] += st2;
    }
}

// the workgroup's partial row of channel block nb0 from the lanes' running sums (all 256 threads call it)
__device__ __forceinline__ void stats_flush(f32x4 (&lst)[2], int i, int g, int wid, int nb0, int nc, const EpiArgs &ep, long long part) {
    __shared__ f32x4 sred[4][2][4];
    f32x4 st1 = lst[0], st2 = lst[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) { st1[q] = row_sum16(st1[q]); st2[q] = row_sum16(st2[q]); }
    if (i == 15) { sred[wid][0][g] = st1; sred[wid][1][g] = st2; }
    doda_sync();
    const int col = nb0 * 16 + 4 * g;
    if (wid == 0 && i == 15 && col < nc) {
        const f32x4 a1 = (sred[0][0][g] + sred[1][0][g]) + (sred[2][0][g] + sred[3][0][g]);
        const f32x4 a2 = (sred[0][1][g] + sred[1][1][g]) + (sred[2][1][g] + sred[3][1][g]);
        stats_emit(ep, part, nc, col, a1, a2);
    }
    doda_sync();   // sred is reused by the next channel block
}

// PERSISTENT: 3 workgroups per CU (LDS footprint), XCD (blockIdx & 7) walks its own contiguous range of
// tiles.  A tile's dependency chain is  list -> rows -> LDS -> multiply -> store; inside a training step the
// operands come from HBM, not from the Infinity Cache a back-to-back micro-benchmark enjoys, and each link
// costs 1-2 us.  So the list (and count) of the workgroup's NEXT tile is requested while the current one is
// multiplied, the epilogue's operands are requested before the multiply phase, and the weight fragments are
// loaded once per workgroup: one exposed round trip per tile (the rows) instead of three.
// MODE 0: bf16, 16 input channels (32-byte rows, a PAIR of offsets per MFMA, 14 units, 3 workgroups per CU).
// MODE 1: bf16, 32 input channels (64-byte rows, one kernel offset per MFMA k = 32, 27 units; LDS 75 KB -> 2
//         workgroups per CU, which still keeps ~70 KB of row loads in flight per CU).
// MODE 2: fp32, 16 input channels (64-byte rows staged exactly as MODE 1; four v_mfma_f32_16x16x4_f32 per unit and
//         subtile — the reference's precision; MFMA-bound at ~53 us for the level-1 layer instead of 80 us).
// DUAL (64-byte rows, 32 output channels: the level-2 layers): both channel blocks in ONE pass over the units — every operand
// row read out of LDS feeds two MFMAs.  (Stamps, level-2 rulebook of the bench batch, 600 tiles on 512 workgroups: two passes
// took 8.4 us of a 14 us tile with the LDS array as the busiest unit.)
template <int MODE, bool OUT32, bool STATS, int MAXNB, bool DUAL = false>
__global__ __launch_bounds__(256, MODE == 0 ? 3 : 2) void conv_tile(const void *__restrict__ x, unsigned x_bytes,
                                                 const void *__restrict__ wp, unsigned wp_bytes, int nc, int NB,
                                                 const int32_t *__restrict__ tbl, int ld, int n_out,
                                                 const TileBookView tb, void *__restrict__ y, unsigned y_bytes,
                                                 const void *__restrict__ res, const EpiArgs ep) {
    constexpr bool WIDE = MODE != 0;
    static_assert(MODE != 2 || OUT32, "fp32 features have fp32 outputs");
    static_assert(!DUAL || (MODE == 1 && (MAXNB == 2 || MAXNB == 4 || !STATS)), "two channel blocks per pass: bf16, 64-byte rows");
    static_assert(MAXNB != 4 || DUAL, "statistics of four channel blocks: two dual passes");
    constexpr int NBA = DUAL ? 2 : 1;                          // channel blocks per pass over the units
    constexpr int S = 4, NU = WIDE ? TB_K : (TB_K + 1) / 2;
    constexpr int RB = WIDE ? 64 : 32;                         // bytes per staged row
    constexpr int PPR = RB / 16;                               // 16-byte pieces per row
    constexpr int CAP = WIDE ? TB_CAP64 : TB_LMAX;             // distinct rows this kernel stages (LDS budget; 10-bit local indices)
    constexpr int NLJ = PPR;                                   // 16-byte list loads per thread (one per pass of 256 / PPR entries)
    constexpr int NRL = 4 * NLJ;                               // row loads per thread
    constexpr int NLP = TB_LIDX_BYTES / 16;                    // 16-byte pieces of the index strip (640)
    constexpr int NLI = (NLP + 255) / 256;                     // ... per thread
    __shared__ __attribute__((aligned(16))) unsigned char rows_s[(CAP + 1) * RB];   // slot 0: the zero row
    __shared__ __attribute__((aligned(16))) unsigned lidx_s[TB_T * TB_LW];   // ten planes of 256 words (tilebook.hpp)
    // BatchNorm statistics: ONE partial row per persistent workgroup (<= 768 rows instead of one per 256 output rows),
    // accumulated per lane across the workgroup's tiles and reduced once at the end (tile_epilogue / stats_flush)
    // channel blocks with statistics: MAXNB = 1 or 2 (up to 32 output channels: every tilebook layer of the U-Net; the
    // dispatcher sends anything else to the dense-table kernel).  A template parameter: the second block's accumulators
    // cost the 16 -> 16 kernel (MAXNB = 1) registers it needs for its third wave per SIMD
    static_assert(MAXNB == 1 || MAXNB == 2 || MAXNB == 4, "statistics of one, two or (dual passes) four channel blocks");
    f32x4 lst[STATS ? MAXNB : 1][2];
#pragma unroll
    for (int b = 0; b < (STATS ? MAXNB : 1); ++b) lst[b][0] = lst[b][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tid0 = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)TB_K * (unsigned)ld * 4u, 0x00020000);

    const int L = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = tb.nt >> 3, rn = tb.nt & 7;
    const int lo = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    const int cnt = qn + (xcd < rn ? 1 : 0);

    // piece h = 16 bytes: PPR consecutive lanes take the pieces of one list entry (a wave reads 32 or 16
    // consecutive entries per instruction).  Entries past the count are -1: their row offset is out
    // of range and loads zeros — no lane is masked, no branch.
    // Round 4: the list is read with 16-byte loads.  Its storage order (tb_upos) keeps entries e0, e0 + 256, e0 + 512,
    // e0 + 768 in one 16-byte group, so a thread takes the groups e0 = j * (256 / PPR) + tid / PPR, j < PPR: 2 or 4 list
    // instructions per thread instead of 8 or 15 four-byte ones (the tile kernels are paced by the number of
    // vector-memory instructions a CU's texture path retires); rid[4 j + k] = entry e0 + 256 k.
    auto entry_of = [&](int j, int k, int tid) { return j * (256 / PPR) + tid / PPR + 256 * k; };
    auto load_list = [&](int tile, int tid, unsigned (&rid)[NRL]) {
        const u32x4 *ul4 = reinterpret_cast<const u32x4 *>(tb.ulist + (size_t)tile * TB_UMAX);
#pragma unroll
        for (int j = 0; j < NLJ; ++j) {
            const u32x4 v = ul4[j * (256 / PPR) + tid / PPR];
#pragma unroll
            for (int k = 0; k < 4; ++k) rid[4 * j + k] = v[k];   // (entries >= CAP: loaded, never written to LDS — a select
                                                                  // here made the wave wait for the NEXT tile's list before it
                                                                  // could park the current tile's rows)
        }
    };
    unsigned rid[NRL];
    int U = 0;
    if (slot < cnt) {
        load_list(lo + slot, tid0, rid);
        U = tb.ucount[lo + slot];
    }
#ifdef DODA_TILE_STAMPS
    int stamp_it = -1;
#endif
    for (int tt = slot; tt < cnt; tt += L) {
        const int tile = lo + tt, t0 = tile * TB_T;
#ifdef DODA_TILE_STAMPS
        ++stamp_it;
#endif
        TILE_STAMP(0);
        // every address below depends only on the lane; laundering the lane id once per tile keeps hipcc from
        // hoisting them out of the tile loop into ~100 long-lived registers
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, i = lane & 15, g = lane >> 4;
        const bool staged = U <= CAP;
        const int row0 = t0 + wid * 64;
        const unsigned half = (unsigned)(WIDE ? g : (g & 1)) * 16u;   // the lane's 16-byte piece of an operand row
        // weight fragments (pair packing; offset 27 lies past the packed buffer: zeros) are streamed per unit,
        // three units ahead: held in registers (56) next to the prefetch state they cost the third wave per
        // SIMD; 13.8 KB of fragments stay in the CU's L1
        // pair packing [o][nb][32 slots], wide packing [o][nb][64 lanes]; 16 bytes per slot
        unsigned lane_w = WIDE ? (unsigned)lane * 16u
                               : (unsigned)(g >> 1) * (unsigned)NB * 512u + (unsigned)((g & 1) * 16 + i) * 16u;
        auto loadw = [&](int u, int b = 0) {
            return __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)u * (unsigned)NB * 1024u + lane_w + (unsigned)b * (WIDE ? 1024u : 512u), 0, 0);
        };
        // the epilogue's operands of the first channel block travel with the rows
        EpiPre<OUT32> pre;
        epi_prefetch<S, OUT32, STATS>(pre, row0, i, g, 0, nc, n_out, y_bytes, res, ep);

        // ---- phase A: rows of this tile (their list is already here), index strip, next tile's list ----
        if (staged) {
            u32x4 rr[NRL];
#pragma unroll
            for (int k = 0; k < NRL; ++k)
                rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, rid[k] * (unsigned)RB + (unsigned)(tid & (PPR - 1)) * 16u, 0, 0);
            u32x4 li4[NLI];
            const u32x4 *li = reinterpret_cast<const u32x4 *>(tb.lidx + (size_t)tile * (TB_T * TB_LW));
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                li4[k] = li[e < NLP ? e : 0];
            }
            if (tt + L < cnt) {
                load_list(tile + L, tid, rid);
                U = tb.ucount[tile + L];
            }
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                if (e < NLP) reinterpret_cast<u32x4 *>(lidx_s)[e] = li4[k];
            }
            if (tid < PPR) reinterpret_cast<u32x4 *>(rows_s)[tid] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                const int e = entry_of(k >> 2, k & 3, tid);
                if (e < CAP) reinterpret_cast<u32x4 *>(rows_s)[PPR + e * PPR + (tid & (PPR - 1))] = rr[k];
            }
        } else {
            // A tile WITHOUT a list (more distinct neighbour rows than the kernel stages: 6 of the bench scene's 2349).  It
            // used to walk the dense table with two dependent round trips per pair of units (table entries -> rows): 14 in
            // series, a tail the whole grid waited for (+1.2 us on the level-1 layer; with the BatchNorm prologue +10 us).
            // Now its 27 x 256 table entries are fetched ONCE, coalesced (thread = row, 27 loads in flight), into the LDS
            // the rows would have used, and the unit loop gathers rows two units ahead from indices it reads from LDS.
            static_assert((CAP + 1) * RB >= TB_K * TB_T * 4, "the table slice fits the row buffer");
            int te[TB_K];
            unsigned ldv = (unsigned)ld;
            asm volatile("" : "+s"(ldv));   // (laundered: otherwise its 27 multiples are hoisted out of the tile loop)
#pragma unroll
            for (int o = 0; o < TB_K; ++o) {
                const unsigned voff = (unsigned)(t0 + tid) < (unsigned)n_out ? ((unsigned)o * ldv + (unsigned)(t0 + tid)) * 4u : OOB;
                te[o] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_t, voff, 0, 0);
            }
            if (tt + L < cnt) {
                load_list(tile + L, tid, rid);
                U = tb.ucount[tile + L];
            }
            int *tab_s = reinterpret_cast<int *>(rows_s);
#pragma unroll
            for (int o = 0; o < TB_K; ++o) tab_s[o * TB_T + tid] = (unsigned)(t0 + tid) < (unsigned)n_out ? te[o] : -1;
        }
        TILE_STAMP(1);
        doda_sync();
        TILE_STAMP(2);

        for (int nb0 = 0; nb0 < NB; nb0 += NBA) {
            if (nb0 > 0) {
                lane_w += (WIDE ? 1024u : 512u) * (unsigned)NBA;
                epi_prefetch<S, OUT32, STATS>(pre, row0, i, g, nb0, nc, n_out, y_bytes, res, ep);
            }

            // ---- phase B ----
            f32x4 acc[S][NBA];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int b = 0; b < NBA; ++b) acc[s][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (staged) {
                // local indices two units ahead, operand rows one unit ahead of the MFMAs (the scheduling
                // barriers keep hipcc from sinking the reads next to their use, which left one LDS round trip
                // exposed per MFMA)
                // local indices (tilebook.hpp): plane 5 p + k of the strip holds, for the lane's four rows (adjacent words: one
                // 16-byte read), the fields of the offsets 2 (3 k + q) + p.  Pair units take offsets 2 u and 2 u + 1: the lane's
                // half of the pair picks the parity — a different plane, the same shift — so a word serves three units; the
                // 64-byte-row kernels walk offsets 0, 1, 2, ...: two word streams (even / odd offsets), six units per word
                const unsigned *my = lidx_s + wid * 64 + tb_lsigma(i) * 4 + ((!WIDE && (g >> 1)) ? 5 * TB_T : 0);
                auto loadp = [&](int plane) { return *reinterpret_cast<const u32x4 *>(my + plane * TB_T); };
                constexpr int NSTR = WIDE ? 2 : 1;       // word streams
                u32x4 lw[NSTR][2];
                auto word_of = [&](int u) -> const u32x4 & { return WIDE ? lw[u & 1][((u >> 1) / 3) & 1] : lw[0][(u / 3) & 1]; };
                auto fetch = [&](int u, u32x4 (&xa)[S]) {
                    const unsigned sh = WIDE ? 10u * (unsigned)((u >> 1) % 3) : 10u * (unsigned)(u % 3);
                    const u32x4 &l = word_of(u);
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        xa[s] = *reinterpret_cast<const u32x4 *>(rows_s + ((__builtin_amdgcn_ubfe(l[s], sh, 10u) * (unsigned)RB) | half));
                };
                // the word after the one unit u starts: requested in the iteration in which unit u's word is first used
                auto next_word = [&](int u) {
                    if constexpr (WIDE) {
                        const int p = u & 1, k = (u >> 1) / 3;
                        if ((u >> 1) % 3 == 0 && 2 * (3 * (k + 1)) + p < TB_K) lw[p][(k + 1) & 1] = loadp(5 * p + k + 1);
                    } else {
                        const int k = u / 3;
                        if (u % 3 == 0 && 3 * (k + 1) < NU) lw[0][(k + 1) & 1] = loadp(k + 1);
                    }
                };
                u32x4 xa[2][S], wr[4][NBA];
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int b = 0; b < NBA; ++b) wr[u][b] = loadw(u, b);
                lw[0][0] = loadp(0);
                if constexpr (WIDE) lw[1][0] = loadp(5);
                fetch(0, xa[0]);
                // (MODE 2, fp32: skipping the MFMAs of (offset, 16-row subtile) slots without a present neighbour — ~45 % of
                // them on a surface scene — was built twice (round 3: per-subtile test inside the unit; round 4: wave-uniform
                // presence masks from ballots, operand reads unconditional, one branch per slot) and measured SLOWER both
                // times: 88 / 154 us against 84 us dense at level 1 — 108 branches per wave and tile between MFMA groups
                // and 250 VGPRs cost more than the skipped matrix work; the dense loop stays)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (u + 1 < NU) fetch(u + 1, xa[(u + 1) & 1]);
                    next_word(u);
                    if (u + 3 < NU) {
#pragma unroll
                        for (int b = 0; b < NBA; ++b) wr[(u + 3) & 3][b] = loadw(u + 3, b);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
#pragma unroll
                        for (int b = 0; b < NBA; ++b) {
                            if constexpr (MODE == 2) mma_f32_k16<false>(acc[s][b], wr[u & 3][b], xa[u & 1][s]);
                            else mma_bf16_k32(acc[s][b], wr[u & 3][b], xa[u & 1][s]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // tile without a list: same units; the table slice sits in LDS (phase A), indices two units ahead, rows one
                // unit ahead of the MFMAs (absent neighbour / row past n_out: -1 -> out-of-range load -> zeros)
                const int *tab_s = reinterpret_cast<const int *>(rows_s) + wid * 64 + i;
                auto ldi = [&](int u, int (&d)[S]) {
                    const int osel = WIDE ? u : 2 * u + (g >> 1);
#pragma unroll
                    for (int s = 0; s < S; ++s) d[s] = osel < TB_K ? tab_s[osel * TB_T + s * 16] : -1;
                };
                auto ldx = [&](const int (&d)[S], u32x4 (&xr)[S]) {
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        xr[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, d[s] >= 0 ? (unsigned)d[s] * (unsigned)RB + half : OOB, 0, 0);
                };
                int ix[3][S];
                u32x4 xo[2][S];
                ldi(0, ix[0]);
                ldi(1, ix[1]);
                ldx(ix[0], xo[0]);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (u + 2 < NU) ldi(u + 2, ix[(u + 2) % 3]);
                    if (u + 1 < NU) ldx(ix[(u + 1) % 3], xo[(u + 1) & 1]);
                    u32x4 wu[NBA];
#pragma unroll
                    for (int b = 0; b < NBA; ++b) wu[b] = loadw(u, b);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
#pragma unroll
                        for (int b = 0; b < NBA; ++b) {
                            if constexpr (MODE == 2) mma_f32_k16<false>(acc[s][b], wu[b], xo[u & 1][s]);
                            else mma_bf16_k32(acc[s][b], wu[b], xo[u & 1][s]);
                        }
                    }
                }
            }
            TILE_STAMP(3);
            // (statistics: static register index — NB <= MAXNB is checked by the launcher)
            if constexpr (DUAL) {
                // the second block's epilogue operands: requested now (the lines its rows share with the first block's are in
                // the cache), consumed after the first block's epilogue
                EpiPre<OUT32> pre1;
                epi_prefetch<S, OUT32, STATS>(pre1, row0, i, g, nb0 + 1, nc, n_out, y_bytes, res, ep);
                if (MAXNB == 4 && nb0 == 2) {   // (the second dual pass of a 64-output-channel layer: blocks 2 and 3)
                    tile_epilogue<S, OUT32, STATS, false, 0, NBA>(acc, pre, row0, i, g, nb0, nc, n_out, rs_y, res, ep, lst[STATS && MAXNB == 4 ? 2 : 0]);
                    tile_epilogue<S, OUT32, STATS, false, 1, NBA>(acc, pre1, row0, i, g, nb0 + 1, nc, n_out, rs_y, res, ep, lst[STATS && MAXNB == 4 ? 3 : 0]);
                } else {
                    tile_epilogue<S, OUT32, STATS, false, 0, NBA>(acc, pre, row0, i, g, nb0, nc, n_out, rs_y, res, ep, lst[0]);
                    tile_epilogue<S, OUT32, STATS, false, 1, NBA>(acc, pre1, row0, i, g, nb0 + 1, nc, n_out, rs_y, res, ep, lst[STATS && MAXNB > 1 ? 1 : 0]);
                }
            } else {
                if (nb0 == 0) tile_epilogue<S, OUT32, STATS>(acc, pre, row0, i, g, nb0, nc, n_out, rs_y, res, ep, lst[0]);
                else tile_epilogue<S, OUT32, STATS>(acc, pre, row0, i, g, nb0, nc, n_out, rs_y, res, ep, lst[STATS ? MAXNB - 1 : 0]);
            }
        }
        TILE_STAMP(4);
        doda_sync();   // the next tile overwrites the staged rows
        TILE_STAMP(5);
    }
    if constexpr (STATS) {   // the workgroup's partial row (zeros when it had no tile)
        const int lane = tid0 & 63, i = lane & 15, g = lane >> 4;
        stats_flush(lst[0], i, g, wid, 0, nc, ep, (long long)blockIdx.x);
#pragma unroll
        for (int b = 1; b < MAXNB; ++b)
            if (b < NB) stats_flush(lst[b], i, g, wid, b, nc, ep, (long long)blockIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------
// conv_tile16 (round 4): the 16 -> 16 channel layer (MODE 0, one channel block) as a SOFTWARE PIPELINE across tiles.
// Wall-clock stamps inside conv_tile (tools/tilestamps.py, level-1 layer, 2349 tiles over 768 workgroups) showed where
// its 28 us go: every tile WAITS 2.1-2.5 us for its rows (one HBM round trip, exposed: the three workgroups of a CU
// start together and stay in step, so they wait together and then contend for the SIMDs together — units 1.35 us
// alone, 2.5-3.0 us in company), the first tile 4.2 us (list, then rows), and 45 workgroups run a fourth tile alone
// (4 us).  Here a workgroup requests the NEXT tile's rows and index strip (into registers: 8 + 3 x 16 bytes per
// thread) as soon as the current tile's are parked in LDS, so the round trip runs under the multiply phase and the
// epilogue; the 14 weight fragments stay in registers for the kernel's lifetime (the streamed loads would queue
// behind the prefetch: vmcnt retires in order), which leaves NO vector-memory wait inside the unit loop.  ~230 VGPRs:
// two workgroups per CU (512 persistent workgroups) — each hides its own latency instead of relying on the third.
// Same arithmetic, same order of operations per output element as conv_tile<0, ...>: results are bit-identical.
template <bool OUT32, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_tile16(const void *__restrict__ x, unsigned x_bytes,
                                                      const void *__restrict__ wp, unsigned wp_bytes, int nc,
                                                      const int32_t *__restrict__ tbl, int ld, int n_out,
                                                      const TileBookView tb, void *__restrict__ y, unsigned y_bytes,
                                                      const void *__restrict__ res, const EpiArgs ep) {
    constexpr int S = 4, NU = (TB_K + 1) / 2, RB = 32, PPR = 2, CAP = TB_LMAX, NLJ = PPR, NRL = 4 * NLJ;
    constexpr int NLP = TB_LIDX_BYTES / 16, NLI = (NLP + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char rows_s[(CAP + 1) * RB];   // slot 0: the zero row
    __shared__ __attribute__((aligned(16))) unsigned lidx_s[TB_T * TB_LW];
    __shared__ f32x4 bnv_s[4][4];   // BatchNorm mean / invstd / gamma / beta of the 16 output channels (data-gradient epilogue)
    f32x4 lst[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (STATS) {
        if (ep.bn_x && tid < 64) {
            const int v = tid >> 4, c = tid & 15;
            const float *src = v == 0 ? ep.bn_mean : v == 1 ? ep.bn_invstd : v == 2 ? ep.bn_gamma : ep.bn_beta;
            reinterpret_cast<float *>(bnv_s)[v * 16 + c] = (src && c < nc) ? src[c] : 0.f;
        }
        // (visible to every wave after the first tile's barrier)
    }
    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)TB_K * (unsigned)ld * 4u, 0x00020000);

    const int L = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = tb.nt >> 3, rn = tb.nt & 7;
    const int lo = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    const int cnt = qn + (xcd < rn ? 1 : 0);
    const unsigned half = (unsigned)(g & 1) * 16u;

    // the kernel's weights: 14 pair fragments (pair packing [o][32 slots] x 16 B; offset 27 lies past the buffer: zeros)
    u32x4 wr[NU];
    {
        const unsigned lane_w = (unsigned)(g >> 1) * 512u + (unsigned)((g & 1) * 16 + i) * 16u;
#pragma unroll
        for (int u = 0; u < NU; ++u) wr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)u * 1024u + lane_w, 0, 0);
    }

    // Lists and strips are read through buffer descriptors: a request for "no tile" (tile < 0) is an out-of-range offset —
    // zeros, no memory traffic — so EVERY iteration issues the same number of vector-memory instructions in the same
    // order and hipcc's wait counts stay exact (a conditional request made it wait for vmcnt(0) before the epilogue:
    // for the next tile's rows)
    const __amdgpu_buffer_rsrc_t rs_ul = __builtin_amdgcn_make_buffer_rsrc((void *)tb.ulist, 0, (unsigned)tb.nt * (unsigned)(TB_UMAX * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_li = __builtin_amdgcn_make_buffer_rsrc((void *)tb.lidx, 0, (unsigned)tb.nt * (unsigned)TB_LIDX_BYTES, 0x00020000);
    auto entry_of = [&](int j, int k) { return j * (256 / PPR) + tid / PPR + 256 * k; };
    const __amdgpu_buffer_rsrc_t rs_uc = __builtin_amdgcn_make_buffer_rsrc((void *)tb.ucount, 0, (unsigned)tb.nt * 4u, 0x00020000);
    auto load_list = [&](int tile, unsigned (&rid)[NRL]) {
        const unsigned base = tile >= 0 ? (unsigned)tile * (unsigned)(TB_UMAX * 4) : OOB;
#pragma unroll
        for (int j = 0; j < NLJ; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_ul, base + (unsigned)(j * (256 / PPR) + tid / PPR) * 16u, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) rid[4 * j + k] = v[k];
        }
    };
    // count, rows (through their list) and index strip of a tile -> registers; `none`: all-ones (every request out of range;
    // count 0).  The count goes FIRST: the tile loop carries it across its back edge (a register copy = a wait for that
    // load), and in-order vmcnt must not make that a wait for the rows
    auto request = [&](int tile, unsigned none, const unsigned (&rid)[NRL], u32x4 (&rr)[NRL], u32x4 (&li4)[NLI], unsigned &count) {
        count = __builtin_amdgcn_raw_buffer_load_b32(rs_uc, none ? OOB : (unsigned)tile * 4u, 0, 0);
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
            const unsigned r = rid[k] | none;   // (a list entry of -1 / -2 is out of range by itself)
            rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, r < 0x04000000u ? r * (unsigned)RB + (unsigned)(tid & (PPR - 1)) * 16u : OOB, 0, 0);
        }
        const unsigned base = none ? OOB : (unsigned)tile * (unsigned)TB_LIDX_BYTES;
#pragma unroll
        for (int k = 0; k < NLI; ++k) {
            const int e = k * 256 + tid;
            li4[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_li, base + (unsigned)(e < NLP ? e : 0) * 16u, 0, 0);
        }
    };

    unsigned rid[NRL];
    u32x4 rr[NRL], li4[NLI];
    unsigned U = 0;   // (per lane, the same value in every lane)
    load_list(slot < cnt ? lo + slot : -1, rid);
    // the first list is needed now anyway: waiting for EVERYTHING requested so far tells hipcc's wait-count pass that the
    // weight fragments have arrived — otherwise every unit of every tile waits "for its fragment", i.e. (in-order vmcnt)
    // for the next tile's rows, and the pipeline is gone
    // (an empty asm that "rewrites" the fragments and the list: the loads cannot sink below it, hipcc waits for them in front
    // of it — vmcnt(0) —, and every later use depends on it)
    asm volatile("" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]),
                      "+v"(wr[8]), "+v"(wr[9]), "+v"(wr[10]), "+v"(wr[11]), "+v"(wr[12]), "+v"(wr[13]),
                      "+v"(rid[0]), "+v"(rid[1]), "+v"(rid[2]), "+v"(rid[3]), "+v"(rid[4]), "+v"(rid[5]), "+v"(rid[6]), "+v"(rid[7]));
    static_assert(NU == 14 && NRL == 8, "operand list of the asm above");
    if (slot >= cnt) return;   // (no tile, no statistics row: the launcher never starts such a workgroup)
    request(lo + slot, 0u, rid, rr, li4, U);
    load_list(slot + L < cnt ? lo + slot + L : -1, rid);
    // four stores to nowhere: the tile loop ends in the epilogue's four stores, and with the same tail of vector-memory
    // instructions on the way INTO the loop the wait counts at its top are exact (without them hipcc assumes the fewest —
    // none — and makes every tile wait for the previous tile's stores to be acknowledged)
    __builtin_amdgcn_sched_barrier(0);   // (they stay BEHIND the loads above)
#pragma unroll
    for (int k = 0; k < S; ++k) __builtin_amdgcn_raw_buffer_store_b32((unsigned)k, rs_y, OOB + 16u * (unsigned)k, 0, 0);   // (distinct: not merged)
    __builtin_amdgcn_sched_barrier(0);
#ifdef DODA_TILE_STAMPS
    const int tid0 = tid;
    int stamp_it = -1;
#endif
    for (int tt = slot; tt < cnt; tt += L) {
        const int tile = lo + tt, t0 = tile * TB_T;
#ifdef DODA_TILE_STAMPS
        ++stamp_it;
#endif
        TILE_STAMP(0);
        const bool staged = __builtin_amdgcn_readfirstlane(U) <= (unsigned)CAP;
        const int row0 = t0 + wid * 64;
        // ---- park this tile's rows and strip (requested one tile ago) ----
        if (staged) {
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                if (e < NLP) reinterpret_cast<u32x4 *>(lidx_s)[e] = li4[k];
            }
            if (tid < PPR) reinterpret_cast<u32x4 *>(rows_s)[tid] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                const int e = entry_of(k >> 2, k & 3);
                if (e < CAP) reinterpret_cast<u32x4 *>(rows_s)[PPR + e * PPR + (tid & (PPR - 1))] = rr[k];
            }
        } else {
            // a tile without a list: its 27 x 256 table entries, coalesced, into the LDS the rows would have used (conv_tile)
            static_assert((CAP + 1) * RB >= TB_K * TB_T * 4, "the table slice fits the row buffer");
            int te[TB_K];
#pragma unroll
            for (int o = 0; o < TB_K; ++o) {
                const unsigned voff = (unsigned)(t0 + tid) < (unsigned)n_out ? ((unsigned)o * (unsigned)ld + (unsigned)(t0 + tid)) * 4u : OOB;
                te[o] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_t, voff, 0, 0);
            }
            int *tab_s = reinterpret_cast<int *>(rows_s);
#pragma unroll
            for (int o = 0; o < TB_K; ++o) tab_s[o * TB_T + tid] = (unsigned)(t0 + tid) < (unsigned)n_out ? te[o] : -1;
        }
        TILE_STAMP(1);
        doda_sync();
        TILE_STAMP(2);

        // ---- requests, in the order their data is needed (vmcnt retires in order): this tile's epilogue operands, the NEXT
        //      tile's rows and strip (its list arrived during the previous tile), the list after that ----
        EpiPre<OUT32> pre;
        epi_prefetch<S, OUT32, STATS, true>(pre, row0, i, g, 0, nc, n_out, y_bytes, res, ep);
        unsigned Unext;
        request(tile + L, tt + L < cnt ? 0u : 0xffffffffu, rid, rr, li4, Unext);
        load_list(tt + 2 * L < cnt ? tile + 2 * L : -1, rid);

        // ---- multiply phase: no vector-memory wait inside ----
        f32x4 acc[S][1];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (staged) {
            // local indices: plane (5 x the lane's offset parity) + u / 3, bits 10 (u % 3) .. + 9 — one 16-byte read per three
            // units for the lane's four rows (tilebook.hpp)
            const unsigned *my = lidx_s + wid * 64 + tb_lsigma(i) * 4 + ((g >> 1) ? 5 * TB_T : 0);
            auto loadp = [&](int k) { return *reinterpret_cast<const u32x4 *>(my + k * TB_T); };
            u32x4 lw[2];
            auto fetch = [&](int u, u32x4 (&xa)[S]) {
                const u32x4 &l = lw[(u / 3) & 1];
#pragma unroll
                for (int s = 0; s < S; ++s)
                    xa[s] = *reinterpret_cast<const u32x4 *>(rows_s + ((__builtin_amdgcn_ubfe(l[s], 10u * (unsigned)(u % 3), 10u) * (unsigned)RB) | half));
            };
            u32x4 xa[2][S];
            lw[0] = loadp(0);
            fetch(0, xa[0]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU) fetch(u + 1, xa[(u + 1) & 1]);
                if (u % 3 == 0 && 3 * (u / 3 + 1) < NU) lw[(u / 3 + 1) & 1] = loadp(u / 3 + 1);   // (first used two units on)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < S; ++s) mma_bf16_k32(acc[s][0], wr[u], xa[u & 1][s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const int *tab_s = reinterpret_cast<const int *>(rows_s) + wid * 64 + i;
            auto ldi = [&](int u, int (&d)[S]) {
                const int osel = 2 * u + (g >> 1);
#pragma unroll
                for (int s = 0; s < S; ++s) d[s] = osel < TB_K ? tab_s[osel * TB_T + s * 16] : -1;
            };
            auto ldx = [&](const int (&d)[S], u32x4 (&xr)[S]) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    xr[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, d[s] >= 0 ? (unsigned)d[s] * (unsigned)RB + half : OOB, 0, 0);
            };
            int ix[3][S];
            u32x4 xo[2][S];
            ldi(0, ix[0]);
            ldi(1, ix[1]);
            ldx(ix[0], xo[0]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) ldi(u + 2, ix[(u + 2) % 3]);
                if (u + 1 < NU) ldx(ix[(u + 1) % 3], xo[(u + 1) & 1]);
#pragma unroll
                for (int s = 0; s < S; ++s) mma_bf16_k32(acc[s][0], wr[u], xo[u & 1][s]);
            }
        }
        TILE_STAMP(3);
        tile_epilogue<S, OUT32, STATS, true>(acc, pre, row0, i, g, 0, nc, n_out, rs_y, res, ep, lst, bnv_s);
        U = Unext;
        TILE_STAMP(4);
        doda_sync();   // the next tile overwrites the staged rows
        TILE_STAMP(5);
    }
    if constexpr (STATS) stats_flush(lst, i, g, wid, 0, nc, ep, (long long)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// conv_up32 (round 4): gathers whose table has (about) ONE source row per output row — the inverse convolution's forward
// (reference model/unet_block.py:78: a fine voxel reads its coarse parent through the k2 s2 rulebook, roles swapped) and the
// data gradient of the strided convolution (unet_block.py:70), 32 input channels.  conv_fast walks the K = 8 table offset by
// offset: eight table loads and eight gathers per row, seven of them out of range — 52 us for the level-2 -> level-1 layer whose
// bytes (48 MB) take 8 us.  Here lane l reads the eight table entries of row l once, the wave exchanges the (source row, offset)
// codes through the LDS crossbar (ds_bpermute), every row's 64 bytes are gathered ONCE, and the eight weight matrices are applied
// as eight MFMA passes whose operand is the row or zero (v_cndmask): matrix work nobody waits for.  Correct for ANY table: a row
// with several sources takes one more pass per extra source (the dispatcher only sends tables with n_in < n_out here).
// Epilogue = the tile kernels' (residual, fp32 output, BatchNorm statistics, one partial row per workgroup = per 256 rows).
template <bool OUT32, bool STATS>
__global__ __launch_bounds__(256) void conv_up32(const void *__restrict__ x, unsigned x_bytes, const void *__restrict__ wp,
                                                 unsigned wp_bytes, int nc, int NB, int K, const int32_t *__restrict__ tbl, int ld,
                                                 int n_out, void *__restrict__ y, unsigned y_bytes, const void *__restrict__ res,
                                                 const EpiArgs ep) {
    constexpr int S = 4, KMAX = 8;
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)K * (unsigned)ld * 4u, 0x00020000);
    const int row0 = (int)blockIdx.x * TB_T + wid * 64;
    // the eight table entries of row (row0 + lane): -1 = no source under that offset
    int te[KMAX];
    {
        const unsigned r = (unsigned)(row0 + lane);
#pragma unroll
        for (int o = 0; o < KMAX; ++o) {
            const bool ok = o < K && r < (unsigned)n_out;
            const int v = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_t, ok ? ((unsigned)o * (unsigned)ld + r) * 4u : OOB, 0, 0);
            te[o] = ok ? v : -1;
        }
    }
    for (int nb0 = 0; nb0 < NB; ++nb0) {
        EpiPre<OUT32> pre;
        epi_prefetch<S, OUT32, STATS>(pre, row0, i, g, nb0, nc, n_out, y_bytes, res, ep);
        u32x4 wr[KMAX];   // the block's weight matrices ([o][nb][64 lanes] x 16 B)
#pragma unroll
        for (int o = 0; o < KMAX; ++o)
            wr[o] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, o < K ? ((unsigned)(o * NB + nb0) * 64u + (unsigned)lane) * 16u : OOB, 0, 0);
        f32x4 acc[S][1];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < KMAX; ++j) {
            // the j-th source of the lane's row as (row << 3 | offset), or -1
            int code = -1, seen = 0;
#pragma unroll
            for (int o = 0; o < KMAX; ++o) {
                const bool v = te[o] >= 0;
                code = (v && seen == j) ? ((te[o] << 3) | o) : code;
                seen += v ? 1 : 0;
            }
            if (__builtin_amdgcn_ballot_w64(code >= 0) == 0ull) break;   // (wave-uniform)
            int cs[S];
            u32x4 xr[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                cs[s] = __builtin_amdgcn_ds_bpermute((s * 16 + i) * 4, code);
                xr[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, cs[s] >= 0 ? (unsigned)(cs[s] >> 3) * 64u + (unsigned)g * 16u : OOB, 0, 0);
            }
#pragma unroll
            for (int o = 0; o < KMAX; ++o) {
                if (__builtin_amdgcn_ballot_w64(code >= 0 && (code & 7) == o) == 0ull) continue;   // no row of the wave under this offset
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const bool mine = cs[s] >= 0 && (cs[s] & 7) == o;
                    u32x4 b;
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = mine ? xr[s][q] : 0u;
                    mma_bf16_k32(acc[s][0], wr[o], b);
                }
            }
        }
        f32x4 lst[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        tile_epilogue<S, OUT32, STATS>(acc, pre, row0, i, g, nb0, nc, n_out, rs_y, res, ep, lst);
        if constexpr (STATS) stats_flush(lst, i, g, wid, nb0, nc, ep, (long long)blockIdx.x);
    }
}

constexpr int BT_MAX_GROUPS = 768;   // 3 workgroups per CU x 256 CUs
// (A/B switches: DODA_TILE_DUAL=0 / doda_set_option(DODA_OPT_TILE_DUAL, 0) keeps the 32-output-channel layers on one channel block
// per pass; doda_set_option(DODA_OPT_TILE_PIPELINE, 0) keeps the 16 -> 16 layers on conv_tile)
bool g_dual = !(getenv("DODA_TILE_DUAL") && getenv("DODA_TILE_DUAL")[0] == '0');
bool g_pipeline = true;
bool g_up = !(getenv("DODA_CONV_UP") && getenv("DODA_CONV_UP")[0] == '0');
bool dual_blocks() { return g_dual; }
constexpr int T16_MAX_GROUPS = 512;  // conv_tile16: 2 workgroups per CU
// conv_tile16 pays off from the point where conv_tile's workgroups run more than one tile each (a single tile per
// workgroup has nothing to prefetch, and three shallow workgroups per CU then beat two)
int tile16_min_tiles() {
    static const int v = [] {
        const char *e = getenv("DODA_TILE16_MIN_TILES");
        const int m = e && *e ? atoi(e) : BT_MAX_GROUPS + 1;
        return m < T16_MAX_GROUPS ? T16_MAX_GROUPS : m;   // (every one of its 512 workgroups must own a tile)
    }();
    return v;
}

bool g_use_tile = true;   // doda_set_option(DODA_OPT_TILE_KERNEL) (A/B measurements)

}  // namespace

bool doda_tile::enabled() { return g_use_tile; }
void doda_tile::set_enabled(bool on) { g_use_tile = on; }
bool doda_tile::pipeline_enabled() { return g_pipeline; }
void doda_tile::set_pipeline(bool on) { g_pipeline = on; }
bool doda_tile::dual_enabled() { return g_dual; }
void doda_tile::set_dual(bool on) { g_dual = on; }
bool doda_tile::up_enabled() { return g_up; }
void doda_tile::set_up(bool on) { g_up = on; }

int doda_tile::launch_conv_up32(bool out32, const void *x, unsigned xb, const void *wp, unsigned wpb, int nc, int NB, int K,
                                const int32_t *tbl, int ld, int n_out, void *y, unsigned yb, const void *res, const EpiArgs &ep,
                                int *n_part, hipStream_t s) {
    const int groups = (n_out + TB_T - 1) / TB_T;
    if (n_part) *n_part = groups;
    const dim3 grid(groups), block(256);
#define GU(O32, ST) hipLaunchKernelGGL((conv_up32<O32, ST>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, K, tbl, ld, n_out, y, yb, res, ep)
    if (out32) { if (ep.stats) GU(true, true); else GU(true, false); }
    else { if (ep.stats) GU(false, true); else GU(false, false); }
#undef GU
    return doda_check_launch();
}

int doda_tile::launch_conv_tile(int mode, bool out32, const void *x, unsigned xb, const void *wp, unsigned wpb, int nc, int NB,
                                const int32_t *tbl, int ld, int n_out, const void *tilebook, void *y, unsigned yb,
                                const void *res, const EpiArgs &ep_in, int *n_part, hipStream_t s) {
    const TileBookView tb = tilebook_view(const_cast<void *>(tilebook), n_out);
    if (mode == 0 && NB == 1 && g_pipeline && tb.nt >= tile16_min_tiles()) {
        const int groups16 = T16_MAX_GROUPS;
        if (n_part) *n_part = groups16;
        const dim3 grid(groups16), block(256);
        const EpiArgs &ep = ep_in;
#define G16(O32, ST)                                                                               \
    hipLaunchKernelGGL((conv_tile16<O32, ST>), grid, block, 0, s, x, xb, wp, wpb, nc, tbl, ld, n_out, tb, y, yb, res, ep)
        if (out32) { if (ep.stats) G16(true, true); else G16(true, false); }
        else { if (ep.stats) G16(false, true); else G16(false, false); }
#undef G16
        return doda_check_launch();
    }
    int groups = (tb.nt + 7) / 8 * 8;   // persistent: 3 (64-byte rows: 2) workgroups per CU, a multiple of the 8 XCDs
    const int max_groups = mode == 0 ? BT_MAX_GROUPS : BT_MAX_GROUPS * 2 / 3;
    if (groups > max_groups) groups = max_groups;
    const dim3 grid(groups), block(256);
    if (ep_in.stats && NB > 2 && !(mode == 1 && NB == 4 && dual_blocks())) return DODA_ERR_UNSUPPORTED;   // (run_gather does not send such calls here)
    const bool two = NB > 1 && ep_in.stats;                   // statistics of a second channel block
    if (n_part) *n_part = groups;      // one statistics row per persistent workgroup
    const EpiArgs &ep = ep_in;
#define GT1(M, O32, ST, NBS)                                                                       \
    hipLaunchKernelGGL((conv_tile<M, O32, ST, NBS>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, tbl, ld, n_out, tb, y, yb, res, ep)
#define GTD(O32, ST)                                                                               \
    hipLaunchKernelGGL((conv_tile<1, O32, ST, 2, true>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, tbl, ld, n_out, tb, y, yb, res, ep)
#define GT(M, O32, ST)                                                                             \
    do {                                                                                           \
        if (ST && two) GT1(M, O32, ST, 2); else GT1(M, O32, ST, 1);                                \
    } while (0)
#define GM(M)                                                                                      \
    do {                                                                                           \
        if (out32) { if (ep.stats) GT(M, true, true); else GT(M, true, false); }                   \
        else { if (ep.stats) GT(M, false, true); else GT(M, false, false); }                       \
    } while (0)
    if (mode == 2) { if (ep.stats) GT(2, true, true); else GT(2, true, false); }
    else if (mode == 1 && NB == 2 && dual_blocks()) {   // 32 output channels: both channel blocks in one pass
        if (out32) { if (ep.stats) GTD(true, true); else GTD(true, false); }
        else { if (ep.stats) GTD(false, true); else GTD(false, false); }
    }
    else if (mode == 1 && NB == 4 && dual_blocks()) {   // 64 output channels: two dual passes (the 32 -> 64 data gradient of level 2)
#define GTQ(O32, ST) hipLaunchKernelGGL((conv_tile<1, O32, ST, 4, true>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, tbl, ld, n_out, tb, y, yb, res, ep)
        if (out32) { if (ep.stats) GTQ(true, true); else GTD(true, false); }
        else { if (ep.stats) GTQ(false, true); else GTD(false, false); }
#undef GTQ
    }
    else if (mode == 1) GM(1);
    else GM(0);
#undef GM
#undef GT
#undef GT1
#undef GTD
    return doda_check_launch();
}


