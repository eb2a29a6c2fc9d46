// LDS-staged sparse-convolution kernels over a tilebook (tilebook.hpp / tilebook.hip): conv_tile (forward and data
// gradient of the bf16 16- / 32-channel and fp32 16-channel SubM layers) and bwd_tile (fused data + weight gradient,
// not on the default path).  Dispatched from spconv_gather.hip (run_gather) through doda_tile::launch_conv_tile;
// replaces spconv v1.2's indice_conv / indice_conv_backward data path (reference call sites
// model/unet_block.py:26,29,48) for the layers whose rulebook carries a tilebook.
#include "common.hpp"
#include "tilebook.hpp"
#include "spconv_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// LDS-staged tile kernel (bf16, 16 or 32 input channels, K = 27 SubM: the layers of the two finest U-Net
// levels and the north-star gate).  conv_fast is paced by the texture path: 28 gather instructions per 32
// output rows at 37 % lane use, each costing >= 16 cycles of the CU's L1 whatever its EXEC mask (DESIGN.md
// §4).  Here a workgroup owns a tile of TB_T = 256 consecutive output rows whose neighbourhood the tilebook
// (tilebook.hpp) lists as ~2.2 x 256 DISTINCT input rows:
//   phase A  every distinct row is loaded once, consecutive lanes covering consecutive list entries (runs
//            of consecutive rows -> whole 128-byte lines), and parked in LDS next to the tile's local-index
//            strip (27 x 256 uint16);
//   phase B  per unit (a pair of offsets x 16 channels, or one offset x 32 channels) one 8-byte LDS read
//            returns the lane's four local indices (one per 16-row subtile), four 16-byte LDS reads fetch
//            the operand rows (an absent neighbour is the shared zero row: same address in every lane, a
//            broadcast), four MFMAs accumulate.  The only vector-memory instruction in the loop is the
//            streamed weight fragment (L1-resident).
// Vector-memory instructions per 64 output rows (16 channels): ~9 rows + 8 list + 4 strip + 14 weights + 4
// stores = 39 against 96.  A tile whose neighbourhood exceeds the kernel's capacity (1216 rows of 32 bytes, 960 of
// 64 bytes; 0.05 % / 17 % of the tiles of a 1 cm scene, none at 2 cm) takes
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

template <int S, bool OUT32, bool STATS>
__device__ __forceinline__ void epi_prefetch(EpiPre<OUT32> &pre, int row0, int i, int g, int nb0, int nc, int n_out,
                                             unsigned y_bytes, const void *__restrict__ res, const EpiArgs &ep) {
    constexpr unsigned OSZ = OUT32 ? 4u : 2u;
    const unsigned col = (unsigned)(nb0 * 16 + 4 * g);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)res, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)ep.bn_x, 0, y_bytes, 0x00020000);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned t = (unsigned)(row0 + s * 16 + i);
        const unsigned voff = (t < (unsigned)n_out && col < (unsigned)nc) ? (t * (unsigned)nc + col) * OSZ : OOB;
        if (res) {
            if constexpr (OUT32) pre.res[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, voff, 0, 0);
            else pre.res[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, voff, 0, 0);
        }
        if (STATS && ep.bn_x) {
            if constexpr (OUT32) pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff, 0, 0);
            else pre.bnx[s] = __builtin_amdgcn_raw_buffer_load_b64(rs_b, voff, 0, 0);
        }
    }
}

__device__ __forceinline__ f32x4 epi_unpack(const u32x4 &v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ f32x4 epi_unpack(const u32x2 &v) {
    return (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                   __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}

template <int S, bool OUT32, bool STATS>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[S][1], const EpiPre<OUT32> &pre, int row0, int i, int g, int wid,
                                              int nb0, int nc, int n_out, __amdgpu_buffer_rsrc_t rs_y,
                                              const void *__restrict__ res, const EpiArgs &ep, int part,
                                              f32x4 *wg_acc = nullptr) {
    constexpr unsigned OSZ = OUT32 ? 4u : 2u;
    __shared__ f32x4 sred[STATS ? 4 : 1][2][4];
    const unsigned col = (unsigned)(nb0 * 16 + 4 * g);
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned t = (unsigned)(row0 + s * 16 + i);
        const unsigned voff = (t < (unsigned)n_out && col < (unsigned)nc) ? (t * (unsigned)nc + col) * OSZ : OOB;
        f32x4 a = acc[s][0];
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
                const unsigned cc = col < (unsigned)nc ? col : 0u;
                const f32x4 mu = *reinterpret_cast<const f32x4 *>(ep.bn_mean + cc);
                const f32x4 is = *reinterpret_cast<const f32x4 *>(ep.bn_invstd + cc);
                const f32x4 xh = (xr - mu) * is;
                if (ep.bn_relu) {
                    const f32x4 ga = *reinterpret_cast<const f32x4 *>(ep.bn_gamma + cc);
                    const f32x4 be = *reinterpret_cast<const f32x4 *>(ep.bn_beta + cc);
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
#pragma unroll
        for (int q = 0; q < 4; ++q) { st1[q] = row_sum16(st1[q]); st2[q] = row_sum16(st2[q]); }
        if (i == 15) { sred[wid][0][g] = st1; sred[wid][1][g] = st2; }
        __syncthreads();
        if (wid == 0 && i == 15 && col < (unsigned)nc) {
            const f32x4 a1 = (sred[0][0][g] + sred[1][0][g]) + (sred[2][0][g] + sred[3][0][g]);
            const f32x4 a2 = (sred[0][1][g] + sred[1][1][g]) + (sred[2][1][g] + sred[3][1][g]);
            if (wg_acc) {     // persistent caller: one partial row per WORKGROUP, summed over its tiles here (LDS, this lane only)
                wg_acc[(nb0 * 2 + 0) * 4 + g] += a1;
                wg_acc[(nb0 * 2 + 1) * 4 + g] += a2;
            } else {
                float *dst = ep.stats + (long long)part * 2 * nc + col;
                *reinterpret_cast<f32x4 *>(dst) = a1;
                *reinterpret_cast<f32x4 *>(dst + nc) = a2;
            }
        }
    }
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
// PRE (ABI 6): BatchNorm(+ReLU) prologue.  The staged rows pass through registers on their way to LDS, so the
//         normalisation z = relu((x - mean) * invstd * gamma + beta) is applied THERE — once per distinct row of a tile
//         (~2.5 x 256), not once per gathered operand (12 x 256) — and the multiply phase reads z from LDS as before.  The
//         rows of the list that are the tile's OWN rows (SubM: the centre tap) are also stored to ep.pre_out: the
//         normalised tensor the weight gradient gathers from.  The BatchNorm's apply launch (a read and a write of the
//         whole tensor, and a launch) is gone; its statistics still come from the producing conv's epilogue + `final`.
//         MEASURED (profiles/r03_bn_prologue.txt): bit-equal, and SLOWER than the launch it removes — the tile loop is a
//         latency chain (list -> rows -> LDS -> multiply -> store, ~9 us per tile) and everything added between the row
//         loads and the barrier lengthens it: level-1 16 -> 16 27.3 -> 38.9 us without overflow tiles (the apply launch
//         costs 11 us), 28.5 -> 49.7 us with the bench scene's six tiles without a list, whose gathered operands are
//         normalised 12 x per row in a serial chain (a tail the whole grid waits for); 32 channels +56 us at 601k rows.
//         (Later in the round the tiles without a list got their table slice staged in LDS and pipelined gathers: the
//         prologue kernel 49.7 -> 40.6 us on the bench scene — break-even with 28.5 + 11 us for conv + apply launch.)
//         Kept as an opt-in (DODA_BN_PROLOGUE=1) with its parity tests.
template <int MODE, bool OUT32, bool STATS, bool PRE = false>
__global__ __launch_bounds__(256) void conv_tile(const void *__restrict__ x, unsigned x_bytes,
                                                 const void *__restrict__ wp, unsigned wp_bytes, int nc, int NB,
                                                 const int32_t *__restrict__ tbl, int ld, int n_out,
                                                 const TileBookView tb, void *__restrict__ y, unsigned y_bytes,
                                                 const void *__restrict__ res, const EpiArgs ep) {
    constexpr bool WIDE = MODE != 0;
    static_assert(MODE != 2 || OUT32, "fp32 features have fp32 outputs");
    constexpr int S = 4, NU = WIDE ? TB_K : (TB_K + 1) / 2;
    constexpr int RB = WIDE ? 64 : 32;                         // bytes per staged row
    constexpr int PPR = RB / 16;                               // 16-byte pieces per row
    constexpr int CAP = WIDE ? TB_CAP64 : TB_UMAX;             // distinct rows this kernel stages (LDS budget)
    constexpr int NRL = (PPR * CAP + 255) / 256;               // row loads per thread
    constexpr int NLI = (TB_K * TB_T * 2 / 16 + 255) / 256;   // 16-byte pieces of the index strip per thread
    __shared__ __attribute__((aligned(16))) unsigned char rows_s[(CAP + 1) * RB];   // slot 0: the zero row
    __shared__ __attribute__((aligned(16))) unsigned short lidx_s[TB_K * TB_T];
    // BatchNorm statistics: ONE partial row per persistent workgroup (its tiles summed in LDS by the lanes that used to
    // write a row per tile): <= 768 rows instead of one per 256 output rows, few enough for the BatchNorm's apply pass
    // to reduce them itself (bn.hip bn_fused_*: the separate `final` launch disappears)
    constexpr int MAXNB = 8;
    __shared__ f32x4 wg_acc[STATS ? MAXNB * 2 * 4 : 1];
    if constexpr (STATS) {
        if (threadIdx.x < MAXNB * 2 * 4) wg_acc[threadIdx.x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const int tid0 = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)TB_K * (unsigned)ld * 4u, 0x00020000);

    const int L = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = tb.nt >> 3, rn = tb.nt & 7;
    const int lo = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    const int cnt = qn + (xcd < rn ? 1 : 0);

    // PRE: the BatchNorm vectors sit in LDS (4 x kc floats) and are read into registers per tile, AFTER the index strip
    // has left its registers for LDS: held across the persistent loop their 32 VGPRs pushed the 16-channel kernel from
    // 3 to 2 workgroups per CU (138 -> 196 VGPRs)
    static_assert(!PRE || (MODE != 2 && !OUT32), "the prologue is built for bf16 features and outputs");
    __shared__ __attribute__((aligned(16))) float pre_s[PRE ? 4 * (RB / 2) : 1];
    __amdgpu_buffer_rsrc_t rs_z = rs_x;
    if constexpr (PRE) {
        constexpr int KC = RB / 2;
        if (tid0 < 4 * KC) {
            const float *src = tid0 < KC ? ep.pre_mean : tid0 < 2 * KC ? ep.pre_invstd : tid0 < 3 * KC ? ep.pre_gamma : ep.pre_beta;
            pre_s[tid0] = src[tid0 & (KC - 1)];
        }
        rs_z = __builtin_amdgcn_make_buffer_rsrc(ep.pre_out, 0, ep.pre_out ? x_bytes : 0u, 0x00020000);
        __syncthreads();
    }
    auto pre_vec = [&](PreVec &p, unsigned c0) {   // channels c0 .. c0 + 7
        constexpr int KC = RB / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            p.mu[h] = *reinterpret_cast<const f32x4 *>(pre_s + c0 + 4 * h);
            p.is[h] = *reinterpret_cast<const f32x4 *>(pre_s + KC + c0 + 4 * h);
            p.ga[h] = *reinterpret_cast<const f32x4 *>(pre_s + 2 * KC + c0 + 4 * h);
            p.be[h] = *reinterpret_cast<const f32x4 *>(pre_s + 3 * KC + c0 + 4 * h);
        }
    };

    // piece h = 16 bytes: PPR consecutive lanes take the pieces of one list entry (a wave reads 32 or 16
    // consecutive entries per instruction).  Entries past the count are -1: their row offset is out
    // of range and loads zeros — no lane is masked, no branch.
    auto load_list = [&](int tile, int tid, unsigned (&rid)[NRL]) {
        const int32_t *ul = tb.ulist + (size_t)tile * TB_UMAX;
#pragma unroll
        for (int k = 0; k < NRL; ++k) {
            const int e = (k * 256 + tid) / PPR;
            rid[k] = e < CAP ? (unsigned)ul[tb_upos(e)] : 0xffffffffu;
        }
    };
    unsigned rid[NRL];
    int U = 0;
    if (slot < cnt) {
        load_list(lo + slot, tid0, rid);
        U = tb.ucount[lo + slot];
    }
    for (int tt = slot; tt < cnt; tt += L) {
        const int tile = lo + tt, t0 = tile * TB_T;
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
        auto loadw = [&](int u) { return __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)u * (unsigned)NB * 1024u + lane_w, 0, 0); };
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
            const u32x4 *li = reinterpret_cast<const u32x4 *>(tb.lidx + (size_t)tile * TB_K * TB_T);
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                li4[k] = li[e < TB_K * TB_T * 2 / 16 ? e : 0];
            }
            if (tt + L < cnt) {
                load_list(tile + L, tid, rid);
                U = tb.ucount[tile + L];
            }
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                if (e < TB_K * TB_T * 2 / 16) reinterpret_cast<u32x4 *>(lidx_s)[e] = li4[k];
            }
            if (tid < PPR) reinterpret_cast<u32x4 *>(rows_s)[tid] = (u32x4){0u, 0u, 0u, 0u};
            // PRE: every load slot is normalised, also the ones past the list's count (zeros in, never referenced): 40
            // VALU instructions per slot, and NO branch — a wave-uniform skip of the empty slots made hipcc drain the
            // vector-memory counter at the join, i.e. wait for the NEXT tile's list here (+17 us per level-1 layer)
            PreVec pv;
            if constexpr (PRE) pre_vec(pv, (unsigned)(tid & (PPR - 1)) * 8u);
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                if constexpr (PRE) rr[k] = pre_apply8(rr[k], pv, ep.pre_relu);
                if (k * 256 + tid < PPR * CAP) reinterpret_cast<u32x4 *>(rows_s)[PPR + k * 256 + tid] = rr[k];
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
        if constexpr (PRE) {
            if (!staged && ep.pre_out) {   // a tile without a list: its own rows, normalised, straight to pre_out
                PreVec pv;
                pre_vec(pv, (unsigned)(tid & (PPR - 1)) * 8u);
#pragma unroll
                for (int k = 0; k < PPR; ++k) {
                    const unsigned r = (unsigned)t0 + (unsigned)(k * 256 + tid) / (unsigned)PPR;
                    const unsigned off = r < (unsigned)n_out ? r * (unsigned)RB + (unsigned)(tid & (PPR - 1)) * 16u : OOB;
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pre_apply8(v, pv, ep.pre_relu), rs_z, off, 0, 0);
                }
            }
        }
        __syncthreads();
        if constexpr (PRE) {
            // side output: the tile's own rows (SubM centre tap, offset 13: always present) leave LDS for pre_out, 64
            // consecutive 16-byte pieces per store instruction, while the multiply phase runs
            // (branch-free: a tile without a list, or a call without pre_out, stores out of range)
#pragma unroll
            for (int j = 0; j < PPR; ++j) {
                const int rl = (lane / PPR) + j * (64 / PPR);          // row inside the wave's 64
                const unsigned r = (unsigned)row0 + (unsigned)rl;
                const unsigned slot_c = lidx_s[(TB_K / 2) * TB_T + wid * 64 + tb_pos(rl)];
                const u32x4 v = *reinterpret_cast<const u32x4 *>(rows_s + slot_c * (unsigned)RB + (unsigned)(lane & (PPR - 1)) * 16u);
                const unsigned off = (staged && r < (unsigned)n_out) ? r * (unsigned)RB + (unsigned)(lane & (PPR - 1)) * 16u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_z, off, 0, 0);
            }
        }

        for (int nb0 = 0; nb0 < NB; ++nb0) {
            if (nb0 > 0) {
                lane_w += WIDE ? 1024u : 512u;
                epi_prefetch<S, OUT32, STATS>(pre, row0, i, g, nb0, nc, n_out, y_bytes, res, ep);
            }

            // ---- phase B ----
            f32x4 acc[S][1];
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (staged) {
                // local indices two units ahead, operand rows one unit ahead of the MFMAs (the scheduling
                // barriers keep hipcc from sinking the reads next to their use, which left one LDS round trip
                // exposed per MFMA)
                const unsigned short *my = lidx_s + wid * 64 + i * 4;
                auto loadl = [&](int u) {
                    const int osel = WIDE ? u : 2 * u + (g >> 1);
                    u32x2 v = {0u, 0u};   // offset 27 of the last pair: the zero row
                    if (osel < TB_K) v = *reinterpret_cast<const u32x2 *>(my + osel * TB_T);
                    return v;
                };
                auto fetch = [&](const u32x2 &l, u32x4 (&xa)[S]) {
                    xa[0] = *reinterpret_cast<const u32x4 *>(rows_s + (l[0] & 0xffffu) * (unsigned)RB + half);
                    xa[1] = *reinterpret_cast<const u32x4 *>(rows_s + (l[0] >> 16) * (unsigned)RB + half);
                    xa[2] = *reinterpret_cast<const u32x4 *>(rows_s + (l[1] & 0xffffu) * (unsigned)RB + half);
                    xa[3] = *reinterpret_cast<const u32x4 *>(rows_s + (l[1] >> 16) * (unsigned)RB + half);
                };
                u32x4 xa[2][S], wr[4];
                u32x2 lr[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) wr[u] = loadw(u);
                lr[0] = loadl(0);
                lr[1] = loadl(1);
                fetch(lr[0], xa[0]);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (u + 2 < NU) lr[(u + 2) % 3] = loadl(u + 2);
                    if (u + 1 < NU) fetch(lr[(u + 1) % 3], xa[(u + 1) & 1]);
                    if (u + 3 < NU) wr[(u + 3) & 3] = loadw(u + 3);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        if constexpr (MODE == 2) mma_f32_k16(acc[s][0], wr[u & 3], xa[u & 1][s]);
                        else mma_bf16_k32(acc[s][0], wr[u & 3], xa[u & 1][s]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // tile without a list: same units; the table slice sits in LDS (phase A), indices two units ahead, rows one
                // unit ahead of the MFMAs (absent neighbour / row past n_out: -1 -> out-of-range load -> zeros)
                const int *tab_s = reinterpret_cast<const int *>(rows_s) + wid * 64 + i;
                PreVec pvh;   // PRE: the vectors of the operand piece this lane gathers (piece `half`, not the staging piece)
                if constexpr (PRE) pre_vec(pvh, half / 2u);
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
                    const u32x4 wu = loadw(u);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        u32x4 xv = xo[u & 1][s];
                        if constexpr (PRE) {   // an absent neighbour stays zero
                            const u32x4 z = pre_apply8(xv, pvh, ep.pre_relu);
                            xv = ix[u % 3][s] >= 0 ? z : (u32x4){0u, 0u, 0u, 0u};
                        }
                        if constexpr (MODE == 2) mma_f32_k16(acc[s][0], wu, xv);
                        else mma_bf16_k32(acc[s][0], wu, xv);
                    }
                }
            }
            tile_epilogue<S, OUT32, STATS>(acc, pre, row0, i, g, wid, nb0, nc, n_out, rs_y, res, ep, tile, STATS ? wg_acc : nullptr);
            if (STATS && nb0 + 1 < NB) __syncthreads();   // the statistics scratch is reused by the next channel block
        }
        __syncthreads();   // the next tile overwrites the staged rows
    }
    if constexpr (STATS) {   // the workgroup's partial row (zeros when it had no tile)
        __syncthreads();
        const int lane = tid0 & 63, i = lane & 15, g = lane >> 4;
        if (wid == 0 && i == 15) {
            for (int nb0 = 0; nb0 < NB && nb0 < MAXNB; ++nb0) {
                const int col = nb0 * 16 + 4 * g;
                if (col < nc) {
                    float *dst = ep.stats + (long long)blockIdx.x * 2 * nc + col;
                    stats_store4(dst, wg_acc[(nb0 * 2 + 0) * 4 + g]);
                    stats_store4(dst + nc, wg_acc[(nb0 * 2 + 1) * 4 + g]);
                }
            }
        }
        stats_finish(ep, nc);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused SubM backward of a 16 -> 16 bf16 layer over a tilebook: data gradient AND weight gradient from ONE
// staging of the output gradient's neighbourhood.  SubM tables are their own transpose
// under offset mirroring, so both gradients read dy through the SAME table entries of the tile's rows s:
//     dx[s]    = sum_o  dy[nbr_o(s)] . W[26-o]^T                    (rows are the MFMA M dimension)
//     dW[o]   += sum_s  x[s]^T . dy[nbr_{26-o}(s)]                  (rows are the MFMA k dimension)
// Phase A stages the distinct dy rows of the tile in LDS exactly as conv_tile does, plus the tile's own x
// rows (8 KB).  Phase B1 is conv_tile's loop.  Phase B2 reads the SAME staged rows in k-order with
// ds_read_b64_tr_b16, whose per-lane addresses make it a gather: lane s = 4q + c of a 16-lane group points
// at chunk c (4 channels) of the row that local index q selects and receives channel t of rows q = 0..3 —
// no second copy of the data, no LDS write -> read round trip (the LDS kernel's 70 % bank-conflict wall,
// profiles/r01_pmc_wgrad_bf16_l1.txt).  Wave w owns the offsets o = w (mod 4): its 7 accumulators (28
// VGPRs) stay in registers across all tiles of the PERSISTENT workgroup, which writes one partial
// [27][16][16] at the end; bwd_tile_reduce sums the partials in a fixed order (deterministic).
// Workgroups: 3 per CU (LDS 53 KB), XCD x walks its own contiguous range of tiles.
// An overflow tile (more distinct rows than TB_CAP64) takes the dense table: B1 gathers from global memory,
// B2 stages four offsets at a time (4 x 256 rows, tile order) and runs the same transposed reads.
// ---------------------------------------------------------------------------------------------
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr_b64(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// hipcc does not count LDS operations issued from inline asm: wait for all of them before the first use
template <class R>
__device__ __forceinline__ void lds_wait(R &first) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(first) : : "memory");
}

constexpr int BT_MAX_GROUPS = 768;   // 3 workgroups per CU x 256 CUs

template <bool STATS>
__global__ __launch_bounds__(256) void bwd_tile(const unsigned short *__restrict__ dy, unsigned feat_bytes,
                                                const unsigned short *__restrict__ x,
                                                const void *__restrict__ wp, unsigned wp_bytes,
                                                const int32_t *__restrict__ tbl, int ld, int n,
                                                const TileBookView tb, void *__restrict__ dx,
                                                const EpiArgs ep, float *__restrict__ part) {
    constexpr int S = 4, NU = (TB_K + 1) / 2;
    // staged rows and the local-index strip sit back to back: the overflow path, which has no strip, stages
    // 4 x TB_T rows across both
    constexpr int CAP = TB_CAP64;   // LDS budget with the x tile next to the rows: 3 workgroups per CU
    constexpr int ROWS_BYTES = (CAP + 1) * 32;   // slot 0: the zero row
    __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS_BYTES + TB_K * TB_T * 2];
    __shared__ __attribute__((aligned(16))) unsigned short xs[TB_T * 16];
    static_assert(ROWS_BYTES % 16 == 0 && 4 * TB_T * 32 <= ROWS_BYTES + TB_K * TB_T * 2, "overflow path stages 4 x TB_T rows");
    unsigned char *const rows_s = smem;
    unsigned short *const lidx_s = reinterpret_cast<unsigned short *>(smem + ROWS_BYTES);

    const int tid0 = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void *)dx, 0, feat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, (unsigned)TB_K * (unsigned)ld * 4u, 0x00020000);

    // persistent schedule: XCD (blockIdx & 7) owns one contiguous range of tiles, its L workgroups stride it
    const int L = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = tb.nt >> 3, rn = tb.nt & 7;
    const int lo = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    const int cnt = qn + (xcd < rn ? 1 : 0);

    f32x4 dw[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) dw[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tt = slot; tt < cnt; tt += L) {
        const int tile = lo + tt, t0 = tile * TB_T;
        // Every address below depends only on the lane: left alone, hipcc hoists ~100 of them out of the
        // tile loop and holds them in registers for the whole kernel (occupancy 3 -> 2).  The lane id is
        // laundered once per tile, so they are recomputed (a few dozen VALU operations per tile).
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, i = lane & 15, g = lane >> 4;
        const unsigned half = (unsigned)(g & 1) * 16u;
        const int q4 = i >> 2, c4 = i & 3;
        const unsigned rows_base = (unsigned)(uintptr_t)rows_s, xs_base = (unsigned)(uintptr_t)xs;
        // transposed-read address of k-step 0: row 8g + q4 (and +4), chunk c4
        const unsigned xs_addr = xs_base + (unsigned)((8 * g + q4) * 32 + c4 * 8);
        // data-grad weight fragments (W[26-o]^T, pair packing) are streamed per unit, three units ahead (the
        // registers go to the weight-gradient accumulators; 13.8 KB of fragments stay in the CU's L1)
        const unsigned lane_w = (unsigned)(g >> 1) * 512u + (unsigned)((g & 1) * 16 + i) * 16u;
        auto loadw = [&](int u) { return __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)u * 1024u + lane_w, 0, 0); };
        // ---- phase A ----
        constexpr int NRL = (2 * CAP + 255) / 256;
        unsigned rid[NRL];
        {
            const int32_t *ul = tb.ulist + (size_t)tile * TB_UMAX;
#pragma unroll
            for (int k = 0; k < NRL; ++k) {
                const int e = (k * 256 + tid) >> 1;
                rid[k] = e < CAP ? (unsigned)ul[tb_upos(e)] : 0xffffffffu;
            }
        }
        constexpr int NLI = (TB_K * TB_T * 2 / 16 + 255) / 256;
        u32x4 li4[NLI];
        {
            const u32x4 *li = reinterpret_cast<const u32x4 *>(tb.lidx + (size_t)tile * TB_K * TB_T);
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                li4[k] = li[e < TB_K * TB_T * 2 / 16 ? e : 0];
            }
        }
        u32x4 xt[2];   // the tile's own x rows (rows past n: out of range -> zeros)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int h = k * 256 + tid;
            xt[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(t0 + (h >> 1)) * 32u + (unsigned)(h & 1) * 16u, 0, 0);
        }
        const int U = tb.ucount[tile];
        const bool staged = U <= CAP;
#pragma unroll
        for (int k = 0; k < 2; ++k) reinterpret_cast<u32x4 *>(xs)[k * 256 + tid] = xt[k];
        if (staged) {
            u32x4 rr[NRL];
#pragma unroll
            for (int k = 0; k < NRL; ++k)
                rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, rid[k] * 32u + (unsigned)(tid & 1) * 16u, 0, 0);
#pragma unroll
            for (int k = 0; k < NLI; ++k) {
                const int e = k * 256 + tid;
                if (e < TB_K * TB_T * 2 / 16) reinterpret_cast<u32x4 *>(lidx_s)[e] = li4[k];
            }
            if (tid < 2) reinterpret_cast<u32x4 *>(rows_s)[tid] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < NRL; ++k)
                if (k * 256 + tid < 2 * CAP) reinterpret_cast<u32x4 *>(rows_s)[2 + k * 256 + tid] = rr[k];
        }
        __syncthreads();

        // ---- phase B1: data gradient (conv_tile's loop) ----
        EpiPre<false> pre;
        epi_prefetch<S, false, STATS>(pre, t0 + wid * 64, i, g, 0, 16, n, feat_bytes, nullptr, ep);
        f32x4 acc[S][1];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int row0 = t0 + wid * 64;
        if (staged) {
            // local indices two units ahead, operand rows one unit ahead, weight fragments three ahead
            const unsigned short *my = lidx_s + wid * 64 + i * 4;
            auto loadl = [&](int u) {
                const int osel = 2 * u + (g >> 1);
                u32x2 v = {0u, 0u};   // offset 27 of the last pair: the zero row
                if (osel < TB_K) v = *reinterpret_cast<const u32x2 *>(my + osel * TB_T);
                return v;
            };
            auto fetch = [&](const u32x2 &l, u32x4 (&xa)[S]) {
                xa[0] = *reinterpret_cast<const u32x4 *>(rows_s + (l[0] & 0xffffu) * 32u + half);
                xa[1] = *reinterpret_cast<const u32x4 *>(rows_s + (l[0] >> 16) * 32u + half);
                xa[2] = *reinterpret_cast<const u32x4 *>(rows_s + (l[1] & 0xffffu) * 32u + half);
                xa[3] = *reinterpret_cast<const u32x4 *>(rows_s + (l[1] >> 16) * 32u + half);
            };
            u32x4 xa[2][S], wr[4];
            u32x2 lr[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) wr[u] = loadw(u);
            lr[0] = loadl(0);
            lr[1] = loadl(1);
            fetch(lr[0], xa[0]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) lr[(u + 2) % 3] = loadl(u + 2);
                if (u + 1 < NU) fetch(lr[(u + 1) % 3], xa[(u + 1) & 1]);
                if (u + 3 < NU) wr[(u + 3) & 3] = loadw(u + 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < S; ++s) mma_bf16_k32(acc[s][0], wr[u & 3], xa[u & 1][s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // (ld laundered per tile: otherwise hipcc hoists the 27 x ld table offsets of this rarely taken
            // path out of the tile loop and holds them in ~60 registers)
            unsigned ldv = (unsigned)ld;
            asm volatile("" : "+s"(ldv));
#pragma unroll
            for (int u0 = 0; u0 < NU; u0 += 2) {
                unsigned go[2][S];
#pragma unroll
                for (int du = 0; du < 2; ++du) {
                    const int osel = 2 * (u0 + du) + (g >> 1);
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int t = row0 + s * 16 + i;
                        const unsigned voff = (osel < TB_K && t < n) ? ((unsigned)osel * ldv + (unsigned)t) * 4u : OOB;
                        go[du][s] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, voff, 0, 0);
                    }
                }
#pragma unroll
                for (int du = 0; du < 2; ++du) {
                    const int osel = 2 * (u0 + du) + (g >> 1);
                    u32x4 xa[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int t = row0 + s * 16 + i;
                        const bool present = osel < TB_K && t < n && (int)go[du][s] >= 0;
                        xa[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, present ? go[du][s] * 32u + half : OOB, 0, 0);
                    }
                    const u32x4 wu = loadw(u0 + du);
#pragma unroll
                    for (int s = 0; s < S; ++s) mma_bf16_k32(acc[s][0], wu, xa[s]);
                }
            }
        }
        tile_epilogue<S, false, STATS>(acc, pre, row0, i, g, wid, 0, 16, n, rs_o, nullptr, ep, tile);

        // ---- phase B2: weight gradient ----
        if (staged) {
            const unsigned xs_a = xs_addr;
            const unsigned short *li_t = lidx_s + tb_pos(8 * g + q4);
#pragma unroll
            for (int ks = 0; ks < TB_T / 32; ++ks) {
                // x^T fragment of the k-step: channel i of rows 32 ks + 8 g + 0..7
                const unsigned xa_addr = xs_a + (unsigned)ks * 32u * 32u;
                s16x4 a_lo = lds_tr_b64(xa_addr), a_hi = lds_tr_b64(xa_addr + 4u * 32u);
                const unsigned short *lk = li_t + (ks >> 1) * 64 + (ks & 1) * 2;
                unsigned ra[7], rb[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) {
                    const int o = wid + 4 * m;             // owned offset; the table entry is its mirror
                    const int om = o < TB_K ? TB_K - 1 - o : 0;
                    ra[m] = rows_base + (unsigned)lk[om * TB_T] * 32u + (unsigned)c4 * 8u;
                    rb[m] = rows_base + (unsigned)lk[om * TB_T + 16] * 32u + (unsigned)c4 * 8u;
                }
                s16x4 b_lo[7], b_hi[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) { b_lo[m] = lds_tr_b64(ra[m]); b_hi[m] = lds_tr_b64(rb[m]); }
                lds_wait(a_lo);
                const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int m = 0; m < 7; ++m) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b_lo[m], b_hi[m], 0, 1, 2, 3, 4, 5, 6, 7));
                    if (wid + 4 * m < TB_K) dw[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, dw[m], 0, 0, 0);
                }
            }
        } else {
            // four offsets per pass: wave w' stages dy[tbl[26 - (4 pass + w')][t0 + r]] at rows_s[w' * 256 + r]
            unsigned ldw = (unsigned)ld;
            asm volatile("" : "+s"(ldw));
            for (int pass = 0; pass < 7; ++pass) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int h = k * 256 + tid, w2 = h >> 9, r = (h >> 1) & 255, o = 4 * pass + w2;
                    int gi = -1;
                    if (o < TB_K && t0 + r < n) gi = tbl[(size_t)(TB_K - 1 - o) * ldw + (size_t)(t0 + r)];
                    reinterpret_cast<u32x4 *>(rows_s)[h] =
                        __builtin_amdgcn_raw_buffer_load_b128(rs_dy, gi >= 0 ? (unsigned)gi * 32u + (unsigned)(h & 1) * 16u : OOB, 0, 0);
                }
                __syncthreads();
                if (4 * pass + wid < TB_K) {
#pragma unroll
                    for (int ks = 0; ks < TB_T / 32; ++ks) {
                        const unsigned xa_addr = xs_addr + (unsigned)ks * 32u * 32u;
                        const unsigned ba = rows_base + (unsigned)(wid * 256 + ks * 32 + 8 * g + q4) * 32u + (unsigned)c4 * 8u;
                        s16x4 a_lo = lds_tr_b64(xa_addr), a_hi = lds_tr_b64(xa_addr + 4u * 32u);
                        s16x4 b_lo = lds_tr_b64(ba), b_hi = lds_tr_b64(ba + 4u * 32u);
                        lds_wait(a_lo);
                        const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < 7; ++m)
                            if (m == pass) dw[m] += d;
                    }
                }
            }
        }
        __syncthreads();   // the next tile overwrites the staged rows
    }
    // partial of this workgroup: dW[o][ci = 4 g + r][co = i] for the wave's offsets
    const int i = tid0 & 15, g = (tid0 & 63) >> 4;
    float *dst = part + (size_t)blockIdx.x * (TB_K * 256);
#pragma unroll
    for (int m = 0; m < 7; ++m) {
        const int o = wid + 4 * m;
        if (o < TB_K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[o * 256 + (4 * g + r) * 16 + i] = dw[m][r];
        }
    }
}

// dw[e] (+)= sum over workgroups, fixed order.  One workgroup per 32 outputs (128-byte segments of the
// partial rows); its 16 lane groups take every 16th partial, eight loads in flight each, and are combined in
// a fixed tree through LDS.  (A thread per output walking all partials serially took ~90 us: 27 workgroups,
// one dependent HBM round trip per four partials.)
__global__ __launch_bounds__(512) void bwd_tile_reduce(const float *__restrict__ part, int n_part, float *__restrict__ dw,
                                                       int accumulate) {
    __shared__ float red[16][32];
    const int j = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + j;
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.f;
    int b = p;
    for (; b + 7 * 16 < n_part; b += 8 * 16) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += part[(size_t)(b + q * 16) * (TB_K * 256) + e];
    }
    for (int q = 0; b < n_part; b += 16, ++q) a[q & 7] += part[(size_t)b * (TB_K * 256) + e];
    red[p][j] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (p == 0) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][j];
        dw[e] = accumulate ? dw[e] + v : v;
    }
}


bool g_use_tile = true;   // doda_spconv_set_tile_kernel (A/B measurements)

}  // namespace

bool doda_tile::enabled() { return g_use_tile; }

int doda_tile::launch_conv_tile(int mode, bool out32, const void *x, unsigned xb, const void *wp, unsigned wpb, int nc, int NB,
                                const int32_t *tbl, int ld, int n_out, const void *tilebook, void *y, unsigned yb,
                                const void *res, const EpiArgs &ep_in, int *n_part, hipStream_t s) {
    const TileBookView tb = tilebook_view(const_cast<void *>(tilebook), n_out);
    int groups = (tb.nt + 7) / 8 * 8;   // persistent: 3 (64-byte rows: 2) workgroups per CU, a multiple of the 8 XCDs
    const int max_groups = mode == 0 ? BT_MAX_GROUPS : BT_MAX_GROUPS * 2 / 3;
    if (groups > max_groups) groups = max_groups;
    const dim3 grid(groups), block(256);
    if (NB > 8 && ep_in.stats) return DODA_ERR_UNSUPPORTED;   // (the per-workgroup statistics accumulators hold 8 channel blocks)
    if (n_part) *n_part = groups;      // one statistics row per persistent workgroup
    EpiArgs ep = ep_in;
    if (ep.stats) doda_fin::arm(ep, groups, (unsigned)groups, s);
    if (ep.pre_mean) {   // BatchNorm prologue: bf16 features and outputs, SubM (the table's rows are the input's rows)
        if (!doda_tile::takes_prologue(mode, out32) || (size_t)xb != (size_t)n_out * (mode == 0 ? 32u : 64u))
            return DODA_ERR_UNSUPPORTED;
#define GP(M, ST)                                                                                  \
    hipLaunchKernelGGL((conv_tile<M, false, ST, true>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, tbl, ld, n_out, tb, y, yb, res, ep)
        if (mode == 1) { if (ep.stats) GP(1, true); else GP(1, false); }
        else { if (ep.stats) GP(0, true); else GP(0, false); }
#undef GP
        return doda_check_launch();
    }
#define GT(M, O32, ST)                                                                             \
    hipLaunchKernelGGL((conv_tile<M, O32, ST>), grid, block, 0, s, x, xb, wp, wpb, nc, NB, tbl, ld, n_out, tb, y, yb, res, ep)
#define GM(M)                                                                                      \
    do {                                                                                           \
        if (out32) { if (ep.stats) GT(M, true, true); else GT(M, true, false); }                   \
        else { if (ep.stats) GT(M, false, true); else GT(M, false, false); }                       \
    } while (0)
    if (mode == 2) { if (ep.stats) GT(2, true, true); else GT(2, true, false); }
    else if (mode == 1) GM(1);
    else GM(0);
#undef GM
#undef GT
    return doda_check_launch();
}

extern "C" size_t doda_spconv_bwd_tile_workspace_bytes(void) {
    return (size_t)BT_MAX_GROUPS * TB_K * 256 * sizeof(float) + align_up((size_t)TB_K * 32 * 16, 256);
}

extern "C" int doda_spconv_bwd_tile_bf16(const uint16_t *dy, const uint16_t *x, int32_t n_rows, const float *w,
                                         int32_t w_packed, const int32_t *tbl, int32_t ld, const void *tilebook,
                                         void *dx, float *dw, int32_t accumulate, void *ws, size_t ws_bytes,
                                         const doda_conv_epilogue *epi, doda_stream_t stream) {
    if (n_rows < 0 || ld < n_rows) return DODA_ERR_INVALID;
    if (n_rows == 0) {
        if (epi && epi->stats_rows_h) *epi->stats_rows_h = 0;
        return DODA_OK;
    }
    if (!dy || !x || !w || !tbl || !tilebook || !dx || !dw || !ws) return DODA_ERR_INVALID;
    if (ws_bytes < doda_spconv_bwd_tile_workspace_bytes()) return DODA_ERR_WORKSPACE;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)tilebook | (uintptr_t)ws) & 15) return DODA_ERR_UNSUPPORTED;
    if ((size_t)n_rows * 32 >= 0x7ffffff0ull || (size_t)TB_K * ld * 4 >= 0xffffffffull) return DODA_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    float *part = (float *)ws;
    void *wpk = (char *)ws + (size_t)BT_MAX_GROUPS * TB_K * 256 * sizeof(float);
    const size_t need = (size_t)TB_K * 32 * 16;
    const void *wp = w;
    if (!w_packed) {   // w: fp32 [27][16 out... stored [K][nc][kc] as every data-grad call] -> W[26-o]^T pair fragments
        doda_tile::pack_pair_layout2(w, wpk, s);
        wp = wpk;
    }
    EpiArgs ep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (epi && epi->stats) {
        if (!epi->stats_rows_h) return DODA_ERR_INVALID;
        ep.stats = epi->stats;
        if (epi->bn_x) {
            if (!epi->bn_mean || !epi->bn_invstd || !epi->bn_gamma || !epi->bn_beta) return DODA_ERR_INVALID;
            ep.bn_x = epi->bn_x;
            ep.bn_mean = epi->bn_mean; ep.bn_invstd = epi->bn_invstd;
            ep.bn_gamma = epi->bn_gamma; ep.bn_beta = epi->bn_beta;
            ep.bn_relu = epi->bn_relu;
        }
    }
    const TileBookView tb = tilebook_view(const_cast<void *>(tilebook), n_rows);
    int groups = (tb.nt + 7) / 8 * 8;
    if (groups > BT_MAX_GROUPS) groups = BT_MAX_GROUPS;
    const unsigned fb = (unsigned)((size_t)n_rows * 32);
    if (ep.stats)
        hipLaunchKernelGGL((bwd_tile<true>), dim3(groups), dim3(256), 0, s, dy, fb, x, wp, (unsigned)need, tbl, (int)ld,
                           (int)n_rows, tb, dx, ep, part);
    else
        hipLaunchKernelGGL((bwd_tile<false>), dim3(groups), dim3(256), 0, s, dy, fb, x, wp, (unsigned)need, tbl, (int)ld,
                           (int)n_rows, tb, dx, ep, part);
    hipLaunchKernelGGL(bwd_tile_reduce, dim3(TB_K * 256 / 32), dim3(512), 0, s, part, groups, dw, accumulate);
    if (epi && epi->stats_rows_h) *epi->stats_rows_h = ep.stats ? tb.nt : 0;
    return doda_check_launch();
}

extern "C" void doda_spconv_set_tile_kernel(int32_t on) { g_use_tile = on != 0; }

