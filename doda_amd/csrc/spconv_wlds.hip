// Weights-in-LDS gather for the 48 -> 48 channel SubM layers (level 3 of DODA's U-Net: 46 k rows at B = 4, 14
// forward / data-grad launches per step).  conv_fast is bound there by weight re-reads: every 32-row workgroup
// streams the layer's 162 KB of packed fragments from L2 (233 MB per launch = the L2 -> L1 rate of the chip,
// DESIGN.md §9).  Here ONE workgroup per CU (8 waves x 32 rows) copies the fragments ONCE into LDS — 121.5 KB:
// the second 32-channel chunk of a 48-channel input only has 16 real channels, so only its lanes g < 2 are kept —
// and every MFMA operand B then comes from LDS; the rows are gathered from global memory as in conv_fast, two
// units ahead, in straight-line code (27 offsets x 2 chunks, compiler-counted waits).
// Replaces spconv v1.2's indice_conv / indice_conv_backward data path for these layers (reference call sites
// model/unet_block.py:26,29,48); same arithmetic as conv_fast<PBF16W> (fp32 accumulation, one bf16 rounding).
#include "common.hpp"
#include "spconv_common.hpp"

namespace {

constexpr int WK = 27, WCH = 48, WNB = 3;
constexpr int W_SLOTS = 96;                         // 16-byte fragments per (offset, channel block): 64 + 32
constexpr int W_LDS_BYTES = WK * WNB * W_SLOTS * 16;   // 124 416

bool g_use_wlds = true;

// statistics / residual / store of one 16-column block held by 8 waves x S subtiles (cf. tile_epilogue)
// Round 4: the residual / BatchNorm-input rows arrive as arguments — requested at the top of the tile, before the table and the
// gathers (three channel blocks used to wait one after the other for their own loads at the END of the tile's chain) — and the
// BatchNorm vectors come out of LDS (bnv: mean / invstd / gamma / beta x 48 channels, filled once per workgroup).
template <int S, bool STATS>
__device__ __forceinline__ void wlds_epilogue(f32x4 (&acc)[S], const u32x2 (&pre_res)[S], const u32x2 (&pre_bnx)[S], int row0,
                                              int i, int g, int wid, int nb, int n_out, __amdgpu_buffer_rsrc_t rs_y,
                                              const void *__restrict__ res, const EpiArgs &ep, int part, f32x4 (*sred)[2][4],
                                              const float (*bnv)[WCH]) {
    const unsigned col = (unsigned)(nb * 16 + 4 * g);
    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
    auto unpack = [](const u32x2 &v) {
        return (f32x4){__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u),
                       __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
    };
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const unsigned t = (unsigned)(row0 + s * 16 + i);
        const unsigned voff = t < (unsigned)n_out ? (t * (ep.y_ld ? ep.y_ld : (unsigned)WCH) + col) * 2u : OOB;   // (ABI 12: row stride of y)
        f32x4 a = acc[s];
        if (res) a += unpack(pre_res[s]);
        u32x2 packed_out;
        packed_out[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
        packed_out[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
        if constexpr (STATS) {
            f32x4 v = unpack(packed_out);
            if (ep.bn_x) {
                const f32x4 xr = unpack(pre_bnx[s]);
                const f32x4 mu = *reinterpret_cast<const f32x4 *>(&bnv[0][col]);
                const f32x4 is = *reinterpret_cast<const f32x4 *>(&bnv[1][col]);
                const f32x4 xh = (xr - mu) * is;
                if (ep.bn_relu) {
                    const f32x4 ga = *reinterpret_cast<const f32x4 *>(&bnv[2][col]);
                    const f32x4 be = *reinterpret_cast<const f32x4 *>(&bnv[3][col]);
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
        __builtin_amdgcn_raw_buffer_store_b64(packed_out, rs_y, voff, 0, 0);
    }
    if constexpr (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { st1[q] = row_sum16(st1[q]); st2[q] = row_sum16(st2[q]); }
        if (i == 15) { sred[wid][0][g] = st1; sred[wid][1][g] = st2; }
        doda_sync();
        if (wid == 0 && i == 15) {
            f32x4 a1 = sred[0][0][g], a2 = sred[0][1][g];
#pragma unroll
            for (int w = 1; w < 8; ++w) { a1 += sred[w][0][g]; a2 += sred[w][1][g]; }
            stats_emit(ep, (long long)part, WCH, col, a1, a2);
        }
        doda_sync();   // sred is reused by the next channel block / tile
    }
}

template <bool STATS>
__global__ __launch_bounds__(512) void conv_wlds48(const unsigned short *__restrict__ x, unsigned x_bytes,
                                                   const u32x4 *__restrict__ wp, const int32_t *__restrict__ tbl,
                                                   unsigned tbl_bytes, int ld, int n_out, void *__restrict__ y,
                                                   unsigned y_bytes, const void *__restrict__ res, const EpiArgs ep) {
    constexpr int S = 2, RW = 16 * S, TM = 8 * RW;   // 256 rows per workgroup pass
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_raw[];
    u32x4 *wl = reinterpret_cast<u32x4 *>(wl_raw);                      // [27][3][96]
    __shared__ f32x4 sred[STATS ? 8 : 1][2][4];
    __shared__ __attribute__((aligned(16))) float bnv[4][WCH];   // BatchNorm vectors of the data-gradient statistics epilogue

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, tbl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);

    if constexpr (STATS) {
        if (ep.bn_x && tid < 4 * WCH) {
            const int v = tid / WCH, c = tid - v * WCH;
            const float *src = v == 0 ? ep.bn_mean : v == 1 ? ep.bn_invstd : v == 2 ? ep.bn_gamma : ep.bn_beta;
            bnv[v][c] = src ? src[c] : 0.f;
        }
    }
    // ---- the layer's fragments, once per workgroup.  Wide packing in global memory: [o][chunk 2][nb 3][lane 64];
    //      chunk 1 keeps lanes 0..31 (channels 32..47), lanes 32..63 would multiply zero padding ----
    const int n_tiles = (n_out + TM - 1) / TM;
    // gather-table entries of the wave's rows under every offset; the FIRST tile's are requested together with the fragments
    // (they used to start their round trip after the fragments were parked and the workgroup had met at the barrier)
    unsigned off[WK][S];
    auto load_table = [&](int tile) {
        const int row0 = tile * TM + wid * RW;
#pragma unroll
        for (int o = 0; o < WK; ++o)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int t = row0 + s * 16 + i;
                const unsigned voff = t < n_out ? ((unsigned)o * (unsigned)ld + (unsigned)t) * 4u : OOB;
                off[o][s] = __builtin_amdgcn_raw_buffer_load_b32(rs_t, voff, 0, 0);
            }
    };
    {   // all loads first (16 per thread in flight), then the LDS writes: a rolled loop made 16 dependent round trips
        constexpr int NCP = (WK * WNB * W_SLOTS + 511) / 512;
        u32x4 tmp[NCP];
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
            const int e = k * 512 + tid, ec = e < WK * WNB * W_SLOTS ? e : 0;
            const int slot = ec % W_SLOTS, onb = ec / W_SLOTS, nb = onb % WNB, o = onb / WNB;
            const int src = slot < 64 ? ((o * 2 + 0) * WNB + nb) * 64 + slot : ((o * 2 + 1) * WNB + nb) * 64 + (slot - 64);
            tmp[k] = wp[src];
        }
        load_table((int)blockIdx.x);   // (a workgroup without a tile: every offset out of range)
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
            const int e = k * 512 + tid;
            if (e < WK * WNB * W_SLOTS) wl[e] = tmp[k];
        }
    }
    doda_sync();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TM + wid * RW;
        // the epilogue's operands of all three channel blocks: requested now, consumed at the end of the tile
        u32x2 pre_res[WNB][S], pre_bnx[WNB][S];
        {
            // (ABI 12: the residual and the BatchNorm input may be column slices of wider matrices)
            const unsigned rl = ep.res_ld ? ep.res_ld : (unsigned)WCH, bl = ep.bnx_ld ? ep.bnx_ld : (unsigned)WCH;
            const unsigned dense_bytes = (unsigned)n_out * (unsigned)WCH * 2u;
            const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
                (void *)res, 0, ep.res_ld ? ((unsigned)(n_out - 1) * rl + (unsigned)WCH) * 2u : dense_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
                (void *)ep.bn_x, 0, ep.bnx_ld ? ((unsigned)(n_out - 1) * bl + (unsigned)WCH) * 2u : dense_bytes, 0x00020000);
#pragma unroll
            for (int nb = 0; nb < WNB; ++nb)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const unsigned t = (unsigned)(row0 + s * 16 + i);
                    const unsigned cc = (unsigned)(nb * 16 + 4 * g);
                    const unsigned voff = t < (unsigned)n_out ? (t * rl + cc) * 2u : OOB;
                    const unsigned voff_b = t < (unsigned)n_out ? (t * bl + cc) * 2u : OOB;
                    pre_res[nb][s] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, res ? voff : OOB, 0, 0);
                    if constexpr (STATS) pre_bnx[nb][s] = __builtin_amdgcn_raw_buffer_load_b64(rs_b, ep.bn_x ? voff_b : OOB, 0, 0);
                    else pre_bnx[nb][s] = (u32x2){0u, 0u};
                }
        }
        // byte offsets of the wave's rows under every offset (absent / past the end: out of range -> zeros)
        if (tile != (int)blockIdx.x) load_table(tile);
#pragma unroll
        for (int o = 0; o < WK; ++o)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int t = row0 + s * 16 + i;
                off[o][s] = (t < n_out && (int)off[o][s] >= 0) ? off[o][s] * (unsigned)(WCH * 2) : OOB;
            }

        f32x4 acc[WNB][S];
#pragma unroll
        for (int nb = 0; nb < WNB; ++nb)
#pragma unroll
            for (int s = 0; s < S; ++s) acc[nb][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // unit u = (offset u / 2, chunk u & 1).  chunk 0: lane (i, g) takes channels 8g..8g+7; chunk 1: lanes g < 2
        // take channels 32 + 8g.., the others contribute zeros on both operands
        auto gather = [&](int u, u32x4 (&xa)[S]) {
            const int o = u >> 1, c = u & 1;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const unsigned base = off[o][s];
                const unsigned voff = (c == 0 || g < 2) && base != OOB ? base + (unsigned)(c * 64 + g * 16) : OOB;
                xa[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, 0, 0);
            }
        };
        constexpr int NUNIT = 2 * WK;
        constexpr int PD = 4;   // units of gathers in flight ahead of the MFMAs (8 waves per CU: latency is hidden here)
        u32x4 xa[PD + 1][S];
#pragma unroll
        for (int u = 0; u < PD; ++u) gather(u, xa[u]);
#pragma unroll
        for (int u = 0; u < NUNIT; ++u) {
            if (u + PD < NUNIT) gather(u + PD, xa[(u + PD) % (PD + 1)]);
            const int o = u >> 1, c = u & 1;
            u32x4 wf[WNB];
#pragma unroll
            for (int nb = 0; nb < WNB; ++nb) {
                const u32x4 *frag = wl + (o * WNB + nb) * W_SLOTS;
                wf[nb] = c == 0 ? frag[lane] : (g < 2 ? frag[64 + lane] : (u32x4){0u, 0u, 0u, 0u});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < WNB; ++nb)
#pragma unroll
                for (int s = 0; s < S; ++s) mma_bf16_k32(acc[nb][s], wf[nb], xa[u % (PD + 1)][s]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < WNB; ++nb)
            wlds_epilogue<S, STATS>(acc[nb], pre_res[nb], pre_bnx[nb], row0, i, g, wid, nb, n_out, rs_y, res, ep, tile, sred, bnv);
    }
}

}  // namespace

bool doda_wlds::enabled() { return g_use_wlds; }
void doda_wlds::set_enabled(bool on) { g_use_wlds = on; }

int doda_wlds::launch_conv48(const void *x, unsigned x_bytes, const void *wp, const int32_t *tbl, unsigned tbl_bytes, int ld,
                             int n_out, void *y, unsigned y_bytes, const void *res, const EpiArgs &ep_in, int *n_part,
                             hipStream_t s) {
    const int n_tiles = (n_out + 255) / 256;
    const int grid = n_tiles < 256 ? n_tiles : 256;
    if (n_part) *n_part = n_tiles;
    const EpiArgs &ep = ep_in;
    static bool attr_done = false;
    if (!attr_done) {   // more than 64 KB of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute((const void *)conv_wlds48<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_wlds48<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES) != hipSuccess)
            return DODA_ERR_LAUNCH;
        attr_done = true;
    }
    if (ep.stats)
        hipLaunchKernelGGL((conv_wlds48<true>), dim3(grid), dim3(512), W_LDS_BYTES, s, (const unsigned short *)x, x_bytes,
                           (const u32x4 *)wp, tbl, tbl_bytes, ld, n_out, y, y_bytes, res, ep);
    else
        hipLaunchKernelGGL((conv_wlds48<false>), dim3(grid), dim3(512), W_LDS_BYTES, s, (const unsigned short *)x, x_bytes,
                           (const u32x4 *)wp, tbl, tbl_bytes, ld, n_out, y, y_bytes, res, ep);
    return doda_check_launch();
}
