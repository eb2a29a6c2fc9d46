// Coarse-level executor: the deep levels of DODA's SparseConv U-Net in ONE persistent launch per direction.
//
// Reference: model/unet_block.py:55-100 (UBlock recursion) over model/unet_block.py:9-37 (ResidualBlock:
// BatchNorm1d -> ReLU -> SubMConv3d -> BatchNorm1d -> ReLU -> SubMConv3d + skip), the strided / inverse convolution pair
// (:67-79) and the concatenation skip (:89-93).  At the U-Net's levels 5-7 a batch of four ScanNet scenes holds
// ~1.9 k / 0.4 k / 0.1 k rows: every layer there is a launch-floor kernel (7.5 us for a convolution, 8-10 us for a
// BatchNorm), ~110 launches forward + backward that also cost the issuing thread ~9 us each.
//
// MI355X-first design (numbers: tools/probe/xcdbar.hip, xcdgather.hip):
//   * A device-side OP LIST (doda_cx_op, include/doda_hip.h) is walked by G persistent workgroups of 512 threads that sit
//     on ONE XCD (observed placement: block b runs on XCD b % 8; the grid is 8 G blocks, those with b % 8 != 0 leave at
//     once).  One XCD = one L2: a grid barrier over its 32 workgroups costs 1.1 us (one relaxed agent-scope counter,
//     relaxed polls); over all 256 CUs it costs 4.1 us — more than the kernel boundary it replaces.
//   * Correctness never depends on the placement: everything one workgroup writes for another goes out with sc1
//     (write-through) stores, is drained (s_waitcnt vmcnt(0) in every wave) before the workgroup arrives at the barrier,
//     and is read back with sc1 loads (L1 bypass) — the guide's placement-independent hand-off, no fences.
//   * Op kinds: GEMM (gather-GEMM over a rulebook table: MFMA 16x16x32 bf16 on the library's pre-packed "wide" weight
//     fragments; 16-row tiles; the offsets of a tile split over the waves of a workgroup and summed in a fixed order
//     through LDS; epilogue: residual add, ReLU mask of the BatchNorm in front of the conv (backward), bf16 rounding,
//     BatchNorm statistics partials of the stored values, coalesced 16-byte stores), BNFWD / BNBWD (each workgroup
//     reduces the G partial rows itself — same rows, same order, fp64 — and sweeps the rows it owns), STATS.
//   * Deterministic: static unit -> workgroup assignment, fixed summation orders.
#include "common.hpp"
#include "spconv_common.hpp"
#include <mutex>
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int CX_THREADS = 512;
constexpr int CX_WAVES = 8;
constexpr int CX_MAXC = 256;         // channels of any tensor an op touches
constexpr int CX_MAXG = 64;          // workgroups (statistics rows) of a call
constexpr int CX_SC1 = 16;           // aux bits of a raw buffer access: sc1 (agent scope: write-through store / L1-bypass load)
constexpr unsigned CX_SPIN_LIMIT = 1u << 21;

// LDS map (dynamic)
constexpr int CX_CHUNK_OFF = 0;                                 // a chunk of operands (<= 80 KB); after a unit's chunk loop:
constexpr int CX_RED_BYTES = CX_WAVES * 8 * 64 * 16;            //   f32x4 [8 waves][8 blocks][64 lanes] partial accumulators (64 KB)
constexpr int CX_OUT_BYTES = 128 * 128 * 2;                     //   + bf16 [128 rows][nbu * 16] output staging (32 KB)
constexpr int CX_CHUNK_BYTES = CX_RED_BYTES + CX_OUT_BYTES;     // 96 KB
constexpr int CX_IDX_OFF = CX_CHUNK_OFF + CX_CHUNK_BYTES;       // int [27][128]: the unit's slice of the table
constexpr int CX_IDX_BYTES = 27 * 128 * 4;                      // 13.5 KB
constexpr int CX_VEC_OFF = CX_IDX_OFF + CX_IDX_BYTES;           // float [6][CX_MAXC]
constexpr int CX_VEC_BYTES = 6 * CX_MAXC * 4;
constexpr int CX_WGST_OFF = CX_VEC_OFF + CX_VEC_BYTES;          // float [2][CX_MAXC]: this workgroup's statistics row of a GEMM
constexpr int CX_WGST_BYTES = 2 * CX_MAXC * 4;
constexpr int CX_DEC_OFF = CX_WGST_OFF + CX_WGST_BYTES;         // int [CX_LPT * 8]: what fragment f of a chunk is (cx_gemm)
constexpr int CX_DEC_BYTES = 512;
constexpr int CX_EPI_OFF = CX_DEC_OFF + CX_DEC_BYTES;         // bf16 [128 rows][nbu * 16]: the unit's tile of the epilogue operand
constexpr int CX_EPI_BYTES = 128 * 128 * 2;                     //   (forward: residual; data gradient: the BatchNorm's input) 32 KB
constexpr int CX_RED_OFF = CX_CHUNK_OFF;                        // (row-local ops: scratch)
constexpr int CX_LDS_BYTES = CX_EPI_OFF + CX_EPI_BYTES;         // ~150 KB

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t cx_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7ffffff0, 0x00020000);   // (masking by the explicit OOB offset)
}
__device__ __forceinline__ float cx_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// debug stamps (tools/cxstamps.py): workgroup 0, thread 0 appends (label << 56 | 100 MHz wall clock) to a caller-provided buffer
__device__ unsigned long long *g_cx_stamps = nullptr;
__device__ __forceinline__ void cx_stamp(int me, unsigned label) {
    unsigned long long *b = g_cx_stamps;
    if (b && me == 0 && threadIdx.x == 0) {
        const unsigned long long k = b[0];
        if (k < 4000) { b[1 + k] = ((unsigned long long)label << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull); b[0] = k + 1; }
    }
}

struct CxCtl {
    unsigned *ctr, *err;
    unsigned target;     // value the counter reaches when every workgroup has arrived at the NEXT barrier
    int G;
    bool dead;
};

// All workgroups of the call meet here.  Every wave first drains its own write-through stores.
__device__ __forceinline__ void cx_barrier(CxCtl &c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    doda_sync();
    c.target += (unsigned)c.G;
    if (threadIdx.x == 0 && !c.dead) {
        __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(c.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > CX_SPIN_LIMIT) {   // a workgroup that never arrives must not hang the device: flag and fall through
                __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.dead = true;
                break;
            }
        }
    }
    doda_sync();
    // (one lane polled; the others learn of a time-out at the next barrier through the flag: not needed — a dead call's
    // results are discarded by the host, it only has to terminate)
}

// 8 bf16 of a 16-byte chunk -> floats
__device__ __forceinline__ void cx_unpack8(const u32x4 &v, float (&f)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[2 * q] = __uint_as_float(v[q] << 16);
        f[2 * q + 1] = __uint_as_float(v[q] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 cx_pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (unsigned)f2bf(f[2 * q]) | ((unsigned)f2bf(f[2 * q + 1]) << 16);
    return v;
}

// ---- GEMM -------------------------------------------------------------------------------------------------------------
// A workgroup UNIT = (tpw consecutive 16-row tiles) x (nbu consecutive 16-channel output blocks), all K offsets.  What a
// single XCD's CU can ingest (~64 B/clk) is the bound here, and a load that is waited for one at a time costs a full L2 /
// fabric round trip (~1.2 us): a first version whose waves each walked their own (offset, chunk) units — one gathered
// operand + the weight fragments, wait, MFMAs — spent 1 ms per direction on levels 5-7.  So the operands travel in CHUNKS
// of `oc` offsets: all 512 threads request a chunk's weight fragments (coalesced 16-byte pieces of the pre-packed buffer)
// and gathered rows (already in MFMA operand order) into registers — TWO chunks in flight, up to 160 KB per CU —, park a
// chunk in LDS, and the eight waves multiply it from there: wave = (tile t, slice s of the chunk's (offset, k-chunk)
// units), the slices of a tile summed in a fixed order through LDS at the end.  The plan (tpw, nbu, oc) is chosen per op on
// the host (plan_gemm) so that every workgroup gets a unit and no unit ingests more than it must: many tiles -> 64-row
// units with every channel block (weights read once per 64 rows); few tiles -> the channel blocks spread over workgroups.
constexpr int CX_LPT = 8;                         // 16-byte loads per thread and chunk (chunk <= 64 KB)
constexpr int CX_CHUNK_ITEMS = CX_LPT * CX_THREADS;

struct GemmPlan { int tpw, nbu, oc; };
__host__ __device__ inline GemmPlan unpack_plan(int v) { return GemmPlan{v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff}; }
inline int pack_plan(const GemmPlan &p) { return p.tpw | (p.nbu << 8) | (p.oc << 16); }

// (ONE instantiation with the block count at run time: eight template instances made the kernel 171 KB of code against a
// 64 KB instruction cache, and every op began with ~2.5 us of instruction fetch)
constexpr int NBU_MAX = 8;
__device__ __forceinline__ void cx_gemm_units(const doda_cx_op &op, int me, int G, char *smem, const GemmPlan pl) {
    const int NBU = pl.nbu;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int n_out = op.rows, K = op.K, c_in = op.c_in, c_out = op.c_out;
    const int CC = (c_in + 31) >> 5, NB = c_out >> 4;
    const int T = (n_out + 15) >> 4;
    const bool identity = (op.flags & DODA_CX_F_IDENTITY) != 0;
    const bool bwd = op.aux != nullptr;          // ReLU mask + backward statistics of the BatchNorm in front of the conv
    const bool relu = (op.flags & DODA_CX_F_RELU) != 0;
    const int tpw = pl.tpw, oc = pl.oc;
    const int tsh = tpw == 1 ? 0 : tpw == 2 ? 1 : tpw == 4 ? 2 : 3;
    const int nsl = CX_WAVES >> tsh;              // slices of a chunk's units per tile
    const int MS = (T + tpw - 1) >> tsh, NS = (NB + NBU - 1) / NBU;
    const int n_units = MS * NS;
    const int NCH = (K + oc - 1) / oc;
    const int t_in = wave & (tpw - 1), sl = wave >> tsh;
    const int rows_u = tpw * 16;

    const rsrc_t rs_x = cx_rsrc(op.x), rs_w = cx_rsrc(op.w), rs_y = cx_rsrc(op.y);
    const rsrc_t rs_r = cx_rsrc(op.res ? op.res : op.x), rs_a = cx_rsrc(op.aux ? op.aux : op.x);
    const unsigned x_pitch = (unsigned)op.x_ld * 2u;
    u32x4 *cbuf = reinterpret_cast<u32x4 *>(smem + CX_CHUNK_OFF);
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + CX_CHUNK_OFF);                        // (after the chunk loop)
    unsigned short *outt = reinterpret_cast<unsigned short *>(smem + CX_CHUNK_OFF + CX_RED_BYTES);
    int *strip = reinterpret_cast<int *>(smem + CX_IDX_OFF);
    const float *vec = reinterpret_cast<const float *>(smem + CX_VEC_OFF);
    float *wgst = reinterpret_cast<float *>(smem + CX_WGST_OFF);
    unsigned short *epit = reinterpret_cast<unsigned short *>(smem + CX_EPI_OFF);

    const int nWf = oc * CC * NBU;                            // 1 KB fragments of a chunk: weights [oc][CC][NBU], then gathered rows [oc][tpw][CC]
    int dec[CX_LPT];
#pragma unroll
    for (int q = 0; q < CX_LPT; ++q) dec[q] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(smem + CX_DEC_OFF)[q * CX_WAVES + wave]);
    for (int u = me; u < n_units; u += G) {
        const int mg = NS == 1 ? u : u / NS, ng = u - mg * NS;
        const int tile0 = mg * tpw, nb0 = ng * NBU;
        // request chunk c into R: fragment q * 8 + wave of the chunk (wave-uniform), one 16-byte piece per lane.  dec[q] says
        // what the fragment is — weights (offset in chunk << 16 | k-chunk << 8 | block), bit 30: gathered rows (offset in chunk
        // << 16 | k-chunk << 8 | tile), -1: nothing — decoded once per op into LDS (cx_gemm).  STRAIGHT-LINE code: one
        // unconditional load per q with a selected descriptor and offset (nothing to fetch = out-of-range offset), so the
        // nine table reads and the nine requests go out back to back; with a branch per q every request waited for its own
        // LDS read (1.8 us to issue 18 loads).
        auto issue = [&](int c, u32x4 (&R)[CX_LPT]) {
            const int o0 = c * oc;
            int id[CX_LPT];
#pragma unroll
            for (int q = 0; q < CX_LPT; ++q) {
                const int dq = __builtin_amdgcn_readfirstlane(dec[q]), o = o0 + ((dq >> 16) & 0xff), t = dq & (tpw - 1);
                id[q] = strip[(o < K ? o : K - 1) * rows_u + t * 16 + r];
            }
#pragma unroll
            for (int q = 0; q < CX_LPT; ++q) {
                // (readfirstlane: the fragment kind must be PROVABLY wave-uniform where it selects the buffer descriptor — kept in a
                // VGPR the select became a waterfall loop per request: 5 us per chunk with no memory traffic at all)
                const int dq = __builtin_amdgcn_readfirstlane(dec[q]), lo = dq & 0xff, cc = (dq >> 8) & 0xff, o = o0 + ((dq >> 16) & 0xff);
                const bool live = dq >= 0 && c < NCH && o < K, is_a = (dq & 0x40000000) != 0;
                const int c0 = cc * 32 + g * 8;
                const unsigned w_off = ((unsigned)((o * CC + cc) * NB + nb0 + lo) * 64u + (unsigned)lane) * 16u;
                const unsigned a_off = (unsigned)id[q] * x_pitch + (unsigned)c0 * 2u;
                const bool ok = live && (is_a ? (id[q] >= 0 && c0 < c_in) : (nb0 + lo < NB));
                const unsigned off = ok ? (is_a ? a_off : w_off) : OOB;
                R[q] = __builtin_amdgcn_raw_buffer_load_b128(is_a ? rs_x : rs_w, off, 0, CX_SC1);
            }
        };
        auto park = [&](u32x4 (&R)[CX_LPT]) {
#pragma unroll
            for (int q = 0; q < CX_LPT; ++q) {
                cbuf[(q * CX_WAVES + wave) * 64 + lane] = R[q];   // (unconditional: a branch per piece made hipcc wait vmcnt(0) — for
                //                                                    the NEWER chunk in flight as well — before every write)
            }
        };
        u32x4 Ra[CX_LPT], Rb[CX_LPT];
        cx_stamp(me, 10);
        // the epilogue's operand tile (residual rows, or the BatchNorm input of a data-gradient call): requested now, read
        // from LDS when the accumulators are ready
        const int nbu_here = NB - nb0 < NBU ? NB - nb0 : NBU;
        const bool has_epi = op.res != nullptr || bwd;
        const int epi_ld = bwd ? op.aux_ld : op.res_ld;
        u32x4 ev[4];
        if (has_epi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = q * CX_THREADS + tid, cpr = nbu_here * 2;
                const int rr = i / cpr, ck = i - rr * cpr, row = tile0 * 16 + rr;
                const bool ok = rr < rows_u && row < n_out;
                ev[q] = __builtin_amdgcn_raw_buffer_load_b128(bwd ? rs_a : rs_r, ok ? ((unsigned)row * (unsigned)epi_ld + (unsigned)(nb0 * 16 + ck * 8)) * 2u : OOB, 0, CX_SC1);
            }
        }
        doda_sync();                                       // (strip / chunk buffer of the previous unit are done with)
        for (int e = tid; e < K * rows_u; e += CX_THREADS) {   // the unit's slice of the table
            const int o = e >> (4 + tsh), rr = e & (rows_u - 1), row = tile0 * 16 + rr;
            int v = -1;
            if (row < n_out) v = identity ? row : op.tbl[(long long)o * op.tbl_ld + row];
            strip[e] = v;
        }
        doda_sync();
        cx_stamp(me, 11);
        f32x4 acc[NBU_MAX];
#pragma unroll
        for (int j = 0; j < NBU_MAX; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto multiply = [&](int c) {
            const int n_o = (c + 1) * oc <= K ? oc : K - c * oc;
            int o_l = 0, cc = sl;
            for (int uu = sl; uu < n_o * CC; uu += nsl, cc += nsl) {
                while (cc >= CC) { cc -= CC; ++o_l; }
                const u32x4 xa = cbuf[(nWf + (o_l * tpw + t_in) * CC + cc) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NBU_MAX; ++j)
                    if (j < NBU) mma_bf16_k32(acc[j], cbuf[((o_l * CC + cc) * NBU + j) * 64 + lane], xa);
            }
        };
        // chunk c is multiplied from LDS while chunks c + 1 and c + 2 are in flight (the loop starts two chunks early: its
        // first two trips only request)
        for (int c = -2; c < NCH; c += 2) {
            if (c >= 0) {
                doda_sync();        // every wave is done with the chunk in LDS
                park(Ra);
            }
            issue(c + 2, Ra);
            if (c >= 0) {
                doda_sync();
                multiply(c);
            }
            if (c + 1 >= 0 && c + 1 < NCH) {
                doda_sync();
                park(Rb);
            }
            issue(c + 3, Rb);
            if (c == -2 && has_epi) {   // (the epilogue operand tile: parked once the first two chunks are on their way)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = q * CX_THREADS + tid, cpr = nbu_here * 2;
                    const int rr = i / cpr, ck = i - rr * cpr;
                    if (rr < rows_u) *reinterpret_cast<u32x4 *>(epit + rr * (NBU * 16) + ck * 8) = ev[q];
                }
            }
            if (c + 1 >= 0 && c + 1 < NCH) {
                doda_sync();
                multiply(c + 1);
            }
        }
        doda_sync();                // the chunk buffer becomes the reduction / output staging area
        cx_stamp(me, 16);
#pragma unroll
        for (int j = 0; j < NBU_MAX; ++j)
            if (j < NBU) red[(wave * 8 + j) * 64 + lane] = acc[j];
        doda_sync();
        // epilogue: wave j finishes channel block nb0 + j of every tile of the unit
        if (wave < nbu_here) {
            const int ch = (nb0 + wave) * 16 + g * 4;
            f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < tpw; ++t) {
                f32x4 v = red[(t * 8 + wave) * 64 + lane];
                for (int s2 = 1; s2 < nsl; ++s2) v += red[((s2 * tpw + t) * 8 + wave) * 64 + lane];
                const int row = (tile0 + t) * 16 + r;
                const bool ok = row < n_out;
                (void)ok;
                const u32x2 et = has_epi ? *reinterpret_cast<const u32x2 *>(epit + (t * 16 + r) * (NBU * 16) + wave * 16 + g * 4) : (u32x2){0u, 0u};
                if (op.res) {
                    const u32x2 rr = et;
                    v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
                    v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
                }
                f32x4 xh = {0.f, 0.f, 0.f, 0.f};
                if (bwd) {
                    const u32x2 xr = et;
                    const float xv[4] = {__uint_as_float(xr[0] << 16), __uint_as_float(xr[0] & 0xffff0000u),
                                         __uint_as_float(xr[1] << 16), __uint_as_float(xr[1] & 0xffff0000u)};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        xh[q] = (xv[q] - vec[ch + q]) * vec[CX_MAXC + ch + q];
                        if (relu) {
                            const float yv = xh[q] * vec[2 * CX_MAXC + ch + q] + vec[3 * CX_MAXC + ch + q];
                            v[q] = yv > 0.f ? v[q] : 0.f;
                        }
                    }
                }
                unsigned short ob[4];
                f32x4 vr;
#pragma unroll
                for (int q = 0; q < 4; ++q) { ob[q] = f2bf(v[q]); vr[q] = cx_bf2f(ob[q]); }
                if (op.stats) {
                    f32x4 a = vr, b = bwd ? vr * xh : vr * vr;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[q] = row_sum16(a[q]); b[q] = row_sum16(b[q]); }
                    st1 += a; st2 += b;
                }
                u32x2 pk;
                pk[0] = (unsigned)ob[0] | ((unsigned)ob[1] << 16);
                pk[1] = (unsigned)ob[2] | ((unsigned)ob[3] << 16);
                *reinterpret_cast<u32x2 *>(outt + (t * 16 + r) * (NBU * 16) + wave * 16 + g * 4) = pk;
            }
            if (op.stats && r == 15) {   // one writer per channel and unit, units in program order: deterministic
#pragma unroll
                for (int q = 0; q < 4; ++q) { wgst[ch + q] += st1[q]; wgst[CX_MAXC + ch + q] += st2[q]; }
            }
        }
        doda_sync();
        cx_stamp(me, 17);
        {   // the unit's rows x channel blocks in 16-byte write-through stores
            const int cpr = nbu_here * 2;
            for (int i = tid; i < rows_u * cpr; i += CX_THREADS) {
                const int rr = i / cpr, ck = i - rr * cpr;
                const int row = tile0 * 16 + rr;
                if (row < n_out) {
                    const u32x4 d = *reinterpret_cast<const u32x4 *>(outt + rr * (NBU * 16) + ck * 8);
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, ((unsigned)row * (unsigned)op.y_ld + (unsigned)(nb0 * 16 + ck * 8)) * 2u, 0, CX_SC1);
                }
            }
        }
    }
}

__device__ __forceinline__ void cx_gemm(const doda_cx_op &op, int me, int G, char *smem) {
    float *vec = reinterpret_cast<float *>(smem + CX_VEC_OFF);
    float *wgst = reinterpret_cast<float *>(smem + CX_WGST_OFF);
    const int c = threadIdx.x;
    if (c < CX_MAXC) { wgst[c] = 0.f; wgst[CX_MAXC + c] = 0.f; }
    if (op.aux && c < op.c_out) {   // the BatchNorm in front of the conv (backward epilogue): its vectors -> LDS
        vec[c] = op.mean[c];
        vec[CX_MAXC + c] = op.invstd[c];
        vec[2 * CX_MAXC + c] = op.gamma[c];
        vec[3 * CX_MAXC + c] = op.beta[c];
    }
    cx_stamp(me, 20);
    const GemmPlan pl = unpack_plan(op.reserved);
    if (c < CX_LPT * CX_WAVES) {   // fragment c of a chunk: weights [oc][CC][nbu], then gathered rows [oc][tpw][CC]
        const int CC = (op.c_in + 31) >> 5, nWf = pl.oc * CC * pl.nbu, nAf = pl.oc * pl.tpw * CC;
        int d = -1;
        if (c < nWf) { const int j = c % pl.nbu, f2 = c / pl.nbu; d = ((f2 / CC) << 16) | ((f2 % CC) << 8) | j; }
        else if (c < nWf + nAf) { const int fa = c - nWf, f2 = fa / CC; d = 0x40000000 | ((f2 / pl.tpw) << 16) | ((fa % CC) << 8) | (f2 % pl.tpw); }
        reinterpret_cast<int *>(smem + CX_DEC_OFF)[c] = d;
    }
    doda_sync();
    cx_stamp(me, 21);
    if (pl.nbu >= 1 && pl.nbu <= NBU_MAX) cx_gemm_units(op, me, G, smem, pl);
    doda_sync();
    if (op.stats) {   // this workgroup's partial row: what it accumulated over its units, zeros elsewhere
        const rsrc_t rs_s = cx_rsrc(op.stats);
        for (int e = threadIdx.x; e < 2 * op.c_out; e += CX_THREADS) {
            const int h = e >= op.c_out ? 1 : 0, ch = e - h * op.c_out;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wgst[h * CX_MAXC + ch]), rs_s, ((unsigned)(me * 2 + h) * (unsigned)op.c_out + (unsigned)ch) * 4u, 0, CX_SC1);
        }
    }
}

// The G partial rows [G][2][cw] of a statistics array, columns [0, cw) -> LDS `dst` (same layout) with every thread
// requesting 16-byte pieces at once: ONE round trip (a loop of dependent 4-byte loads cost 64 of them: 10 us per op).
__device__ __forceinline__ void cx_fetch_partials(const float *sp, int n_part, int cw, float *dst) {
    const rsrc_t rs = cx_rsrc(sp);
    const int n16 = (n_part * 2 * cw) >> 2;
    for (int i = threadIdx.x; i < n16; i += CX_THREADS)
        reinterpret_cast<u32x4 *>(dst)[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)i * 16u, 0, CX_SC1);
}

// rows [r0, r1) this workgroup owns in the row-local ops
__device__ __forceinline__ void cx_own_rows(int rows, int me, int G, int &r0, int &r1) {
    const int per = (rows + G - 1) / G;
    r0 = me * per < rows ? me * per : rows;
    r1 = r0 + per < rows ? r0 + per : rows;
}

// ---- BatchNorm(+ReLU) forward: y = [relu]((x - mean) * invstd * gamma + beta) ------------------------------------------
__device__ __forceinline__ void cx_bnfwd(const doda_cx_op &op, int me, int G, char *smem) {
    float *vec = reinterpret_cast<float *>(smem + CX_VEC_OFF);
    const int C = op.c_in, rows = op.rows, tid = threadIdx.x;
    float *part = reinterpret_cast<float *>(smem + CX_RED_OFF);   // [G][2][ca] then [G][2][cb]
    const bool training = (op.flags & DODA_CX_F_TRAINING) != 0;
    const int ca = op.c_split;
    if (training) {
        cx_fetch_partials(op.stats, op.n_part, ca, part);
        if (ca < C) cx_fetch_partials(op.stats_b, op.n_part, C - ca, part + op.n_part * 2 * ca);
    }
    float g_ = 0.f, b_ = 0.f, rm_ = 0.f, rv_ = 0.f;
    if (tid < C) {   // (requested before the partial rows are waited for)
        g_ = op.gamma[tid]; b_ = op.beta[tid];
        if (op.running_mean && (me == 0 || !training)) { rm_ = op.running_mean[tid]; rv_ = op.running_var[tid]; }
    }
    doda_sync();
    if (tid < C) {
        float mu, is;
        if (training) {
            const float *sp = tid < ca ? part : part + op.n_part * 2 * ca;
            const int cw = tid < ca ? ca : C - ca, cl = tid < ca ? tid : tid - ca;
            double s1 = 0.0, s2 = 0.0;
            for (int p = 0; p < op.n_part; ++p) {
                s1 += (double)sp[(p * 2) * cw + cl];
                s2 += (double)sp[(p * 2 + 1) * cw + cl];
            }
            const double d = s1 / rows;
            double var = s2 / rows - d * d;
            if (var < 0.0) var = 0.0;
            mu = (float)d;
            is = (float)(1.0 / sqrt(var + (double)op.eps));
            if (me == 0) {
                if (op.running_mean) {
                    const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
                    const double mom = (double)op.momentum;
                    op.running_mean[tid] = (float)((1.0 - mom) * (double)rm_ + mom * d);
                    op.running_var[tid] = (float)((1.0 - mom) * (double)rv_ + mom * unbiased);
                }
                if (tid == 0 && op.nbt) *op.nbt += 1;
            }
        } else {
            mu = rm_;
            is = 1.0f / sqrtf(rv_ + op.eps);
        }
        if (me == 0 && op.mean) { op.mean[tid] = mu; op.invstd[tid] = is; }
        vec[tid] = mu;
        vec[CX_MAXC + tid] = is;
        vec[2 * CX_MAXC + tid] = g_;
        vec[3 * CX_MAXC + tid] = b_;
    }
    doda_sync();
    int r0, r1;
    cx_own_rows(rows, me, G, r0, r1);
    const int cpr = C >> 3;
    const bool relu = (op.flags & DODA_CX_F_RELU) != 0;
    const rsrc_t rs_x = cx_rsrc(op.x), rs_y = cx_rsrc(op.y);
    for (int i = tid; i < (r1 - r0) * cpr; i += CX_THREADS) {
        const int rr = i / cpr, ck = i - rr * cpr, row = r0 + rr, c0 = ck * 8;
        const u32x4 xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((unsigned)row * (unsigned)op.x_ld + (unsigned)c0) * 2u, 0, CX_SC1);
        float f[8];
        cx_unpack8(xv, f);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float o = (f[q] - vec[c0 + q]) * vec[CX_MAXC + c0 + q] * vec[2 * CX_MAXC + c0 + q] + vec[3 * CX_MAXC + c0 + q];
            if (relu) o = o > 0.f ? o : 0.f;
            f[q] = o;
        }
        __builtin_amdgcn_raw_buffer_store_b128(cx_pack8(f), rs_y, ((unsigned)row * (unsigned)op.y_ld + (unsigned)c0) * 2u, 0, CX_SC1);
    }
}

// ---- BatchNorm backward apply: dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) + add -------------------
// dz arrives with the ReLU mask applied (GEMM epilogue); stats = (sum dz, sum dz * xhat) partial rows.
__device__ __forceinline__ void cx_bnbwd(const doda_cx_op &op, int me, int G, char *smem) {
    float *vec = reinterpret_cast<float *>(smem + CX_VEC_OFF);
    const int C = op.c_in, rows = op.rows, tid = threadIdx.x;
    float *part = reinterpret_cast<float *>(smem + CX_RED_OFF);
    cx_fetch_partials(op.stats, op.n_part, C, part);
    float is_ = 0.f, mu_ = 0.f, ga_ = 0.f, dg_ = 0.f, db_ = 0.f;
    if (tid < C) {
        is_ = op.invstd[tid]; mu_ = op.mean[tid]; ga_ = op.gamma[tid];
        if (me == 0 && op.dgamma && (op.flags & DODA_CX_F_ACCUM)) { dg_ = op.dgamma[tid]; db_ = op.dbeta[tid]; }
    }
    doda_sync();
    if (tid < C) {
        double s1 = 0.0, s2 = 0.0;
        for (int p = 0; p < op.n_part; ++p) {
            s1 += (double)part[(p * 2) * C + tid];
            s2 += (double)part[(p * 2 + 1) * C + tid];
        }
        vec[tid] = mu_;
        vec[CX_MAXC + tid] = is_;
        vec[2 * CX_MAXC + tid] = ga_ * is_;
        vec[3 * CX_MAXC + tid] = (float)(s1 / rows);
        vec[4 * CX_MAXC + tid] = (float)(s2 / rows);
        if (me == 0 && op.dgamma) { op.dbeta[tid] = db_ + (float)s1; op.dgamma[tid] = dg_ + (float)s2; }
    }
    doda_sync();
    int r0, r1;
    cx_own_rows(rows, me, G, r0, r1);
    const int cpr = C >> 3, csp = op.c_split;
    const rsrc_t rs_z = cx_rsrc(op.x), rs_x = cx_rsrc(op.aux), rs_a = cx_rsrc(op.res ? op.res : op.x);
    const rsrc_t rs_y = cx_rsrc(op.y), rs_y2 = cx_rsrc(op.y2 ? op.y2 : op.y);
    for (int i = tid; i < (r1 - r0) * cpr; i += CX_THREADS) {
        const int rr = i / cpr, ck = i - rr * cpr, row = r0 + rr, c0 = ck * 8;
        const u32x4 zv = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ((unsigned)row * (unsigned)op.x_ld + (unsigned)c0) * 2u, 0, CX_SC1);
        const u32x4 xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((unsigned)row * (unsigned)op.aux_ld + (unsigned)c0) * 2u, 0, CX_SC1);
        u32x4 av = {0u, 0u, 0u, 0u};
        if (op.res) av = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ((unsigned)row * (unsigned)op.res_ld + (unsigned)c0) * 2u, 0, CX_SC1);
        float dz[8], x[8], ad[8];
        cx_unpack8(zv, dz);
        cx_unpack8(xv, x);
        cx_unpack8(av, ad);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float xh = (x[q] - vec[c0 + q]) * vec[CX_MAXC + c0 + q];
            float o = vec[2 * CX_MAXC + c0 + q] * (dz[q] - vec[3 * CX_MAXC + c0 + q] - xh * vec[4 * CX_MAXC + c0 + q]);
            if (op.res) o += ad[q];
            dz[q] = o;
        }
        if (c0 < csp) __builtin_amdgcn_raw_buffer_store_b128(cx_pack8(dz), rs_y, ((unsigned)row * (unsigned)op.y_ld + (unsigned)c0) * 2u, 0, CX_SC1);
        else __builtin_amdgcn_raw_buffer_store_b128(cx_pack8(dz), rs_y2, ((unsigned)row * (unsigned)op.y2_ld + (unsigned)(c0 - csp)) * 2u, 0, CX_SC1);
    }
}

// ---- (sum x, sum x^2) partial row of the rows this workgroup owns --------------------------------------------------------
__device__ __forceinline__ void cx_stats(const doda_cx_op &op, int me, int G, char *smem) {
    float *sred = reinterpret_cast<float *>(smem + CX_RED_OFF);
    const int C = op.c_in, rows = op.rows, tid = threadIdx.x;
    int r0, r1;
    cx_own_rows(rows, me, G, r0, r1);
    const int cpr = C >> 3, RL = CX_THREADS / cpr;
    const int col = tid % cpr, rl = tid / cpr;
    const rsrc_t rs_x = cx_rsrc(op.x);
    if (rl < RL) {
        float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int row = r0 + rl; row < r1; row += RL) {
            const u32x4 xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((unsigned)row * (unsigned)op.x_ld + (unsigned)col * 8u) * 2u, 0, CX_SC1);
            float f[8];
            cx_unpack8(xv, f);
#pragma unroll
            for (int q = 0; q < 8; ++q) { s1[q] += f[q]; s2[q] += f[q] * f[q]; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sred[(rl * 2 + 0) * C + col * 8 + q] = s1[q];
            sred[(rl * 2 + 1) * C + col * 8 + q] = s2[q];
        }
    }
    doda_sync();
    const rsrc_t rs_s = cx_rsrc(op.stats);
    for (int e = tid; e < 2 * C; e += CX_THREADS) {
        float t = 0.f;
        for (int k = 0; k < RL; ++k) t += sred[k * 2 * C + e];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rs_s, ((unsigned)me * 2u * (unsigned)C + (unsigned)e) * 4u, 0, CX_SC1);
    }
    doda_sync();
}

__global__ __launch_bounds__(CX_THREADS) void coarse_exec(const doda_cx_op *__restrict__ ops, int n_ops, unsigned *sync, int G, int xcds) {
    if ((int)(blockIdx.x & 7) >= xcds) return;
    const int me = (int)(blockIdx.x >> 3) * xcds + (int)(blockIdx.x & 7);
    if (me >= G) return;
    extern __shared__ __attribute__((aligned(16))) char cx_smem[];
    CxCtl ctl{sync, sync + 1, 0u, G, false};   // (the counter is zero at launch: the last workgroup out resets it, below)
    // The op in hand lives in REGISTERS (wave-uniform values): read through a reference to global memory, every field access
    // inside a loop was a vector load followed by s_waitcnt vmcnt(0) — which also drained the operand chunks in flight (a
    // chunk round cost 5 us with no memory traffic of its own).  The next op's 216 bytes are requested while this one runs and
    // handed over through LDS.
    __shared__ int opbuf[2][64];
    constexpr int OPW = (int)(sizeof(doda_cx_op) / 4);
    if (threadIdx.x < OPW) opbuf[0][threadIdx.x] = reinterpret_cast<const int *>(ops)[threadIdx.x];
    for (int i = 0; i < n_ops; ++i) {
        int nxt = 0;
        if (threadIdx.x < OPW && i + 1 < n_ops) nxt = reinterpret_cast<const int *>(ops + i + 1)[threadIdx.x];
        doda_sync();
        struct alignas(8) { int w[OPW]; } raw;
#pragma unroll
        for (int k = 0; k < OPW; ++k) raw.w[k] = __builtin_amdgcn_readfirstlane(opbuf[i & 1][k]);
        doda_cx_op op;
        __builtin_memcpy(&op, &raw, sizeof(op));
        cx_stamp(me, 1);
        if (op.flags & DODA_CX_F_BARRIER) cx_barrier(ctl);
        cx_stamp(me, 2);
        switch (op.kind) {
        case DODA_CX_GEMM: cx_gemm(op, me, G, cx_smem); break;
        case DODA_CX_BNFWD: cx_bnfwd(op, me, G, cx_smem); break;
        case DODA_CX_BNBWD: cx_bnbwd(op, me, G, cx_smem); break;
        case DODA_CX_STATS: cx_stats(op, me, G, cx_smem); break;
        default: break;
        }
        cx_stamp(me, 3);
        if (threadIdx.x < OPW) opbuf[(i + 1) & 1][threadIdx.x] = nxt;
    }
    // the last workgroup to leave puts the barrier counter back to zero for the next launch (every workgroup is past its
    // last barrier once it has counted itself out; a launch whose barrier timed out still ends with a clean counter)
    doda_sync();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)G - 1u) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// pinned staging ring for the op-list upload (as optim.hip: grow-only slots, reuse guarded by an event)
struct Slot {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int ev_dev = -1;
    bool pending = false;
};
std::mutex g_mu;
Slot g_slot[4];
unsigned g_next = 0;
bool g_attr_set = false;

int env_int(const char *name, int dflt, int lo, int hi) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v < lo ? lo : v > hi ? hi : v;
}

bool bad_channels(int c) { return c <= 0 || c > CX_MAXC || (c % 8) != 0; }

// (tpw, nbu, oc) of a GEMM op: see cx_gemm_units.  Cost of a candidate = rounds of units x KB a unit ingests (its share of
// the weight fragments + the rows it gathers, about half of the table's slots being present) + a fixed per-unit part.
GemmPlan plan_gemm(const doda_cx_op &o, int G) {
    const int K = o.K, CC = (o.c_in + 31) / 32, NB = o.c_out / 16, T = (o.rows + 15) / 16;
    GemmPlan best{1, NB < 8 ? NB : 8, 1};
    double best_cost = 1e30;
    for (int tpw = 1; tpw <= 8; tpw *= 2)
        for (int nbu = 1; nbu <= 8 && nbu <= NB; ++nbu) {
            if (CC * (nbu + tpw) > CX_CHUNK_ITEMS / 64) continue;          // one offset must fit a chunk
            const int units = ((T + tpw - 1) / tpw) * ((NB + nbu - 1) / nbu);
            const int rounds = (units + G - 1) / G;
            const double kb = (double)K * CC * (nbu + 0.5 * tpw) + 48.0;
            const double cost = rounds * kb;
            if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && tpw > best.tpw)) { best_cost = cost; best.tpw = tpw; best.nbu = nbu; }
        }
    int oc = (CX_CHUNK_ITEMS / 64) / (CC * (best.nbu + best.tpw));
    best.oc = oc < 1 ? 1 : oc > K ? K : oc;
    return best;
}

}  // namespace

// debug: where a launch spends its time (tools/cxstamps.py).  buf_dev: uint64 [4001] device words, [0] = 0 before a launch; NULL: off.
extern "C" int doda_coarse_debug_stamps(void *buf_dev) {
    unsigned long long *p = (unsigned long long *)buf_dev;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cx_stamps), &p, sizeof(p)) == hipSuccess ? DODA_OK : DODA_ERR_LAUNCH;
}

extern "C" int32_t doda_coarse_workgroups(void) {
    static const int g = env_int("DODA_CX_WGS", 32, 1, CX_MAXG);
    return g;
}

extern "C" size_t doda_coarse_desc_bytes(int32_t n_ops) { return n_ops > 0 ? align_up((size_t)n_ops * sizeof(doda_cx_op), 256) : 0; }

extern "C" int doda_coarse_run(const doda_cx_op *ops_h, int32_t n_ops, void *desc_dev, size_t desc_bytes, uint32_t *sync_dev,
                               doda_stream_t stream) {
    if (n_ops == 0) return DODA_OK;
    if (n_ops < 0 || !ops_h || !desc_dev || !sync_dev) return DODA_ERR_INVALID;
    const size_t need = doda_coarse_desc_bytes(n_ops);
    if (desc_bytes < need) return DODA_ERR_WORKSPACE;
    const int G = doda_coarse_workgroups();
    static const int xcds = env_int("DODA_CX_XCDS", 1, 1, 8);
    for (int k = 0; k < n_ops; ++k) {
        const doda_cx_op &o = ops_h[k];
        if (o.rows < 0 || o.n_part != G) return DODA_ERR_INVALID;
        switch (o.kind) {
        case DODA_CX_GEMM: {
            const int NB = (o.c_out + 15) / 16;
            if (!o.x || !o.w || !o.y || (!o.tbl && !(o.flags & DODA_CX_F_IDENTITY))) return DODA_ERR_INVALID;
            if (bad_channels(o.c_in) || bad_channels(o.c_out) || o.c_out % 16 != 0 || o.c_in < 32) return DODA_ERR_UNSUPPORTED;
            (void)NB;
            if (o.K < 1 || o.K > 27 || ((o.flags & DODA_CX_F_IDENTITY) && o.K != 1)) return DODA_ERR_UNSUPPORTED;
            if (o.x_ld < o.c_in || o.y_ld < o.c_out || o.x_ld % 8 || o.y_ld % 8 || (o.res && (o.res_ld < o.c_out || o.res_ld % 4)) ||
                (o.aux && (o.aux_ld < o.c_out || o.aux_ld % 4 || !o.mean || !o.invstd || !o.gamma || !o.beta)))
                return DODA_ERR_INVALID;
            break;
        }
        case DODA_CX_BNFWD:
            if (!o.x || !o.y || !o.gamma || !o.beta || bad_channels(o.c_in) || o.x_ld % 8 || o.y_ld % 8) return DODA_ERR_INVALID;
            if ((o.flags & DODA_CX_F_TRAINING) ? (!o.stats || o.c_split <= 0 || o.c_split > o.c_in || (o.c_split < o.c_in && !o.stats_b) || !o.mean || !o.invstd)
                                               : (!o.running_mean || !o.running_var))
                return DODA_ERR_INVALID;
            break;
        case DODA_CX_BNBWD:
            if (!o.x || !o.aux || !o.y || !o.stats || !o.mean || !o.invstd || !o.gamma || bad_channels(o.c_in) || o.x_ld % 8 || o.aux_ld % 8 ||
                o.y_ld % 8 || (o.res && o.res_ld % 8) || o.c_split <= 0 || o.c_split > o.c_in || o.c_split % 8 ||
                (o.c_split < o.c_in && (!o.y2 || o.y2_ld % 8)) || (o.dgamma && !o.dbeta))
                return DODA_ERR_INVALID;
            break;
        case DODA_CX_STATS:
            if (!o.x || !o.stats || bad_channels(o.c_in) || o.x_ld % 8) return DODA_ERR_INVALID;
            break;
        default: return DODA_ERR_INVALID;
        }
    }
    hipStream_t s = as_stream(stream);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_attr_set) {
            if (hipFuncSetAttribute((const void *)coarse_exec, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS_BYTES) != hipSuccess)
                return DODA_ERR_LAUNCH;
            g_attr_set = true;
        }
        Slot &sl = g_slot[g_next++ & 3u];
        if (sl.pending) { (void)hipEventSynchronize(sl.ev); sl.pending = false; }
        if (sl.cap < need) {
            if (sl.host) (void)hipHostFree(sl.host);
            sl.cap = align_up(need, 4096) * 2;
            if (hipHostMalloc(&sl.host, sl.cap, hipHostMallocDefault) != hipSuccess) { sl.host = nullptr; sl.cap = 0; return DODA_ERR_NOMEM; }
        }
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        if (sl.ev && sl.ev_dev != cur_dev) { (void)hipEventDestroy(sl.ev); sl.ev = nullptr; }
        if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return DODA_ERR_LAUNCH;
        sl.ev_dev = cur_dev;
        memcpy(sl.host, ops_h, (size_t)n_ops * sizeof(doda_cx_op));
        for (int k = 0; k < n_ops; ++k) {
            doda_cx_op &o = ((doda_cx_op *)sl.host)[k];
            if (o.kind == DODA_CX_GEMM) o.reserved = pack_plan(plan_gemm(o, G));
        }
        if (hipMemcpyAsync(desc_dev, sl.host, (size_t)n_ops * sizeof(doda_cx_op), hipMemcpyHostToDevice, s) != hipSuccess) return DODA_ERR_LAUNCH;
        (void)hipEventRecord(sl.ev, s);
        sl.pending = true;
    }
    hipLaunchKernelGGL(coarse_exec, dim3((unsigned)(8 * ((G + xcds - 1) / xcds))), dim3(CX_THREADS), CX_LDS_BYTES, s, (const doda_cx_op *)desc_dev, n_ops,
                       (unsigned *)sync_dev, G, xcds);
    return doda_check_launch();
}
