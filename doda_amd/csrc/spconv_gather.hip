// Sparse-convolution arithmetic: gather -> MFMA -> single wide store per output row.
// fp32 and bf16 feature storage from one template.
//
// Replaces spconv v1.2's indice_conv / indice_conv_backward data path (per offset: sparse_gather
// kernel, cuBLAS mm, sparse_scatter_add kernel; reference call sites model/unet_block.py:26,29,
// 48,70,78).  MI355X-first design:
//   * OUTPUT-STATIONARY.  A wave owns S subtiles of 16 consecutive output rows x NBW blocks of 16
//     output channels.  Per kernel offset o it gathers rows tbl[o][t] of x straight into MFMA
//     operand layout (lane (row, g) loads the 4-channel quad g of its row: 16 B fp32 / 8 B bf16,
//     a whole 64 B / 32 B row per 4 lanes), multiplies by W[o] and accumulates in registers.
//     Every output row is written once: no gather/scatter buffers in HBM, no atomics, deterministic.
//   * The block's slice of the gather table is staged once into LDS with coalesced reads; each
//     wave then derives the set of offsets that have at least one present row (27-bit mask in an
//     SGPR) and walks only those — absent offsets cost nothing.
//   * All global loads of the next (offset, 16-channel chunk) unit are issued before the MFMAs of
//     the current one (two register buffer sets, no copies), so the in-order vmcnt wait of unit u
//     never covers unit u+1: gather latency overlaps MFMA + the other waves.
//   * Weights are pre-packed (tiny kernel, L2-resident result) into MFMA fragment order, rounded
//     to bf16 for the bf16 path, with the transposition / offset mirroring of the data-grad
//     layouts folded in, so a B fragment is one 16 B / 8 B load per lane.
//   * MFMA operands are swapped (W fragment first, x fragment second) so that each lane ends up
//     with 4 consecutive output channels of ONE row: a single 16 B / 8 B store per fragment.
//   * fp32: v_mfma_f32_16x16x4_f32 (exact fmaf chain); bf16: v_mfma_f32_16x16x16_bf16, fp32
//     accumulate, one RNE rounding at the store.
// Bound: HBM for table + feature bytes; MFMA only for the dense contraction.
#include <cstdio>
#include "common.hpp"
#include "tilebook.hpp"
#include "spconv_common.hpp"
#include "wgrad_pairs.hpp"
#include <stdlib.h>
#include <utility>
#include <type_traits>

namespace {

struct F32 {
    typedef float elem;
    typedef f32x4 frag;  // 4 channels of one row / 4 k-slots of a weight column
    static __device__ __forceinline__ frag zero() { return (frag){0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ elem from_float(float f) { return f; }
    static __device__ __forceinline__ float to_float(elem e) { return e; }
    static __device__ __forceinline__ void mma(f32x4 &acc, const frag &w, const frag &x) {
        mma_f32_k16<false>(acc, __builtin_bit_cast(u32x4, w), __builtin_bit_cast(u32x4, x));
    }
    static __device__ __forceinline__ void store4(elem *p, const f32x4 &v) {
        *reinterpret_cast<f32x4 *>(p) = v;
    }
};

struct BF16 {
    typedef unsigned short elem;
    typedef s16x4 frag;
    static __device__ __forceinline__ frag zero() { return (frag){0, 0, 0, 0}; }
    static __device__ __forceinline__ elem from_float(float f) { return f2bf(f); }
    static __device__ __forceinline__ float to_float(elem e) { return __uint_as_float((unsigned)e << 16); }
    static __device__ __forceinline__ void mma(f32x4 &acc, const frag &w, const frag &x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w, x, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(elem *p, const f32x4 &v) {
        s16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (short)f2bf(v[q]);
        *reinterpret_cast<s16x4 *>(p) = o;
    }
};

// packed[o][cc][nb][lane] = frag{ B_o[cc*16 + 4g + q][nb*16 + i] : q = 0..3 }, lane = 16g + i
// B_o = W[o] (layout 0, w: [K][kc][nc]), W[o]^T (1, w: [K][nc][kc]), W[K-1-o]^T (2).
template <class T>
__global__ __launch_bounds__(256) void pack_weights(const float *__restrict__ w, int K, int kc,
                                                    int nc, int n_chunk, int NB, int wl,
                                                    typename T::frag *__restrict__ packed) {
    const long long total = (long long)K * n_chunk * NB * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int nb = (int)(r % NB); r /= NB;
    const int cc = (int)(r % n_chunk);
    const int o = (int)(r / n_chunk);
    const int i = lane & 15, g = lane >> 4;
    const int col = nb * 16 + i;
    typename T::frag v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cc * 16 + 4 * g + q;
        float f = 0.f;
        if (c < kc && col < nc) {
            if (wl == 0) f = w[((long long)o * kc + c) * nc + col];
            else f = w[((long long)(wl == 2 ? K - 1 - o : o) * nc + col) * kc + c];
        }
        v[q] = T::from_float(f);
    }
    packed[e] = v;
}

// wide bf16 packing (PBF16W): packed[o][cc32][nb][lane] = 8 bf16 { B_o[cc*32 + 8g + q][nb*16 + i] }
__device__ __forceinline__ u32x4_t pack_wide_frag(const float *__restrict__ w, int K, int kc, int nc,
                                                  int wl, int o, int cc, int nb, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int col = nb * 16 + i;
    unsigned short h[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = cc * 32 + 8 * g + q;
        float f = 0.f;
        if (c < kc && col < nc) {   // (a weight tensor has far fewer than 2^31 elements: 32-bit indices)
            if (wl == 0) f = w[((unsigned)o * (unsigned)kc + (unsigned)c) * (unsigned)nc + (unsigned)col];
            else f = w[((unsigned)(wl == 2 ? K - 1 - o : o) * (unsigned)nc + (unsigned)col) * (unsigned)kc + (unsigned)c];
        }
        h[q] = f2bf(f);
    }
    u32x4_t r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (unsigned)h[2 * q] | ((unsigned)h[2 * q + 1] << 16);
    return r;
}

__global__ __launch_bounds__(256) void pack_weights_wide(const float *__restrict__ w, int K, int kc,
                                                         int nc, int n_chunk, int NB, int wl,
                                                         u32x4_t *__restrict__ packed, int pair) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair) {
        if (e >= (long long)K * NB * 32) return;
        const long long r = e >> 5;
        packed[e] = pack_wide_frag(w, K, kc, nc, wl, (int)(r / NB), 0, (int)(r % NB), (int)(e & 31));
        return;
    }
    const long long total = (long long)K * n_chunk * NB * 64;
    if (e >= total) return;
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int nb = (int)(r % NB); r /= NB;
    const int cc = (int)(r % n_chunk);
    const int o = (int)(r / n_chunk);
    packed[e] = pack_wide_frag(w, K, kc, nc, wl, o, cc, nb, lane);
}

// One launch for many weight tensors (all layers of a network, both layouts): block b looks up its
// descriptor through the inclusive block prefix `blk_end`.
struct PackDesc {
    const float *w;
    void *out;
    int K, kc, nc, layout, elem_bytes, n_chunk, NB, pad;
};

template <class T>
__device__ __forceinline__ void pack_one(const PackDesc &d, long long e) {
    const int lane = (int)(e & 63);
    const unsigned r = (unsigned)(e >> 6), r1 = r / (unsigned)d.NB;
    const int nb = (int)(r - r1 * (unsigned)d.NB);
    const int o = (int)(r1 / (unsigned)d.n_chunk), cc = (int)(r1 - (unsigned)o * (unsigned)d.n_chunk);
    const int i = lane & 15, g = lane >> 4;
    const int col = nb * 16 + i;
    typename T::frag v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cc * 16 + 4 * g + q;
        float f = 0.f;
        if (c < d.kc && col < d.nc) {
            if ((d.layout & 3) == 0) f = d.w[((long long)o * d.kc + c) * d.nc + col];
            else f = d.w[((long long)((d.layout & 3) == 2 ? d.K - 1 - o : o) * d.nc + col) * d.kc + c];
        }
        v[q] = T::from_float(f);
    }
    reinterpret_cast<typename T::frag *>(d.out)[e] = v;
}

__global__ __launch_bounds__(256) void pack_weights_multi(const PackDesc *__restrict__ descs,
                                                          const int *__restrict__ blk_end, int n_desc) {
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {  // first descriptor whose inclusive end exceeds this block
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x < blk_end[mid]) hi = mid; else lo = mid + 1;
    }
    const PackDesc d = descs[lo];
    const int first = lo == 0 ? 0 : blk_end[lo - 1];
    // (32-bit index arithmetic: a tensor's fragment count is far below 2^31, and the three 64-bit divisions per thread of the first
    // form — ~100 instructions each — were most of this kernel's 47 us for 16 bytes of output per thread)
    const unsigned e = (unsigned)(blockIdx.x - first) * 256u + threadIdx.x;
    if (d.layout & 0x20) {  // pair packing (kc == 16): [o][nb][32 slots] = lanes 0..31 of a wide fragment
        if (e >= (unsigned)d.K * (unsigned)d.NB * 32u) return;
        const unsigned r = e >> 5, o = r / (unsigned)d.NB;
        reinterpret_cast<u32x4_t *>(d.out)[e] =
            pack_wide_frag(d.w, d.K, d.kc, d.nc, d.layout & 3, (int)o, 0, (int)(r - o * (unsigned)d.NB), (int)(e & 31));
        return;
    }
    const unsigned total = (unsigned)d.K * (unsigned)d.n_chunk * (unsigned)d.NB * 64u;
    if (e >= total) return;
    if (d.layout & 0x10) {  // wide bf16 fragments
        const unsigned r = e >> 6, r1 = r / (unsigned)d.NB, nb = r - r1 * (unsigned)d.NB;
        const unsigned o = r1 / (unsigned)d.n_chunk, cc = r1 - o * (unsigned)d.n_chunk;
        reinterpret_cast<u32x4_t *>(d.out)[e] =
            pack_wide_frag(d.w, d.K, d.kc, d.nc, d.layout & 3, (int)o, (int)cc, (int)nb, (int)(e & 63));
    } else if (d.elem_bytes == 4) pack_one<F32>(d, e);
    else pack_one<BF16>(d, e);
}

constexpr int MAX_K = 27;

template <class T, int NBW, int S>
__global__ __launch_bounds__(256) void conv_gather(const typename T::elem *__restrict__ x, int kc,
                                                   const typename T::frag *__restrict__ wp, int nc,
                                                   int NB, const int32_t *__restrict__ tbl, int ld,
                                                   int K, int n_out,
                                                   typename T::elem *__restrict__ y, int vec_ok,
                                                   const typename T::elem *__restrict__ res) {
    typedef typename T::frag frag;
    typedef typename T::elem elem;
    constexpr int TM = 4 * 16 * S;  // rows per block
    __shared__ int32_t idx_tile[MAX_K * TM];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    // work item = (row tile, channel-block group), group fastest; contiguous items per XCD
    const int n_nbg = (NB + NBW - 1) / NBW;
    const int item = xcd_work_item(blockIdx.x, gridDim.x);
    const long long r0 = (long long)(item / n_nbg) * TM;
    const int nb0 = (item % n_nbg) * NBW;

    // ---- stage the block's table slice (coalesced) ----
    for (int e = threadIdx.x; e < K * TM; e += 256) {
        const int o = e / TM, r = e - o * TM;
        idx_tile[e] = (r0 + r < n_out) ? tbl[(long long)o * ld + r0 + r] : -1;
    }
    doda_sync();

    const int wrow = wid * 16 * S;  // this wave's first row inside the tile
    if (r0 + wrow >= n_out) return;

    // ---- offsets with at least one present row among this wave's rows ----
    unsigned int active = 0;
    for (int o = 0; o < K; ++o) {
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) any |= idx_tile[o * TM + wrow + s * 16 + i] >= 0;
        if (__ballot(any) != 0ull) active |= 1u << o;
    }
    active = __builtin_amdgcn_readfirstlane(active);

    f32x4 acc[S][NBW];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n_chunk = (kc + 15) / 16;

    auto load_unit = [&](int o, int cc, frag (&xa)[S], frag (&wb)[NBW]) {
        const int c0 = cc * 16 + 4 * g;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int idx = idx_tile[o * TM + wrow + s * 16 + i];
            xa[s] = T::zero();
            if (idx >= 0 && c0 < kc) {
                const elem *row = x + (long long)idx * kc + c0;
                if (vec_ok) {
                    xa[s] = *reinterpret_cast<const frag *>(row);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < kc) xa[s][q] = row[q];
                }
            }
        }
        const frag *wrow_p = wp + (((long long)o * n_chunk + cc) * NB + nb0) * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            if (nb0 + nb < NB) wb[nb] = wrow_p[nb * 64];
    };
    auto mma_unit = [&](const frag (&xa)[S], const frag (&wb)[NBW]) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int s = 0; s < S; ++s) T::mma(acc[s][nb], wb[nb], xa[s]);
    };
    // advance (o, cc) to the next unit; returns false when exhausted (wave-uniform)
    auto advance = [&](int &o, int &cc) -> bool {
        if (++cc < n_chunk) return true;
        cc = 0;
        if (active == 0) return false;
        o = __builtin_ctz(active);
        active &= active - 1;
        return true;
    };

    if (active != 0) {
        frag xa0[S], xa1[S], wb0[NBW], wb1[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { wb0[nb] = T::zero(); wb1[nb] = T::zero(); }
        int o = __builtin_ctz(active), cc = 0;
        active &= active - 1;
        load_unit(o, cc, xa0, wb0);
        for (;;) {
            const bool m1 = advance(o, cc);
            if (m1) load_unit(o, cc, xa1, wb1);
            mma_unit(xa0, wb0);
            if (!m1) break;
            const bool m0 = advance(o, cc);
            if (m0) load_unit(o, cc, xa0, wb0);
            mma_unit(xa1, wb1);
            if (!m0) break;
        }
    }

    // D[i = channel 4g+r][j = row]: lane (row = lane&15, g) holds 4 consecutive channels
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const long long t = r0 + wrow + s * 16 + i;
        if (t < n_out) {
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int col = (nb0 + nb) * 16 + 4 * g;
                if (nb0 + nb < NB) {
                    elem *p = y + t * nc + col;
                    if (res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < nc) acc[s][nb][r] += T::to_float(res[t * nc + col + r]);
                    }
                    if (vec_ok && col + 3 < nc) {
                        T::store4(p, acc[s][nb]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < nc) p[r] = T::from_float(acc[s][nb][r]);
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fast path (kc % 16 == 0, nc % 16 == 0, 16-byte aligned, buffers < 2 GB): the same algorithm with
// the instruction stream and the memory pipeline under explicit control.  Measured on MI355X the
// generic kernel is instruction-issue bound at 16..32 channels (~4000 instructions per 64-row
// wave) and, once that is fixed, latency bound: hipcc waits vmcnt(0) before every MFMA group of a
// software-pipelined loop, so nothing is ever in flight (guide §5.7 / T3+T4).  Hence:
//   * every global access is a raw BUFFER op: 32-bit offsets (one VALU add per gather) and
//     hardware range checking — an absent neighbour is an out-of-range offset that returns
//     zeros, a row past n_out is a dropped store: no exec-mask branches in the loop;
//   * phase 0 turns the wave's slice of the gather table into BYTE OFFSETS once (one multiply
//     per table entry, not per gather) in a wave-private LDS strip and derives the active-offset
//     mask from the same registers (ballot); weight fragments are addressed on the scalar side;
//   * the (offset, chunk) units run through a ring of D register sets filled by INLINE-ASM
//     buffer loads that hipcc does not count, with hand-placed `s_waitcnt vmcnt((D-1)*L)`:
//     D-1 units (L = S + NBW loads each) stay in flight across every MFMA group.  When the real
//     units run out the ring is topped up with an all-out-of-range dummy unit (returns zeros),
//     which keeps L constant and the loop branch-free.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 make_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;  // stride 0: raw buffer, byte offsets
    r[2] = bytes;                          // num_records
    r[3] = 0x00020000u;
    return r;
}

// Per-mode policy of the fast kernel.  CH = input channels per (offset, chunk) unit; every lane
// owns a quarter of the unit's bytes (FSZ = sizeof(raw)).
//   PF32 : fp32 features, 16 channels per unit, 4 x v_mfma_f32_16x16x4_f32
//   PBF16: bf16 features, 16 channels per unit, v_mfma_f32_16x16x16_bf16  (8-byte lane loads)
//   PBF16W: bf16 features, 32 channels per unit, v_mfma_f32_16x16x32_bf16 (16-byte lane loads):
//          half the gathers, weight loads and MFMAs per channel — these kernels are paced by
//          the NUMBER of 64-lane memory instructions through the per-CU texture path.
//   PBF16P: bf16 features, kc == 16: two offsets x 16 channels per unit (see the struct).
// Gathers: lanes whose row is absent (offset >= 2^31) are masked out of EXEC so the texture
// addresser does not spend a cycle per absent 4-lane quad; their destination is pre-zeroed.
// Lane 0 always stays active: a VMEM instruction whose EXEC is all-zero may not count in vmcnt,
// which would break the counted waits of the ring.  The mask arithmetic writes SCC, so "scc" is in
// the clobber list (without it the compiler kept an s_cmp result live across the block).
#define DODA_ASM_LOAD(INSTR)                                                                       \
    static __device__ __forceinline__ void load(raw &dst, unsigned voff, const u32x4 &rs, unsigned soff) { \
        asm volatile(INSTR " %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff));   \
    }                                                                                              \
    static __device__ __forceinline__ void gather(raw &dst, unsigned voff, const u32x4 &rs, unsigned soff) { \
        unsigned long long keep;                                                                   \
        dst = raw{};                                                                               \
        asm volatile("v_cmp_lt_i32 vcc, -1, %2\n\t"                                                \
                     "s_or_b64 vcc, vcc, 1\n\t"                                                    \
                     "s_and_saveexec_b64 %1, vcc\n\t" INSTR " %0, %2, %3, %4 offen\n\t"            \
                     "s_mov_b64 exec, %1"                                                          \
                     : "+v"(dst), "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff) : "vcc", "scc");     \
    }

struct PF32 {
    typedef float elem;
    typedef u32x4 raw;
    typedef F32 pack;  // weight pre-pack flavour
    static constexpr int CH = 16;
    static constexpr bool PAIR = false;
    DODA_ASM_LOAD("buffer_load_dwordx4")
    static __device__ __forceinline__ void mma(f32x4 &acc, const raw &w, const raw &x) { mma_f32_k16<false>(acc, w, x); }
};
// PF32S: the same kernel with every unit as two bf16 MFMAs on bf16 head / tail splits of both operands (spconv_common.hpp
// mma_f32_k16<true>): fp32 layers of many rows (run_gather: EpiArgs::f32_split)
struct PF32S : PF32 {
    static __device__ __forceinline__ void mma(f32x4 &acc, const raw &w, const raw &x) { mma_f32_k16<true>(acc, w, x); }
};
struct PBF16 {
    typedef unsigned short elem;
    typedef u32x2 raw;
    typedef BF16 pack;
    static constexpr int CH = 16;
    static constexpr bool PAIR = false;
    DODA_ASM_LOAD("buffer_load_dwordx2")
    static __device__ __forceinline__ void mma(f32x4 &acc, const raw &w, const raw &x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, w), __builtin_bit_cast(s16x4, x), acc, 0, 0, 0);
    }
};
struct PBF16W {
    typedef unsigned short elem;
    typedef u32x4 raw;
    typedef BF16 pack;
    static constexpr int CH = 32;
    static constexpr bool PAIR = false;
    DODA_ASM_LOAD("buffer_load_dwordx4")
    static __device__ __forceinline__ void mma(f32x4 &acc, const raw &w, const raw &x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
// PBF16P: bf16 features with exactly 16 input channels.  A unit is a PAIR of kernel offsets: the
// k = 32 of one v_mfma_f32_16x16x32_bf16 is (offset a | offset b) x 16 channels.  Lane groups
// g = 0,1 gather the two 16-byte halves of the row under offset a, g = 2,3 under offset b, and read
// the matching halves of W[a] / W[b] — half the gathers, weight loads and MFMAs of PBF16 for the
// layers that dominate the step (level-1 SubM 16 -> 16).
struct PBF16P : PBF16W {
    static constexpr int CH = 16;
    static constexpr bool PAIR = true;
};

// one wide output store: lane holds 4 consecutive channels of one row
template <class P, bool OUT32>
__device__ __forceinline__ void store_frag(const f32x4 &v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
    if (OUT32 || sizeof(typename P::elem) == 4) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
    } else {
        s16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (short)f2bf(v[q]);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), r, voff, 0, 0);
    }
}

// wait until at most N of the asm-issued loads are outstanding; `first` (and every register passed
// to touch() right after) is marked as written here so no consumer can be scheduled above it
template <int N, class R>
__device__ __forceinline__ void wait_vm(R &first) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(first) : "n"(N));
}
template <class R>
__device__ __forceinline__ void touch(R &r) {
    asm volatile("" : "+v"(r));
}

__device__ __forceinline__ float bf16_rounded(float f) {
    return __uint_as_float((unsigned)f2bf(f) << 16);
}

// SPLIT (few rows, S = NBW = 1): the block's four waves share ONE 16-row tile and one channel
// block; each takes every fourth active offset and the four partial accumulators are summed in a
// fixed order through LDS.  At the coarse U-Net levels a wave otherwise walks 80-190 units in
// series (27 offsets x 3..7 chunks, ~70 ns each) while the chip is nearly empty.
// STATS: the BatchNorm-statistics epilogue is a compile-time variant, so the plain kernels keep their
// register allocation and instruction stream.
template <class P, int NBW, int S, int D, bool OUT32, bool SPLIT = false, bool STATS = false, int PRE = 0>
__global__ __launch_bounds__(256) void conv_fast(const typename P::elem *__restrict__ x,
                                                 unsigned x_bytes, int kc,
                                                 const void *__restrict__ wp,
                                                 unsigned wp_bytes, int nc, int NB,
                                                 const int32_t *__restrict__ tbl,
                                                 unsigned tbl_bytes, int ld, int K, int n_out,
                                                 void *__restrict__ y, unsigned y_bytes,
                                                 const void *__restrict__ res, const EpiArgs ep, const PreArgs pre) {
    typedef typename P::elem elem;
    typedef typename P::raw raw;
    constexpr unsigned OSZ = OUT32 ? 4u : (unsigned)sizeof(elem);   // output element size
    constexpr int RW = 16 * S;                 // rows per wave
    constexpr int OPI = 256 / RW;              // table offsets fetched per (16-byte) load instruction
    constexpr int NLD = (MAX_K + 1 + OPI - 1) / OPI;  // strips 0..MAX_K; strip MAX_K is all-OOB
    constexpr int NOP = PRE >= 3 ? 3 : PRE == 2 ? 2 : 1;   // gathered operands per row (PRE >= 2: dz, the BatchNorm's input, the skip gradient)
    constexpr int L = S * NOP + NBW;           // asm loads per unit
    constexpr unsigned ESZ = sizeof(elem), FSZ = sizeof(raw);
    constexpr int CH = P::CH;
    static_assert((D - 1) * L <= 63, "vmcnt field");
    static_assert(PRE == 0 || (FSZ == 16 && !P::PAIR && !OUT32), "the folded BatchNorm works on 16-byte pieces of same-dtype rows");
    __shared__ __attribute__((aligned(16))) unsigned off_tile[4][(MAX_K + 1) * RW];
    // PRE: per-channel vectors of the folded BatchNorm (spconv_common.hpp pre_finish), zero past kc.  Everything the kernel's
    // prologue reads — the statistics totals, the gather table below, the first rows of the side output — is REQUESTED before
    // anything is waited for: one exposed round trip instead of three (a coarse-level kernel is its chain of round trips).
    constexpr int PESZ = PRE == 0 ? 2 : (int)ESZ, PKIND = PRE == 0 ? 1 : PRE;
    typedef PreForm<PESZ, PKIND> PF;
    __shared__ __attribute__((aligned(16))) float pre_co[PRE == 0 ? 1 : PF::NVEC][PRE == 0 ? 4 : PRE_MAX_C];
    PreRaw pre_raw;
    u32x4 sw_x = {0u, 0u, 0u, 0u}, sw_u = sw_x, sw_a = sw_x;     // side-output sweep, first piece of this thread
    unsigned sw_e = 0, sw_n = 0, sw_ppr = 1;                    // (piece numbers fit 32 bits: rows x row bytes < 2^31 on the fast path)
#ifdef DODA_PRE_ABLATE
    const int ablate = pre.kind >> 8;
#endif
    if constexpr (PRE != 0) {
#ifdef DODA_PRE_ABLATE
        if (ablate & 4) { pre_raw = PreRaw{}; } else
#endif
        pre_raw = pre_request<PRE>(pre, kc);
        constexpr unsigned NV = 16 / ESZ;
        sw_ppr = (unsigned)kc / NV;     // 16-byte pieces per row (kc % NV == 0 on this path)
        sw_n = (unsigned)pre.rows * sw_ppr;
        sw_e = blockIdx.x * 256u + threadIdx.x;
        {   // requested unconditionally (a thread without a piece re-reads piece 0): no branch, no wait before the table loads
            const unsigned e = sw_e < sw_n ? sw_e : 0u;
            const unsigned r = e / sw_ppr, c0 = (e - r * sw_ppr) * NV;
            sw_x = *reinterpret_cast<const u32x4 *>(x + (size_t)r * (ep.x_ld ? ep.x_ld : (unsigned)kc) + c0);
            if constexpr (PRE >= 2) sw_u = *reinterpret_cast<const u32x4 *>((const elem *)pre.aux + (size_t)r * pre.aux_ld + c0);
            if constexpr (PRE >= 3) sw_a = *reinterpret_cast<const u32x4 *>((const elem *)pre.add + (size_t)r * pre.add_ld + c0);
        }
    }

    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int n_nbg = (NB + NBW - 1) / NBW;
    const int item = xcd_work_item(blockIdx.x, gridDim.x);
    const int nb0 = (item % n_nbg) * NBW;
    const int row0 = SPLIT ? (item / n_nbg) * RW : ((item / n_nbg) * 4 + wid) * RW;   // wave's first output row

    const u32x4 rs_x = make_rsrc(x, x_bytes), rs_w = make_rsrc(wp, wp_bytes);
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void *)tbl, 0, tbl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, y_bytes, 0x00020000);

    // ---- phase 0: table slice -> byte offsets in LDS, active-offset mask ----
    // 16-byte table loads: a lane takes 4 consecutive rows of one offset, so one instruction covers
    // OPI = 256 / RW offsets (the kernel is paced by the NUMBER of vector-memory instructions; this
    // is 4 instead of 14 per 32-row wave).  Strip MAX_K stays all-out-of-range (the dummy unit).
    const unsigned row_bytes = (ep.x_ld ? ep.x_ld : (unsigned)kc) * ESZ;
    unsigned active = 0;
    {
        constexpr int LPO = RW / 4;            // lanes per offset
        const int rq = (lane % LPO) * 4, oq = lane / LPO;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int o = j * OPI + oq;
            const bool ok = o < K && row0 + rq < n_out;
            const unsigned voff = ok ? ((unsigned)o * (unsigned)ld + (unsigned)(row0 + rq)) * 4u : OOB;
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_t, voff, 0, 0);
            u32x4 off;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool present = ok && row0 + rq + k < n_out && (int)t[k] >= 0;
                off[k] = present ? (PRE >= 2 ? t[k] : t[k] * row_bytes) : OOB;   // (PRE >= 2: the row INDEX — three operands, three row strides)
                any |= present;
            }
            if (o <= MAX_K) *reinterpret_cast<u32x4 *>(&off_tile[wid][o * RW + rq]) = off;
            const unsigned long long b = __ballot(any);
#pragma unroll
            for (int q = 0; q < OPI; ++q) {
                const unsigned long long part = (b >> (q * LPO)) & ((1ull << LPO) - 1ull);
                if (part != 0ull && j * OPI + q < 32) active |= 1u << (j * OPI + q);
            }
        }
    }
    active = __builtin_amdgcn_readfirstlane(active);
#ifdef DODA_PRE_ABLATE
    if constexpr (PRE != 0) {
        if (ablate & 4) { for (int v = 0; v < PF::NVEC; ++v) pre_co[v][threadIdx.x] = 0.5f; }
        else pre_finish<(int)ESZ, PRE>(pre, kc, pre_raw, pre_co);
    }
#else
    if constexpr (PRE != 0) pre_finish<(int)ESZ, PRE>(pre, kc, pre_raw, pre_co);
#endif
    if (SPLIT) {  // keep the offsets whose rank among the active ones is wid (mod 4)
        unsigned mine = 0;
        int rank = 0;
        for (unsigned m = active; m != 0; m &= m - 1, ++rank)
            if ((rank & 3) == wid) mine |= m & (0u - m);
        active = __builtin_amdgcn_readfirstlane(mine);
    }
    doda_sync();

    if constexpr (PRE != 0) {
        // the transformed rows ONCE per row, for the weight gradient (forward: the BatchNorm's output; backward: the gradient of
        // its input): the launch's workgroups share the rows of x evenly, each piece goes through the same pre_piece as the gathers
        constexpr unsigned NV = 16 / ESZ;
        const unsigned xl = ep.x_ld ? ep.x_ld : (unsigned)kc;
#ifdef DODA_PRE_ABLATE
        if (!(ablate & 1))
#endif
        for (unsigned e = sw_e; e < sw_n; e += gridDim.x * 256u) {
            const unsigned r = e / sw_ppr, c0 = (e - r * sw_ppr) * NV;
            if (e != sw_e) {     // (the first piece was requested in the prologue)
                sw_x = *reinterpret_cast<const u32x4 *>(x + (size_t)r * xl + c0);
                if constexpr (PRE >= 2) sw_u = *reinterpret_cast<const u32x4 *>((const elem *)pre.aux + (size_t)r * pre.aux_ld + c0);
                if constexpr (PRE >= 3) sw_a = *reinterpret_cast<const u32x4 *>((const elem *)pre.add + (size_t)r * pre.add_ld + c0);
            }
            PreCo<(int)ESZ, PRE> cv;
            pre_load_co<(int)ESZ, PRE>(pre_co, (int)c0, cv);
            *reinterpret_cast<u32x4 *>((elem *)pre.side + (size_t)r * pre.side_ld + c0) = pre_piece<(int)ESZ, PRE>(sw_x, sw_u, sw_a, cv, pre.relu, ~0u);
        }
    }

    f32x4 acc[S][NBW];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr bool PAIR = P::PAIR;
    const int n_chunk = PAIR ? 1 : (kc + CH - 1) / CH;
    // this lane's share of a unit's bytes / its first channel inside the unit / its slot inside a
    // packed W fragment.  PAIR: groups g = 2,3 belong to the unit's second offset.
    const unsigned lane_x = (unsigned)(PAIR ? (g & 1) : g) * FSZ;
    const int lane_c = g * (CH / 4);
    const unsigned lane_w = (unsigned)(PAIR ? ((g & 1) * 16 + i) : lane) * FSZ;
    const bool second = PAIR && g >= 2;
    const unsigned *my_off = &off_tile[wid][i];

    // (round 6, tried and REMOVED: requesting the epilogue's operands — residual row, BatchNorm input and vectors — here, in front of
    // the unit loop of the 16-row split blocks.  8.4 -> 8.2 us per launch, and unsafe: the ring below is inline asm whose loads hipcc
    // does not see; it sank the compiler-visible requests BETWEEN ring loads in one instantiation (PBF16P, statistics), which breaks the
    // hand-counted vmcnt waits (found in the ISA while chasing a flaky two-rank test whose cause turned out to be elsewhere).  Every compiler-visible vector
    // load of this kernel must be consumed before the first asm load or issued after the ring has drained.)
    const int n_units = PAIR ? (__builtin_popcount(active) + 1) / 2 : __builtin_popcount(active) * n_chunk;
    if (n_units > 0) {
        raw xa[D][S], wb[D][NBW];
        // PRE: the second / third gathered operand of a row, its presence mask and the unit's channel chunk, per ring slot
        raw ua[D][PRE >= 2 ? S : 1], aa[D][PRE >= 3 ? S : 1];
        unsigned pm[D][S];
        int ccs[D];
        const u32x4 rs_u = make_rsrc(PRE >= 2 ? pre.aux : nullptr,
                                     PRE >= 2 ? (unsigned)(((size_t)(pre.rows - 1) * pre.aux_ld + kc) * ESZ) : 0u);
        const u32x4 rs_a = make_rsrc(PRE >= 3 ? pre.add : nullptr,
                                     PRE >= 3 ? (unsigned)(((size_t)(pre.rows - 1) * pre.add_ld + kc) * ESZ) : 0u);
        const unsigned aux_rb = PRE >= 2 ? pre.aux_ld * ESZ : 0u, add_rb = PRE >= 3 ? pre.add_ld * ESZ : 0u;
        // (o, cc, have): the next unit to issue.  Every asm load below is UNCONDITIONAL (a slot with
        // nothing left to fetch gets the all-out-of-range dummy unit, strip MAX_K): conditional
        // definitions of ring registers make the register allocator resolve phis with moves of
        // registers whose loads are still in flight (observed as a nondeterministic race).
        // PRE_CC_OUTER: the chunk as the OUTER loop (all active offsets of chunk 0, then of chunk 1, ...): a lane's channels — and with
        // them its registers of per-channel vectors — change n_chunk times per wave instead of once per unit
#ifdef DODA_PRE_CC_OUTER
        constexpr bool PRE_CC_OUTER = true;
#else
        constexpr bool PRE_CC_OUTER = false;
#endif
        const unsigned active_all = active;
        PreCo<PESZ, PKIND> pre_cv;
        int pre_cc = -1;
        int o = __builtin_ctz(active), cc = 0, o2 = MAX_K;
        active &= active - 1;
        if (PAIR && active != 0) { o2 = __builtin_ctz(active); active &= active - 1; }
        bool have = true;
        int to_issue = n_units;

        auto issue = [&](raw (&xr)[S], raw (&wr)[NBW], raw (&ur)[PRE >= 2 ? S : 1], raw (&ar)[PRE >= 3 ? S : 1], unsigned (&pmr)[S],
                         int &ccr) {
            if constexpr (PAIR) {
                // offset of this lane's half of the unit; strip MAX_K / a W offset past the packed
                // buffer are the all-zero dummies (odd offset count, ring top-up)
                const int osel = have ? (second ? o2 : o) : MAX_K;
                const unsigned *p = my_off + osel * RW;
                unsigned off[S];   // all LDS reads first (one ds_read2 / one wait), then the gathers
#pragma unroll
                for (int s = 0; s < S; ++s) off[s] = p[s * 16];
#pragma unroll
                for (int s = 0; s < S; ++s) P::gather(xr[s], off[s] + lane_x, rs_x, 0u);
                const unsigned voff_w = __umul24((unsigned)osel, (unsigned)NB * 32u * FSZ) + lane_w;
                const unsigned soff_w = (unsigned)nb0 * 32u * FSZ;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) P::load(wr[nb], voff_w, rs_w, soff_w + nb * 32u * FSZ);
                if (have) {
                    --to_issue;
                    if (active == 0) have = false;
                    else {
                        o = __builtin_ctz(active); active &= active - 1;
                        o2 = MAX_K;
                        if (active != 0) { o2 = __builtin_ctz(active); active &= active - 1; }
                    }
                }
                return;
            }
            // a lane whose channels lie past kc (partial last chunk) fetches the all-absent strip
            const bool lane_ok = have && (cc * CH + lane_c < kc);
            const unsigned *p = my_off + (lane_ok ? o : MAX_K) * RW;
            const unsigned soff_x = (unsigned)cc * (unsigned)CH * ESZ;
            unsigned off[S];
#pragma unroll
            for (int s = 0; s < S; ++s) off[s] = p[s * 16];
            if constexpr (PRE != 0) {
                ccr = cc;
#pragma unroll
                for (int s = 0; s < S; ++s) pmr[s] = (int)off[s] >= 0 ? ~0u : 0u;
            }
            if constexpr (PRE >= 2) {   // the strip holds row indices: one multiply per operand
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const bool pr = (int)off[s] >= 0;
                    P::gather(xr[s], pr ? off[s] * row_bytes + lane_x : OOB, rs_x, soff_x);
                    P::gather(ur[s], pr ? off[s] * aux_rb + lane_x : OOB, rs_u, soff_x);
                    if constexpr (PRE >= 3) P::gather(ar[s], pr ? off[s] * add_rb + lane_x : OOB, rs_a, soff_x);
                }
            } else {
#pragma unroll
                for (int s = 0; s < S; ++s) P::gather(xr[s], off[s] + lane_x, rs_x, soff_x);
            }
            unsigned soff_w = (unsigned)((o * n_chunk + cc) * NB + nb0) * 64u * FSZ;
            // (PRE >= 2: hipcc computes this uniform value on the vector side — the asm's "s" operand then fails to assemble)
            if constexpr (PRE >= 2) soff_w = (unsigned)__builtin_amdgcn_readfirstlane((int)soff_w);
            const unsigned voff_w = have ? lane_w : OOB;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) P::load(wr[nb], voff_w, rs_w, soff_w + nb * 64u * FSZ);
            if (have) {
                --to_issue;
                if constexpr (PRE != 0 && PRE_CC_OUTER) {
                    if (active == 0) {
                        if (++cc == n_chunk) have = false;
                        else { active = active_all; o = __builtin_ctz(active); active &= active - 1; }
                    } else { o = __builtin_ctz(active); active &= active - 1; }
                } else if (++cc == n_chunk) {
                    cc = 0;
                    if (active == 0) have = false;
                    else { o = __builtin_ctz(active); active &= active - 1; }
                }
            }
        };
        auto consume = [&](raw (&xr)[S], raw (&wr)[NBW], raw (&ur)[PRE >= 2 ? S : 1], raw (&ar)[PRE >= 3 ? S : 1], unsigned (&pmr)[S],
                           int ccr) {
#pragma unroll
            for (int s = 1; s < S; ++s) touch(xr[s]);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) touch(wr[nb]);
            if constexpr (PRE != 0) {
                if constexpr (PRE >= 2) {
#pragma unroll
                    for (int s = 0; s < S; ++s) touch(ur[s]);
                }
                if constexpr (PRE >= 3) {
#pragma unroll
                    for (int s = 0; s < S; ++s) touch(ar[s]);
                }
#ifdef DODA_PRE_ABLATE
                if (!(ablate & 2))
#endif
                {
                if (ccr != pre_cc) {                // (wave-uniform: a new chunk = this lane's next 16 / ESZ channels)
                    pre_cc = ccr;
                    pre_load_co<PESZ, PKIND>(pre_co, ccr * CH + lane_c, pre_cv);
                }
#pragma unroll
                for (int s = 0; s < S; ++s)
                    xr[s] = pre_piece<PESZ, PKIND>(xr[s], ur[PRE >= 2 ? s : 0], ar[PRE >= 3 ? s : 0], pre_cv, pre.relu, pmr[s]);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int s = 0; s < S; ++s)
                    P::mma(acc[s][nb], wr[nb], xr[s]);
        };

#pragma unroll
        for (int k = 0; k < D; ++k) issue(xa[k], wb[k], ua[k], aa[k], pm[k], ccs[k]);
        int consumed = 0;
        // steady state: D-1 units (L loads each) stay in flight behind the one being consumed
        while (to_issue > 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                wait_vm<(D - 1) * L>(xa[k][0]);
                consume(xa[k], wb[k], ua[k], aa[k], pm[k], ccs[k]);
                issue(xa[k], wb[k], ua[k], aa[k], pm[k], ccs[k]);
            }
            consumed += D;
        }
        // tail: everything is issued; drain once, then use the slots that still hold real units
        wait_vm<0>(xa[0][0]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k > 0) touch(xa[k][0]);
            if (consumed + k < n_units) consume(xa[k], wb[k], ua[k], aa[k], pm[k], ccs[k]);
            else {
#pragma unroll
                for (int s = 1; s < S; ++s) touch(xa[k][s]);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) touch(wb[k][nb]);
                if constexpr (PRE >= 2) {
#pragma unroll
                    for (int s = 0; s < S; ++s) touch(ua[k][s]);
                }
                if constexpr (PRE >= 3) {
#pragma unroll
                    for (int s = 0; s < S; ++s) touch(aa[k][s]);
                }
            }
        }
    }

    if constexpr (SPLIT) {
        __shared__ f32x4 part[3][S][NBW][64];
        if (wid > 0) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) part[wid - 1][s][nb][lane] = acc[s][nb];
        }
        doda_sync();
        if (wid > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const f32x4 p = part[w][s][nb][lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[s][nb][q] += p[q];
                }
    }
    // lane (row = lane&15, g) holds channels 4g..4g+3 of each 16-channel block: one wide store.
    // Channel block outermost: the statistics of one block (8 registers) are reduced and parked in LDS
    // before the next block starts, and the scheduling barrier keeps hipcc from interleaving the blocks
    // (with all blocks in flight the epilogue needed more registers than the gather loop and cost
    // occupancy: 46 -> 92 VGPRs on the 48-channel tile).
    __shared__ f32x4 sred[STATS ? (SPLIT ? 1 : 4) : 1][STATS ? NBW : 1][2][4];
    const int part = item / n_nbg;   // the workgroup's row tile
    // row strides (ABI 11; 0 = dense): y, the residual and the BatchNorm input of the data-gradient statistics may be column
    // slices of wider matrices
    const unsigned y_ld = ep.y_ld ? ep.y_ld : (unsigned)nc, res_ld = ep.res_ld ? ep.res_ld : (unsigned)nc,
                   bnx_ld = ep.bnx_ld ? ep.bnx_ld : (unsigned)nc;
    const unsigned res_bytes = ep.res_ld ? (unsigned)(((size_t)(n_out - 1) * res_ld + nc) * OSZ) : (ep.y_ld ? (unsigned)((size_t)n_out * nc * OSZ) : y_bytes);
    const unsigned bnx_bytes = ep.bnx_ld ? (unsigned)(((size_t)(n_out - 1) * bnx_ld + nc) * OSZ) : (ep.y_ld ? (unsigned)((size_t)n_out * nc * OSZ) : y_bytes);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const unsigned col = (unsigned)((nb0 + nb) * 16 + 4 * g);
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const unsigned t = (unsigned)(row0 + s * 16 + i);
            const bool in_range = t < (unsigned)n_out && col < (unsigned)nc;
            const unsigned voff = in_range ? (t * y_ld + col) * OSZ : OOB;
            const unsigned voff_res = in_range ? (t * res_ld + col) * OSZ : OOB;
            const unsigned voff_bnx = in_range ? (t * bnx_ld + col) * OSZ : OOB;
            if (res) {   // y = conv + res (residual add of the block fused into the store; res has y's dtype)
                const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)res, 0, res_bytes, 0x00020000);
                // (res_bcast: one row for every output row — the Linear head's bias, reference model/unet.py:64)
                const unsigned voff_r = ep.res_bcast ? (in_range ? col * OSZ : OOB) : voff_res;
                if (OUT32 || sizeof(elem) == 4) {
                    const f32x4 r4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, voff_r, 0, 0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[s][nb][q] += r4[q];
                } else {
                    const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rs_r, voff_r, 0, 0);
                    acc[s][nb][0] += __uint_as_float(r2[0] << 16);
                    acc[s][nb][1] += __uint_as_float(r2[0] & 0xffff0000u);
                    acc[s][nb][2] += __uint_as_float(r2[1] << 16);
                    acc[s][nb][3] += __uint_as_float(r2[1] & 0xffff0000u);
                }
            }
            u32x2 packed_out = {0u, 0u};   // bf16 output: rounded ONCE, for the store and for the statistics
            if constexpr (!(OUT32 || sizeof(elem) == 4)) {
                packed_out[0] = (unsigned)f2bf(acc[s][nb][0]) | ((unsigned)f2bf(acc[s][nb][1]) << 16);
                packed_out[1] = (unsigned)f2bf(acc[s][nb][2]) | ((unsigned)f2bf(acc[s][nb][3]) << 16);
            }
            if constexpr (STATS) {
                // rows past n_out / channels past nc carry zeros (their loads were out of range)
                f32x4 v = acc[s][nb];
                if constexpr (!(OUT32 || sizeof(elem) == 4)) {
                    v = (f32x4){__uint_as_float(packed_out[0] << 16), __uint_as_float(packed_out[0] & 0xffff0000u),
                                __uint_as_float(packed_out[1] << 16), __uint_as_float(packed_out[1] & 0xffff0000u)};
                }
                if (ep.bn_x) {
                    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)ep.bn_x, 0, bnx_bytes, 0x00020000);
                    f32x4 xr;
                    if (OUT32 || sizeof(elem) == 4) {
                        xr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_bnx, 0, 0));
                    } else {
                        const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rs_x, voff_bnx, 0, 0);
                        xr = (f32x4){__uint_as_float(r2[0] << 16), __uint_as_float(r2[0] & 0xffff0000u),
                                     __uint_as_float(r2[1] << 16), __uint_as_float(r2[1] & 0xffff0000u)};
                    }
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
            if constexpr (OUT32 || sizeof(elem) == 4)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[s][nb]), rs_y, voff, 0, 0);
            else
                __builtin_amdgcn_raw_buffer_store_b64(packed_out, rs_y, voff, 0, 0);
        }
        if constexpr (STATS) {
            // sum over the 16 rows of the lane group: lane 15 of each group holds the total
#pragma unroll
            for (int q = 0; q < 4; ++q) { st1[q] = row_sum16(st1[q]); st2[q] = row_sum16(st2[q]); }
            if (i == 15) { sred[SPLIT ? 0 : wid][nb][0][g] = st1; sred[SPLIT ? 0 : wid][nb][1][g] = st2; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (STATS) {
        if constexpr (!SPLIT) doda_sync();   // the four waves hold four row ranges of the tile
        if (wid == 0 && i == 15) {
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int col = (nb0 + nb) * 16 + 4 * g;
                if (col < nc) {   // (nc % 4 == 0 on this path)
                    f32x4 a1 = sred[0][nb][0][g], a2 = sred[0][nb][1][g];
                    if constexpr (!SPLIT) {
                        a1 = (a1 + sred[1][nb][0][g]) + (sred[2][nb][0][g] + sred[3][nb][0][g]);
                        a2 = (a2 + sred[1][nb][1][g]) + (sred[2][nb][1][g] + sred[3][nb][1][g]);
                    }
                    stats_emit(ep, (long long)part, nc, col, a1, a2);
                }
            }
        }
    }
}

// kernel shapes that carry the folded BatchNorm (PreArgs): 16-byte pieces of same-dtype rows, split blocks — the shapes of the
// coarse U-Net levels, where a BatchNorm sweep of its own is a launch-floor kernel
template <class P, int NBW, int S, bool SPLIT>
constexpr bool pre_shape() {
    return SPLIT && ((NBW == 1 && S == 1) || (NBW == 4 && S == 2)) && (std::is_same<P, PBF16W>::value || std::is_same<P, PF32>::value);
}

template <class P, int NBW, int S, bool SPLIT = false>
int launch_fast(const typename P::elem *x, int kc, const void *wp, size_t wp_bytes, int nc, int NB,
                const int32_t *tbl, int ld, int K, int n_out, long long n_in, void *y, bool out32,
                const void *res, const EpiArgs &ep_in, int *n_part, hipStream_t s, const PreArgs *pre = nullptr) {
    if constexpr (std::is_same<P, PF32>::value) {   // fp32 layers of many rows: the bf16 head / tail instantiation
        if (ep_in.f32_split && !(pre && pre->kind))
            return launch_fast<PF32S, NBW, S, SPLIT>(x, kc, wp, wp_bytes, nc, NB, tbl, ld, K, n_out, n_in, y, out32, res, ep_in, n_part, s);
    }
    const dim3 grid(div_up(n_out, (SPLIT ? 1 : 4) * 16 * S) * div_up(NB, NBW)), block(256);
    if (n_part) *n_part = div_up(n_out, (SPLIT ? 1 : 4) * 16 * S);
    const EpiArgs &ep = ep_in;
    // Ring depth.  Re-measured after the EXEC-masked gathers and the wide / pair units went in: with
    // every load hitting L1 (ablation) the kernel time did not move, i.e. the unit loop is paced by
    // instruction issue and by how many waves a SIMD can interleave, not by memory latency.  Depth 8
    // cost 124 VGPRs (4 waves per SIMD); depth 3: level-1 16->16 52 -> 38 us, level-2 32->32 37 -> 32 us.
    // (round 6: depth 6 for the 16-row split blocks of the coarse levels — ~20 units per wave, one wave per SIMD — measured the same
    // 8.1-8.2 us per launch: those kernels are not paced by the ring either)
    constexpr int D = 3;
    const size_t esz = sizeof(typename P::elem);
    const unsigned xb = (unsigned)(ep.x_ld ? ((size_t)(n_in - 1) * ep.x_ld + kc) * esz : (size_t)n_in * kc * esz);
    const unsigned tb = (unsigned)((size_t)K * ld * 4);
    const PreArgs none{};
    if (pre && pre->kind) {
        if constexpr (pre_shape<P, NBW, S, SPLIT>()) {
            if (out32 && esz != 4) return DODA_ERR_UNSUPPORTED;
            const unsigned yb = (unsigned)(ep.y_ld ? ((size_t)(n_out - 1) * ep.y_ld + nc) * esz : (size_t)n_out * nc * esz);
#define DODA_PRE_GO(KIND)                                                                                            \
            do {                                                                                                     \
                if (ep.stats)                                                                                        \
                    hipLaunchKernelGGL((conv_fast<P, NBW, S, D, false, SPLIT, true, KIND>), grid, block, 0, s, x, xb, kc, wp,  \
                                       (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, *pre);     \
                else                                                                                                 \
                    hipLaunchKernelGGL((conv_fast<P, NBW, S, D, false, SPLIT, false, KIND>), grid, block, 0, s, x, xb, kc, wp, \
                                       (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, *pre);     \
            } while (0)
            if ((pre->kind & 0xff) == 1) DODA_PRE_GO(1);
            else if ((pre->kind & 0xff) == 2) DODA_PRE_GO(2);
            else if ((pre->kind & 0xff) == 3) DODA_PRE_GO(3);
            else return DODA_ERR_INVALID;
#undef DODA_PRE_GO
            return doda_check_launch();
        } else {
            return DODA_ERR_UNSUPPORTED;
        }
    }
    if (out32 && sizeof(typename P::elem) != 4) {
        const unsigned yb = (unsigned)(ep.y_ld ? ((size_t)(n_out - 1) * ep.y_ld + nc) * 4 : (size_t)n_out * nc * 4);
        if (ep.stats)
            hipLaunchKernelGGL((conv_fast<P, NBW, S, D, true, SPLIT, true>), grid, block, 0, s, x, xb, kc, wp,
                               (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, none);
        else
            hipLaunchKernelGGL((conv_fast<P, NBW, S, D, true, SPLIT, false>), grid, block, 0, s, x, xb, kc, wp,
                               (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, none);
    } else {
        const unsigned yb = (unsigned)(ep.y_ld ? ((size_t)(n_out - 1) * ep.y_ld + nc) * esz : (size_t)n_out * nc * esz);
        if (ep.stats)
            hipLaunchKernelGGL((conv_fast<P, NBW, S, D, false, SPLIT, true>), grid, block, 0, s, x, xb, kc, wp,
                               (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, none);
        else
            hipLaunchKernelGGL((conv_fast<P, NBW, S, D, false, SPLIT, false>), grid, block, 0, s, x, xb, kc, wp,
                               (unsigned)wp_bytes, nc, NB, tbl, tb, ld, K, n_out, y, yb, res, ep, none);
    }
    return doda_check_launch();
}

template <class T, int NBW, int S>
int launch(const typename T::elem *x, int kc, const typename T::frag *wp, int nc, int NB,
           const int32_t *tbl, int ld, int K, int n_out, typename T::elem *y, int vec_ok,
           const typename T::elem *res, hipStream_t s) {
    const dim3 grid(div_up(n_out, 4 * 16 * S) * div_up(NB, NBW)), block(256);
    hipLaunchKernelGGL((conv_gather<T, NBW, S>), grid, block, 0, s, x, kc, wp, nc, NB, tbl, ld, K,
                       n_out, y, vec_ok, res);
    return doda_check_launch();
}

template <class T> struct FastPolicy;
template <> struct FastPolicy<F32> { typedef PF32 narrow; typedef PF32 wide; typedef PF32 pair; };
template <> struct FastPolicy<BF16> { typedef PBF16 narrow; typedef PBF16W wide; typedef PBF16P pair; };

// Fragment packing the fast kernel uses for a layer: 0x10 wide (bf16, >= 32 input channels),
// 0x20 pair (bf16, exactly 16 input channels, more than one offset), 0 narrow.
inline int pack_mode(int K, int kc, int elem_bytes) {
    if (elem_bytes == 2 && kc >= 32 && kc % 8 == 0) return 0x10;
    if (elem_bytes == 2 && kc == 16 && K >= 2) return 0x20;
    return 0;
}

template <class T>
int run_gather(const void *x_, int kc, const float *w, int nc, const int32_t *tbl, int ld, int K,
               int n_out, void *y_, int wl, void *ws, size_t ws_bytes, long long n_in, bool out32,
               const void *res, hipStream_t s,
               const EpiArgs &ep_arg = EpiArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr},
               int *n_part = nullptr, const void *tilebook = nullptr, int tilebook_rows = 0, const PreArgs *pre = nullptr) {
    typedef typename T::elem elem;
    typedef typename T::frag frag;
    const elem *x = (const elem *)x_;
    elem *y = (elem *)y_;
    const int NB = (nc + 15) / 16;
    EpiArgs ep = ep_arg;
    {   // OPT-IN: fp32 layers of at least DODA_F32_SPLIT_ROWS output rows (e.g. 65536; 0: all; unset / -1: none) multiply bf16
        // head / tail splits of both operands (spconv_common.hpp mma_f32_k16).  Measured (round 5): fp32 step 11.85 -> 11.1 ms
        // with every weight gradient and the >= 65536-row gathers split, every 1e-4 kernel test and the golden's gradient
        // NORMS (5e-3) still green — but the elementwise distance of the U-Net's gradients from the fp64 golden grows from
        // ~1e-3 to ~7e-3 (2^-16 products through 70 layers), and fp32 is this repository's PARITY precision: exact by default
        static const long long min_rows = [] { const char *e = getenv("DODA_F32_SPLIT_ROWS"); return e && *e ? atoll(e) : -1ll; }();
        ep.f32_split = (sizeof(elem) == 4 && min_rows >= 0 && (long long)n_out >= min_rows) ? 1 : 0;
    }
    // all rows the table may reference must sit inside the 2 GB buffer window of the fast path
    const bool x_rows_bytes_ok = n_in > 0 && (size_t)n_in * kc * sizeof(elem) < 0x7ffffff0ull;
    const size_t va = 4 * sizeof(elem);  // vector access granule
    const int vec_ok = (kc % 4 == 0) && (nc % 4 == 0) && ((uintptr_t)x % va == 0) &&
                       ((uintptr_t)y % va == 0);
    // ABI 11: row strides (column slices of wider matrices) and the folded BatchNorm in conv_fast; ABI 12: the LDS-staged kernels
    // (conv_tile*, conv_up32, conv_wlds48) take strided OUTPUT-side operands (y, residual, BatchNorm input) — their gathered x stays dense
    const bool strided = (ep.x_ld && ep.x_ld != (unsigned)kc) || ep.y_ld || ep.res_ld || ep.bnx_ld;
    const bool x_dense = !ep.x_ld || ep.x_ld == (unsigned)kc;
    const bool folded = pre && pre->kind != 0;
    const size_t x_ld = ep.x_ld ? ep.x_ld : (size_t)kc, y_ld = ep.y_ld ? ep.y_ld : (size_t)nc;
    const bool ld_ok = x_ld % 4 == 0 && y_ld % 4 == 0 && ep.res_ld % 4 == 0 && ep.bnx_ld % 4 == 0 &&
                       (size_t)n_in * x_ld * sizeof(elem) < 0x7ffffff0ull && (size_t)n_out * y_ld * 4 < 0x7fffffffull &&
                       (size_t)n_out * (ep.res_ld ? ep.res_ld : (size_t)nc) * 4 < 0x7fffffffull &&
                       (size_t)n_out * (ep.bnx_ld ? ep.bnx_ld : (size_t)nc) * 4 < 0x7fffffffull;
    const bool fast = (kc % 4 == 0) && (nc % 4 == 0) && ((uintptr_t)x % 16 == 0) &&
                      ((uintptr_t)y % 16 == 0) && ((size_t)n_out * nc * 4 < 0x7fffffffull) &&
                      ((size_t)K * ld * 4 < 0xffffffffull) && x_rows_bytes_ok && ld_ok;
    if ((strided || folded) && !fast) return DODA_ERR_UNSUPPORTED;
    if (out32 && sizeof(elem) != 4 && !fast) return DODA_ERR_UNSUPPORTED;
    if (ep.stats && !fast) return DODA_ERR_UNSUPPORTED;   // the statistics ride in the fast kernel's epilogue only
    const int mode = pack_mode(K, kc, (int)sizeof(elem));
    const bool wide = fast && mode == 0x10, pair = fast && mode == 0x20;
    {   // DODA_TRACE_GATHER=1: one line per call on stderr (which layer shapes reach which kernel: tools/gathermap.py)
        static const bool trace = getenv("DODA_TRACE_GATHER") && getenv("DODA_TRACE_GATHER")[0] == '1';
        if (trace)
            fprintf(stderr, "doda_gather K=%d kc=%d nc=%d n_out=%d n_in=%lld layout=%d esz=%d out32=%d stats=%d bn=%d res=%d tilebook=%d\n",
                    K, kc, nc, n_out, n_in, wl & 0xff, (int)sizeof(elem), (int)out32, ep.stats ? 1 : 0, ep.bn_x ? 1 : 0,
                    res ? 1 : 0, tilebook ? 1 : 0);
    }
    const int n_chunk = wide ? (kc + 31) / 32 : (kc + 15) / 16;
    const size_t need = pair ? (size_t)K * NB * 32 * 16 : (size_t)K * n_chunk * NB * 64 * (wide ? 16 : sizeof(frag));
    const void *wp;
    if (wl & 0x100) {  // `w` already holds fragment-packed weights (doda_spconv_pack_multi)
        if (mode != 0 && !fast) return DODA_ERR_UNSUPPORTED;  // packed for a mode this call cannot take
        wp = (const void *)w;
    } else {
        if (!ws || ws_bytes < need) return DODA_ERR_WORKSPACE;
        wp = ws;
        const long long total = pair ? (long long)K * NB * 32 : (long long)K * n_chunk * NB * 64;
        if (wide || pair)
            hipLaunchKernelGGL(pack_weights_wide, dim3(div_up(total, 256)), dim3(256), 0, s, w, K, kc, nc,
                               n_chunk, NB, wl & 3, (u32x4_t *)ws, pair ? 1 : 0);
        else
            hipLaunchKernelGGL((pack_weights<T>), dim3(div_up(total, 256)), dim3(256), 0, s, w, K, kc,
                               nc, n_chunk, NB, wl & 3, (frag *)ws);
    }
    if (ep.res_bcast && !fast) return DODA_ERR_UNSUPPORTED;   // (the broadcast residual lives in conv_fast's epilogue)
    if (folded) {
        // the folded BatchNorm: 16-byte pieces of rows in the output's dtype (bf16 >= 32 channels, fp32), at most PRE_MAX_C
        // channels, every operand 16-byte aligned; split blocks — 16 rows x one channel block while the grid stays small,
        // 32 rows x four channel blocks above (the shapes the unfolded call would take at the coarse levels)
        const size_t va16 = 16 / sizeof(elem);
        const int pk = pre->kind & 0xff;     // (the high bits carry the DODA_PRE_ABLATE mask of debug builds)
        if (out32 || pair || (sizeof(elem) == 2 && !wide) || kc > PRE_MAX_C || kc % (int)va16 != 0 || x_ld % va16 != 0 ||
            !pre->side || pre->side_ld % va16 != 0 || ((uintptr_t)pre->side % 16) != 0 || pre->rows != (int)n_in ||
            (pk >= 2 && (!pre->aux || pre->aux_ld % va16 != 0 || ((uintptr_t)pre->aux % 16) != 0 || !pre->mean || !pre->invstd ||
                                !pre->tot.ta || (size_t)n_in * pre->aux_ld * sizeof(elem) >= 0x7ffffff0ull)) ||
            (pk >= 3 && (!pre->add || pre->add_ld % va16 != 0 || ((uintptr_t)pre->add % 16) != 0 ||
                                (size_t)n_in * pre->add_ld * sizeof(elem) >= 0x7ffffff0ull)) ||
            (pk == 1 && !pre->tot.ta && (!pre->tot.rm || !pre->tot.rv)) || !pre->gamma || !pre->beta)
            return DODA_ERR_UNSUPPORTED;
        typedef typename FastPolicy<T>::wide PWp;
        typedef typename FastPolicy<T>::narrow PNp;
        const long long wf = ((long long)n_out + 15) / 16;
        static const long long small_max = [] { const char *e = getenv("DODA_PRE_SMALL_BLOCKS"); return e && *e ? atoll(e) : 2048ll; }();
        if (wf * NB <= small_max) {
            if (wide) return launch_fast<PWp, 1, 1, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s, pre);
            return launch_fast<PNp, 1, 1, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s, pre);
        }
        if (wide) return launch_fast<PWp, 4, 2, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s, pre);
        return launch_fast<PNp, 4, 2, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s, pre);
    }
    // K <= 8, 32 input channels, fewer input rows than output rows (the k2 s2 rulebook read from the fine side: one source row
    // per output row): conv_up32 (spconv_tile.hip).  DODA_CONV_UP=0 / doda_set_option(DODA_OPT_CONV_UP, 0): conv_fast as before.
    {
        if (doda_tile::up_enabled() && wide && sizeof(elem) == 2 && kc == 32 && K <= 8 && K > 1 && n_in < (long long)n_out && nc % 16 == 0 &&
            !ep.res_bcast && x_dense && doda_tile::enabled()) {
            const unsigned xb = (unsigned)((size_t)n_in * kc * sizeof(elem));
            const unsigned yb = (unsigned)((((size_t)n_out - 1) * y_ld + nc) * (out32 ? 4 : sizeof(elem)));
            return doda_tile::launch_conv_up32(out32, x_, xb, wp, (unsigned)need, nc, NB, K, tbl, ld, n_out, y_, yb, res, ep, n_part, s);
        }
    }
    // A tilebook of this table and rows of 32 / 64 bytes: the LDS-staged tile kernel (spconv_tile.hip)
    if (!ep.res_bcast && x_dense) {
        // (fp32 rows: the tile kernel's fp32 mode is bound by the fp32 matrix rate like the dense-table kernel and measured
        // within a few percent of it; DODA_F32_CONV_TILE=0 keeps fp32 forward / data-grad calls on conv_fast even when the
        // table carries a tilebook — the fp32 weight gradient uses the tilebook either way)
        static const bool f32_tile = !(getenv("DODA_F32_CONV_TILE") && getenv("DODA_F32_CONV_TILE")[0] == '0');
        const int tmode = pair ? 0 : (wide && kc == 32) ? 1 : (fast && sizeof(elem) == 4 && kc == 16 && f32_tile) ? 2 : -1;
        // (statistics: the tile kernels' per-lane accumulators hold up to two channel blocks, the dual-pass 64-byte-row kernel four)
        const bool stats_fit = !ep.stats || NB <= 2 || (tmode == 1 && NB == 4 && doda_tile::dual_enabled());
        if (tmode >= 0 && tilebook && K == TB_K && tilebook_rows == n_out && doda_tile::enabled() && stats_fit) {
            const unsigned xb = (unsigned)((size_t)n_in * kc * sizeof(elem));
            const unsigned yb = (unsigned)((((size_t)n_out - 1) * y_ld + nc) * (out32 ? 4 : sizeof(elem)));
            return doda_tile::launch_conv_tile(tmode, out32 || sizeof(elem) == 4, x_, xb, wp, (unsigned)need, nc, NB, tbl, ld,
                                               n_out, tilebook, y_, yb, res, ep, n_part, s);
        }
    }
    // 48 -> 48 channels on a mid-size level: the layer's fragments in LDS, one workgroup per CU (spconv_wlds.hip)
    if (!ep.res_bcast && x_dense && wide && kc == 48 && nc == 48 && K == 27 && !out32 && n_out >= 8192 && n_out <= 262144 && doda_wlds::enabled()) {
        const unsigned xb = (unsigned)((size_t)n_in * kc * sizeof(elem));
        const unsigned yb = (unsigned)((((size_t)n_out - 1) * y_ld + nc) * sizeof(elem));
        return doda_wlds::launch_conv48(x_, xb, wp, tbl, (unsigned)((size_t)K * ld * 4), ld, n_out, y_, yb, res, ep, n_part, s);
    }
    // Tile choice: many rows -> more subtiles per wave and all channel blocks in one wave (x is
    // gathered once); few rows -> one subtile, channel blocks spread over the grid so the chip
    // still sees thousands of waves.
    const long long waves_full = ((long long)n_out + 15) / 16;
    typedef typename FastPolicy<T>::narrow PN;
    typedef typename FastPolicy<T>::wide PW;
    typedef typename FastPolicy<T>::pair PP;
#define GO(NBW, S)                                                                                 \
    do {                                                                                           \
        if (wide) return launch_fast<PW, NBW, S>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
        if (pair) return launch_fast<PP, NBW, S>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
        if (fast) return launch_fast<PN, NBW, S>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
        return launch<T, NBW, S>(x, kc, (const frag *)wp, nc, NB, tbl, ld, K, n_out, y, vec_ok, (const elem *)res, s); \
    } while (0)
    {   // few rows, long unit chains: split the offsets of a 16-row tile over the block's waves
        // measured (rocprofv3, per dispatch): 795 / 210 / 49 blocks 12.7 -> 9.5, 12.2 -> 6.1,
        // 16.0 -> 6.3 us; 2808 blocks (level 4) 18.0 -> 24.7 us, so only below ~1k blocks
#define GS(NBW, S)                                                                                 \
    do {                                                                                           \
        if (wide) return launch_fast<PW, NBW, S, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
        if (pair) return launch_fast<PP, NBW, S, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
        return launch_fast<PN, NBW, S, true>(x, kc, wp, need, nc, NB, tbl, ld, K, n_out, n_in, y_, out32, res, ep, n_part, s); \
    } while (0)
        // (two channel blocks / 32-row tiles per split block were tried at level 4: 14.2 us against
        // 13.0 us for the unsplit <4,1> tile, so the split stays at one block, 16 rows)
        if (fast && (long long)K * n_chunk >= 12 && waves_full * NB <= 1024) GS(1, 1);
        // mid levels, 3-4 channel blocks: 32-row split blocks load each weight fragment once per 32
        // rows instead of once per 16 (level 3, 46k rows x 48 ch: 22.0 -> 19.6 us; level 4, 11k x 64:
        // 13.9 -> 11.9 us); 64-row split blocks and 2-block layers lose (23.2 / 36.3 us)
        if (fast && (long long)K * n_chunk >= 12) {
            if (NB == 3 && waves_full >= 512 && waves_full < 8192) GS(3, 2);
            if (NB == 4 && waves_full >= 512 && waves_full < 2048) GS(4, 2);
        }
#undef GS
    }
    if (NB == 1) {  // measured at M = 600k, 16 ch: S=2 51 us, S=4 56 us, S=1 56 us
        if (waves_full >= 4096) GO(1, 2);
        GO(1, 1);
    }
    if (NB == 2) {
        // level 2 (183k rows, 32 ch): bf16 <2,4> 27.7 us, <2,2> 30.2, <2,1> 35.0; fp32 98.8 / 93.6 / 92.8
        if (waves_full >= 8192 && sizeof(elem) == 2) GO(2, 4);
        if (waves_full >= 8192) GO(2, 2);
        if (waves_full >= 2048) GO(2, 1);
        GO(1, 1);
    }
    if (NB == 3) {   // 48 channels: three channel blocks exactly (a <4,*> tile would load and multiply a zero block)
        if (waves_full >= 512) GO(3, 1);   // level 3 (46k rows): <3,1> 23.2 us, <4,1> 27.8, <3,2> 22.8 (fp32 74.6)
        GO(1, 1);
    }
    if (NB <= 4) {
        if (waves_full >= 8192) GO(4, 2);
        if (waves_full >= 512) GO(4, 1);   // level 4 (11k rows, 64 ch): <4,1> 13.0 us, <2,1> 17.3, <2,2> 14.9
        GO(1, 1);
    }
    if (waves_full >= 4096) GO(8, 1);
    if (waves_full >= 1024) GO(4, 1);
    if (waves_full >= 256) GO(2, 1);
    GO(1, 1);
#undef GO
}

bool bad_args(const void *x, int kc, const float *w, int nc, const int32_t *tbl, int ld, int K,
              int n_out, const void *y, int wl, int *status) {
    if (kc <= 0 || nc <= 0 || K <= 0 || n_out < 0 || ld < n_out || (wl & 3) > 2 || (wl & ~0x103)) {
        *status = DODA_ERR_INVALID;
        return true;
    }
    if (n_out == 0) { *status = DODA_OK; return true; }
    if (!x || !w || !tbl || !y) { *status = DODA_ERR_INVALID; return true; }
    if (K > MAX_K || nc > 4096 || kc > 4096) { *status = DODA_ERR_UNSUPPORTED; return true; }
    return false;
}
}  // namespace

extern "C" size_t doda_spconv_gather_workspace_bytes(int32_t K, int32_t kc, int32_t nc,
                                                     int32_t elem_bytes) {
    if (K <= 0 || kc <= 0 || nc <= 0) return 0;
    if (elem_bytes == 2)  // covers both the 16-channel (8 B) and the 32-channel (16 B) fragment packing
        return align_up((size_t)K * ((kc + 31) / 32) * ((nc + 15) / 16) * 64 * 16, 256);
    return align_up((size_t)K * ((kc + 15) / 16) * ((nc + 15) / 16) * 64 * 4 * (size_t)elem_bytes, 256);
}

extern "C" size_t doda_spconv_pack_desc_bytes(void) { return sizeof(PackDesc); }

// descs_h: n_desc host descriptors {w, out, K, kc, nc, layout, elem_bytes}; fills the derived
// fields and the block prefix, both written to the caller's host arrays for upload.
extern "C" int doda_spconv_pack_plan_h(void *descs_h, int32_t n_desc, int32_t *blk_end_h,
                                       int32_t *total_blocks) {
    if (!descs_h || !blk_end_h || !total_blocks || n_desc <= 0) return DODA_ERR_INVALID;
    PackDesc *d = (PackDesc *)descs_h;
    long long acc = 0;
    for (int k = 0; k < n_desc; ++k) {
        if (d[k].K <= 0 || d[k].K > MAX_K || d[k].kc <= 0 || d[k].nc <= 0 || (d[k].layout & ~3) ||
            (d[k].layout & 3) > 2 || (d[k].elem_bytes != 2 && d[k].elem_bytes != 4))
            return DODA_ERR_INVALID;
        d[k].layout |= pack_mode(d[k].K, d[k].kc, d[k].elem_bytes);   // the mode the gather will expect
        d[k].n_chunk = (d[k].layout & 0x10) ? (d[k].kc + 31) / 32 : (d[k].kc + 15) / 16;
        d[k].NB = (d[k].nc + 15) / 16;
        acc += div_up((d[k].layout & 0x20) ? (long long)d[k].K * d[k].NB * 32
                                           : (long long)d[k].K * d[k].n_chunk * d[k].NB * 64, 256);
        if (acc > 0x7fffffff) return DODA_ERR_UNSUPPORTED;
        blk_end_h[k] = (int32_t)acc;
    }
    *total_blocks = (int32_t)acc;
    return DODA_OK;
}

extern "C" int doda_spconv_pack_multi(const void *descs_dev, const int32_t *blk_end_dev,
                                      int32_t n_desc, int32_t total_blocks, doda_stream_t stream) {
    if (!descs_dev || !blk_end_dev || n_desc <= 0 || total_blocks <= 0) return DODA_ERR_INVALID;
    hipLaunchKernelGGL(pack_weights_multi, dim3(total_blocks), dim3(256), 0, as_stream(stream),
                       (const PackDesc *)descs_dev, blk_end_dev, n_desc);
    return doda_check_launch();
}

// ---- gather with epilogue options (residual add, BatchNorm statistics) ------------------------------
// A/B switches of the kernel selection (measurements and parity tests; every option defaults to 1)
extern "C" int doda_set_option(int32_t option, int32_t value) {
    switch (option) {
    case DODA_OPT_TILE_KERNEL: doda_tile::set_enabled(value != 0); return DODA_OK;
    case DODA_OPT_WLDS_KERNEL: doda_wlds::set_enabled(value != 0); return DODA_OK;
    case DODA_OPT_WDMA_KERNEL: doda_wdma::set_enabled(value != 0); return DODA_OK;
    case DODA_OPT_TILE_PIPELINE: doda_tile::set_pipeline(value != 0); return DODA_OK;
    case DODA_OPT_TILE_DUAL: doda_tile::set_dual(value != 0); return DODA_OK;
    case DODA_OPT_CONV_UP: doda_tile::set_up(value != 0); return DODA_OK;
    case DODA_OPT_PRE_FWD_ROWS: doda_layers::set_fwd_rows(value); return DODA_OK;
    case DODA_OPT_PRE_BWD_ROWS: doda_layers::set_bwd_rows(value); return DODA_OK;
    default: return DODA_ERR_INVALID;
    }
}
extern "C" int32_t doda_get_option(int32_t option) {
    switch (option) {
    case DODA_OPT_TILE_KERNEL: return doda_tile::enabled() ? 1 : 0;
    case DODA_OPT_WLDS_KERNEL: return doda_wlds::enabled() ? 1 : 0;
    case DODA_OPT_WDMA_KERNEL: return doda_wdma::enabled() ? 1 : 0;
    case DODA_OPT_TILE_PIPELINE: return doda_tile::pipeline_enabled() ? 1 : 0;
    case DODA_OPT_TILE_DUAL: return doda_tile::dual_enabled() ? 1 : 0;
    case DODA_OPT_CONV_UP: return doda_tile::up_enabled() ? 1 : 0;
    case DODA_OPT_PRE_FWD_ROWS: return (int32_t)doda_layers::fwd_rows();
    case DODA_OPT_PRE_BWD_ROWS: return (int32_t)doda_layers::bwd_rows();
    default: return -1;
    }
}

extern "C" size_t doda_spconv_stats_capacity(int32_t n_out) { return n_out > 0 ? (size_t)div_up(n_out, 16) : 1; }

extern "C" int doda_spconv_gather_ex(const void *x, int32_t n_in, int32_t kc, int32_t elem_bytes, const float *w,
                                     int32_t nc, const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out, void *y,
                                     int32_t y_is_f32, int32_t w_layout, void *ws, size_t ws_bytes,
                                     const doda_conv_epilogue *epi, doda_stream_t stream) {
    int st;
    if (elem_bytes != 2 && elem_bytes != 4) return DODA_ERR_INVALID;
    if (bad_args(x, kc, w, nc, tbl, ld, K, n_out, y, w_layout, &st)) {
        if (st == DODA_OK && epi && epi->stats_rows_h) *epi->stats_rows_h = 0;
        return st;
    }
    EpiArgs ep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
    const void *res = nullptr;
    int n_part = 0;
    if (epi) {
        res = epi->residual;
        ep.res_bcast = (res && epi->residual_bcast) ? 1 : 0;
        if (epi->stats) {
            if (!epi->stats_rows_h) return DODA_ERR_INVALID;
            ep.stats = epi->stats;
            ep.stats_tot = epi->stats_totals;
            if (epi->bn_x) {
                if (!epi->bn_mean || !epi->bn_invstd || !epi->bn_gamma || !epi->bn_beta) return DODA_ERR_INVALID;
                ep.bn_x = epi->bn_x;
                ep.bn_mean = epi->bn_mean; ep.bn_invstd = epi->bn_invstd;
                ep.bn_gamma = epi->bn_gamma; ep.bn_beta = epi->bn_beta;
                ep.bn_relu = epi->bn_relu;
            }
        }
    }
    PreArgs pre{};
    const PreArgs *prep = nullptr;
    if (epi) {
        if (epi->x_ld < 0 || epi->y_ld < 0 || epi->residual_ld < 0 || epi->bn_x_ld < 0) return DODA_ERR_INVALID;
        ep.x_ld = (unsigned)epi->x_ld; ep.y_ld = (unsigned)epi->y_ld;
        ep.res_ld = res ? (unsigned)epi->residual_ld : 0u; ep.bnx_ld = ep.bn_x ? (unsigned)epi->bn_x_ld : 0u;
        if ((ep.x_ld && ep.x_ld < (unsigned)kc) || (ep.y_ld && ep.y_ld < (unsigned)nc) || (ep.res_ld && ep.res_ld < (unsigned)nc) ||
            (ep.bnx_ld && ep.bnx_ld < (unsigned)nc))
            return DODA_ERR_INVALID;
        if (ep.x_ld == (unsigned)kc) ep.x_ld = 0;      // dense
        if (ep.y_ld == (unsigned)nc) ep.y_ld = 0;
        if (ep.res_ld == (unsigned)nc) ep.res_ld = 0;
        if (ep.bnx_ld == (unsigned)nc) ep.bnx_ld = 0;
        if (ep.res_bcast) ep.res_ld = 0;
        if (const doda_conv_prologue *q = epi->prologue) {
            if (q->kind < 1 || q->kind > 3 || q->rows != n_in || q->side_ld < kc || (q->kind >= 2 && q->aux_ld < kc) ||
                (q->kind >= 3 && q->add_ld < kc) || (q->kind >= 2 && (!q->dgamma || !q->dbeta || !q->totals)) ||
                (q->kind == 1 && q->totals && (!q->mean || !q->invstd || (!q->running_mean != !q->running_var))) ||
                (q->kind == 1 && q->totals_b && (q->c_a <= 0 || q->c_a >= kc || q->c_a % 4 || !q->totals)))
                return DODA_ERR_INVALID;
            pre.kind = q->kind; pre.relu = q->relu ? 1 : 0; pre.rows = q->rows;
#ifdef DODA_PRE_ABLATE
            { static const int ab = getenv("DODA_PRE_ABLATE") ? atoi(getenv("DODA_PRE_ABLATE")) : 0; pre.kind |= ab << 8; }
#endif
            pre.tot.ta = q->totals; pre.tot.tb = q->kind == 1 ? q->totals_b : nullptr;
            pre.tot.ca = (q->kind == 1 && q->totals_b) ? q->c_a : kc;
            pre.tot.m = q->rows; pre.tot.eps = q->eps; pre.tot.momentum = q->momentum;
            pre.tot.rm = q->running_mean; pre.tot.rv = q->running_var; pre.tot.nbt = (long long *)q->num_batches_tracked;
            if (q->kind == 1) { pre.tot.out_a = q->mean; pre.tot.out_b = q->invstd; }
            else { pre.tot.out_a = q->dgamma; pre.tot.out_b = q->dbeta; pre.tot.accum = q->accumulate ? 1 : 0; pre.tot.rm = nullptr; pre.tot.rv = nullptr; pre.tot.nbt = nullptr; }
            pre.gamma = q->gamma; pre.beta = q->beta; pre.mean = q->mean; pre.invstd = q->invstd;
            pre.side = q->side; pre.side_ld = (unsigned)q->side_ld;
            pre.aux = q->aux; pre.add = q->add; pre.aux_ld = (unsigned)q->aux_ld; pre.add_ld = (unsigned)q->add_ld;
            prep = &pre;
        }
    }
    if (elem_bytes == 4)
        st = run_gather<F32>(x, kc, w, nc, tbl, ld, K, n_out, y, w_layout, ws, ws_bytes, n_in, false, res,
                             as_stream(stream), ep, &n_part, epi ? epi->tilebook : nullptr, epi ? epi->tilebook_rows : 0, prep);
    else
        st = run_gather<BF16>(x, kc, w, nc, tbl, ld, K, n_out, y, w_layout, ws, ws_bytes, n_in, y_is_f32 != 0, res,
                              as_stream(stream), ep, &n_part, epi ? epi->tilebook : nullptr,
                              epi ? epi->tilebook_rows : 0, prep);
    if (st == DODA_OK && epi && epi->stats_rows_h) *epi->stats_rows_h = n_part;
    return st;
}
