// Sparse-convolution weight gradient:  dw[o][ci][co] = sum_t a[tbl[o][t]][ci] * b[t][co].
// (spconv v1.2 indice_conv_backward's per-offset `Xg^T . dYg` GEMMs; reference call sites
// model/unet_block.py:26,29,48,70,78.)  fp32 and bf16 feature storage from one template.
//
// The contraction runs over ROWS, so both MFMA operands need the row index as their k dimension
// while memory holds channels contiguously: rows go through LDS.  Per block and per step of 64
// rows:  the dY tile [64][TB*16] is loaded once (coalesced) into a shared LDS tile and each wave
// lifts its B fragments into registers; then every wave walks ITS OWN subset of kernel offsets
// (o = wave, wave+4, ...): coalesced read of tbl[o][rows], one contiguous row-slice gather per
// lane into a wave-private LDS tile, transposed fragment reads, MFMAs into that offset's
// accumulators.  Gathers for the next offset are issued before the MFMAs of the current one.
// Offsets with no present row in the step are skipped wave-uniformly.  Accumulators (<= 7 offsets
// x TA x TB 16x16 tiles) live in registers for the whole row chunk; per-chunk partials are reduced
// by a second kernel in fixed order: deterministic, no float atomics.
#include "wgrad_pairs.hpp"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RT = 64;    // rows per step
constexpr int MAX_OGW = 7;  // offsets per wave (4 waves x 7 >= 27)
constexpr int PAD = 8;    // LDS row padding in elements (bank spread, keeps rows 16-byte aligned)

struct F32 {
    typedef float elem;
    typedef f32x4 frag;
    typedef f32x4 vec;                      // 16-byte global / LDS access unit
    static constexpr int VEC = 4;
    static constexpr int KSTEPS = RT / 4;   // v_mfma_f32_16x16x4_f32: 4 rows per MFMA
    typedef float kfrag;                    // one MFMA operand
    // fragments of all KSTEPS k-steps for the 16 channels starting at col0 (lane = (i, g))
    template <int STRIDE>
    static __device__ __forceinline__ void frags(const elem *tile, int g, int i, int col0, kfrag (&out)[KSTEPS]) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) out[ks] = tile[(ks * 4 + g) * STRIDE + col0 + i];
    }
    static __device__ __forceinline__ f32x4 mma(kfrag a, kfrag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// F32S (round 5): fp32 operands, every 16 rows as TWO v_mfma_f32_16x16x32_bf16 on bf16 head / tail splits made in registers
// (x = hi + lo + e, |e| <= 2^-17 |x|): the k = 32 of an instruction is (4 rows of the lane group) x (head, tail) —
// A = [a_hi | a_lo] against B = [b_hi | b_hi], then against [b_lo | b_lo]: all four partial products, fp32 accumulate.  The
// fp32 matrix rate of this part is 1/16 of bf16 and four v_mfma_f32_16x16x4_f32 per 16 rows were the largest item of the
// fp32 step (4.8 of 13.8 ms of kernels); same LDS reads (one float per lane, row and operand), same bytes.  Used for layers
// of many rows (plan_job); small layers keep the exact chain.
struct F32S : F32 {
    static constexpr int KSTEPS = RT / 16;
    struct kfrag { unsigned hi[2], lo[2]; };
    template <int STRIDE>
    static __device__ __forceinline__ void frags(const elem *tile, int g, int i, int col0, kfrag (&out)[KSTEPS]) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = tile[(ks * 16 + 4 * g + q) * STRIDE + col0 + i];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 a = {v[2 * h], v[2 * h + 1]};
                const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));   // round to nearest even
                const f32x2 res = {a[0] - __uint_as_float(hu << 16), a[1] - __uint_as_float(hu & 0xffff0000u)};   // exact
                out[ks].hi[h] = hu;
                out[ks].lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(res, bf16x2));
            }
        }
    }
    static __device__ __forceinline__ f32x4 mma(const kfrag &a, const kfrag &b, f32x4 c) {
        const u32x4 av = {a.hi[0], a.hi[1], a.lo[0], a.lo[1]};
        const u32x4 b1 = {b.hi[0], b.hi[1], b.hi[0], b.hi[1]}, b2 = {b.lo[0], b.lo[1], b.lo[0], b.lo[1]};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b1), c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b2), c, 0, 0, 0);
    }
};

struct BF16 {
    typedef unsigned short elem;
    typedef s16x4 frag;
    typedef u32x4 vec;
    static constexpr int VEC = 8;
    static constexpr int KSTEPS = RT / 32;  // v_mfma_f32_16x16x32_bf16: 32 rows per MFMA
    typedef bf16x8 kfrag;
    // Rows are the MFMA k dimension but LDS holds [row][channel]: ds_read_b64_tr_b16 does the 4x16
    // transpose in the LDS crossbar.  Probed on gfx950 (tools/probe/trread.hip): when lane t of a
    // 16-lane group g points at &tile[R + t/4][4*(t&3)] it receives {tile[R+q][t] : q = 0..3}.
    // A 16x16x32 operand wants k = 8g..8g+7 per lane: two reads with R = 8g and R = 8g + 4; the two
    // k-steps are 32 rows apart (immediate offsets).  The asm ends with lgkmcnt(0): hipcc does not
    // count LDS ops issued inside asm.
    template <int STRIDE>
    static __device__ __forceinline__ void frags(const elem *tile, int g, int i, int col0, kfrag (&out)[KSTEPS]) {
        static_assert(KSTEPS == 2, "two k-steps of 32 rows");
        const unsigned addr = (unsigned)(uintptr_t)(tile + (8 * g + (i >> 2)) * STRIDE + col0 + 4 * (i & 3));
        s16x4 lo0, hi0, lo1, hi1;
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                     "ds_read_b64_tr_b16 %1, %4 offset:%5\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:%6\n\t"
                     "ds_read_b64_tr_b16 %3, %4 offset:%7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(lo0), "=&v"(hi0), "=&v"(lo1), "=&v"(hi1)
                     : "v"(addr), "n"(4 * STRIDE * 2), "n"(32 * STRIDE * 2), "n"(36 * STRIDE * 2)
                     : "memory");
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 k0 = __builtin_shufflevector(lo0, hi0, 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 k1 = __builtin_shufflevector(lo1, hi1, 0, 1, 2, 3, 4, 5, 6, 7);
        out[0] = __builtin_bit_cast(bf16x8, k0);
        out[1] = __builtin_bit_cast(bf16x8, k1);
    }
    static __device__ __forceinline__ f32x4 mma(kfrag a, kfrag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// VOK: rows of both operands are 16-byte aligned multiples of 16 bytes.  Then every global access is
// an UNCONDITIONAL 16-byte load from an always-valid address (absent rows read row 0) whose value is
// zeroed by a select: no divergent branch around a load, so hipcc can keep many loads in flight
// (with branches it emitted `s_waitcnt vmcnt(0)` after every single load).
template <class T, int TA, int TB, int OGW, bool VOK>
__device__ __forceinline__ void wgrad_body(const typename T::elem *__restrict__ a, int ca,
                                           const typename T::elem *__restrict__ b, int cb,
                                           const int32_t *__restrict__ tbl, int ld, int K,
                                           int n_rows, int rows_per_chunk, int n_tag,
                                           int n_tbg, int n_og, float *__restrict__ partial, int item) {
    typedef typename T::elem elem;
    typedef typename T::frag frag;
    typedef typename T::kfrag kfrag;
    constexpr int SA = TA * 16 + PAD, SB = TB * 16 + PAD;  // LDS row strides (elements)
    __shared__ __attribute__((aligned(16))) elem b_tile[RT * SB];
    __shared__ __attribute__((aligned(16))) elem a_tile[4][RT * SA];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    // work item = (row chunk, channel-tile group), group fastest
    const int n_grp = n_tag * n_tbg * n_og;
    const int chunk = item / n_grp;
    int grp = item % n_grp;
    const int tag = grp % n_tag; grp /= n_tag;
    const int tbg = grp % n_tbg;
    const int o_base = (grp / n_tbg) * (4 * OGW);   // this block's group of 4*OGW offsets
    const int ca0 = tag * TA * 16, cb0 = tbg * TB * 16;  // channel slices of this block

    f32x4 acc[OGW][TA][TB];
#pragma unroll
    for (int oo = 0; oo < OGW; ++oo)
#pragma unroll
        for (int x_ = 0; x_ < TA; ++x_)
#pragma unroll
            for (int y_ = 0; y_ < TB; ++y_) acc[oo][x_][y_] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long long r_begin = (long long)chunk * rows_per_chunk;
    long long r_end = r_begin + rows_per_chunk;
    if (r_end > n_rows) r_end = n_rows;

    elem *my_a = a_tile[wid];

    // one lane's row slice: NVA 16-byte vectors (TA*16 channels)
    typedef typename T::vec vec;
    constexpr int VEC = T::VEC, NVA = TA * 16 / VEC, NVB = TB * 16 / VEC;
    auto load_vec = [&](const elem *p, const elem *safe, int c, int cmax, bool ok) -> vec {
        const bool live = ok && c < cmax;
        if constexpr (VOK) {
            // raw value; the CONSUMER zeroes dead lanes (a select here would make the compiler wait
            // for the load on the spot)
            return *reinterpret_cast<const vec *>(live ? p : safe);
        } else {
            vec v = vec{};
            if (live) {
                elem tmp[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) tmp[q] = (c + q < cmax) ? p[q] : (elem)0;
                v = *reinterpret_cast<const vec *>(tmp);
            }
            return v;
        }
    };
    auto gather_row = [&](int idx, vec (&dst)[NVA]) {
#pragma unroll
        for (int f = 0; f < NVA; ++f) {
            const int c = ca0 + f * VEC;
            dst[f] = load_vec(a + (long long)(idx >= 0 ? idx : 0) * ca + c, a, c, ca, idx >= 0);
        }
    };

    // Software pipeline over the 64-row steps: the dY vectors and the table entries of step n+1 are
    // requested at the start of step n and the gather of its first offset during step n's last one,
    // so a step no longer begins with three dependent memory latencies (dY tile, table, first gather)
    // — measured as the bound of this kernel (20 steps x ~3 us per block at level 1).
    constexpr int NDY = (RT * NVB + 255) / 256;   // dY vectors per thread and step
    vec dy_reg[NDY];
    int idx_n[OGW];
    auto load_dy = [&](long long r0) {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int e = threadIdx.x + k * 256;
            const int r = e / NVB, f = e - r * NVB;
            const long long row = r0 + r;
            const int c = cb0 + f * VEC;
            dy_reg[k] = load_vec(b + (row < r_end ? row : 0) * cb + c, b, c, cb, e < RT * NVB && row < r_end);
        }
    };
    auto load_idx = [&](long long r0) {
#pragma unroll
        for (int oo = 0; oo < OGW; ++oo) {
            const int o = o_base + wid + 4 * oo;
            const long long row = r0 + lane;
            idx_n[oo] = tbl[(o < K && row < r_end) ? (long long)o * ld + row : 0];   // raw; masked when consumed
        }
    };

    vec rows_cur[NVA], rows_nxt[NVA];
    if (r_begin < r_end) {
        load_dy(r_begin);
        load_idx(r_begin);
        gather_row((o_base + wid < K && r_begin + lane < r_end) ? idx_n[0] : -1, rows_cur);
    }
    for (long long r0 = r_begin; r0 < r_end; r0 += RT) {
        // ---- dY tile (already in registers) -> shared LDS ----
        doda_sync();
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int e = threadIdx.x + k * 256;
            if (e < RT * NVB) {
                const int r = e / NVB, f = e - r * NVB;
                const bool live = r0 + r < r_end && cb0 + f * VEC < cb;
                *reinterpret_cast<vec *>(&b_tile[r * SB + f * VEC]) = (VOK && !live) ? vec{} : dy_reg[k];
            }
        }
        int idx[OGW];
#pragma unroll
        for (int oo = 0; oo < OGW; ++oo)
            idx[oo] = (o_base + wid + 4 * oo < K && r0 + lane < r_end) ? idx_n[oo] : -1;
        const bool more = r0 + RT < r_end;
        if (more) {  // in flight for the whole step
            load_dy(r0 + RT);
            load_idx(r0 + RT);
        }
        doda_sync();
        kfrag bf[TB][T::KSTEPS];
#pragma unroll
        for (int y_ = 0; y_ < TB; ++y_) T::template frags<SB>(b_tile, g, i, y_ * 16, bf[y_]);

        // ---- this wave's offsets ----
#pragma unroll
        for (int oo = 0; oo < OGW; ++oo) {
            // gather for the next offset (or for the first offset of the next step) ahead of the MFMAs
            if (oo + 1 < OGW) gather_row(idx[oo + 1], rows_nxt);
            else if (more) gather_row((o_base + wid < K && r0 + RT + lane < r_end) ? idx_n[0] : -1, rows_nxt);
            if (__ballot(idx[oo] >= 0) != 0ull) {
#pragma unroll
                for (int f = 0; f < NVA; ++f) {
                    const bool live = idx[oo] >= 0 && ca0 + f * VEC < ca;
                    *reinterpret_cast<vec *>(&my_a[lane * SA + f * VEC]) = (VOK && !live) ? vec{} : rows_cur[f];
                }
                // LDS operations of one wave are processed in issue order, so the transposed reads
                // below see the stores above (and the next offset's stores cannot overtake these
                // reads) without a counter wait; only the compiler has to be kept from reordering
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int x_ = 0; x_ < TA; ++x_) {
                    kfrag af[T::KSTEPS];
                    T::template frags<SA>(my_a, g, i, x_ * 16, af);
#pragma unroll
                    for (int ks = 0; ks < T::KSTEPS; ++ks)
#pragma unroll
                        for (int y_ = 0; y_ < TB; ++y_)
                            acc[oo][x_][y_] = T::mma(af[ks], bf[y_][ks], acc[oo][x_][y_]);
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int f = 0; f < NVA; ++f) rows_cur[f] = rows_nxt[f];
        }
    }

    // D[i = ci][j = co]: lane (co = lane&15, g) holds ci = 4g + r
    float *out = partial + (long long)chunk * K * ca * cb;
#pragma unroll
    for (int oo = 0; oo < OGW; ++oo) {
        const int o = o_base + wid + 4 * oo;
        if (o < K) {
#pragma unroll
            for (int x_ = 0; x_ < TA; ++x_)
#pragma unroll
                for (int y_ = 0; y_ < TB; ++y_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = ca0 + x_ * 16 + 4 * g + r, co = cb0 + y_ * 16 + i;
                        if (ci < ca && co < cb)
                            out[((long long)o * ca + ci) * cb + co] = acc[oo][x_][y_][r];
                    }
        }
    }
}

// ---- many layers in one launch --------------------------------------------------------------
// The weight gradients of a network are independent of the rest of the backward pass.  Queued and
// launched together (one launch per kernel variant, one reduce launch), the 71 x 2 launches of a U-Net
// step become ~8, and the coarse levels' small grids run side by side instead of one after another.
struct WJob {            // device descriptor of one layer inside a variant group
    const void *a, *b;
    const int32_t *tbl;
    float *out;          // partials of the job (or dw itself when it has a single row chunk)
    int ca, cb, ld, K, n_rows, rows_per_chunk, n_tag, n_tbg, n_og, blk_end;   // blk_end: inclusive prefix
};
struct RJob {            // one reduction: dw[q] (+)= sum_r partial[r][q]
    const float4 *partial;
    float4 *dw;
    long long n_quad;
    int R, blk_end, accumulate, pad;
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

template <class T, int TA, int TB, int OGW, bool VOK>
__global__ __launch_bounds__(256) void wgrad_multi_kernel(const WJob *__restrict__ jobs, int n_jobs) {
    const int j = find_job(jobs, n_jobs, (int)blockIdx.x);
    const WJob d = jobs[j];
    const int first = j == 0 ? 0 : jobs[j - 1].blk_end;
    wgrad_body<T, TA, TB, OGW, VOK>((const typename T::elem *)d.a, d.ca, (const typename T::elem *)d.b, d.cb,
                                    d.tbl, d.ld, d.K, d.n_rows, d.rows_per_chunk, d.n_tag, d.n_tbg, d.n_og,
                                    d.out, (int)blockIdx.x - first);
}

// (256 / RL) element quads x RL chunk lanes per block, fixed order (as wgrad_reduce4<16>): lane r sums chunks r, r + RL, ..., the
// lane sums are added in ascending r.  RL = d.pad = the smallest power of two >= min(R, 16) (round 6; it was 16 for every job: the
// coarse levels' jobs have 1 .. 4 chunks, so 3/4 .. 15/16 of a block's threads had nothing to read and a 7.5 M-parameter network
// took 117 k blocks of 256 bytes each).  For R <= 16 every lane holds at most one chunk either way: the same sums, bit for bit.
__global__ __launch_bounds__(256) void wgrad_reduce_multi(const RJob *__restrict__ jobs, int n_jobs) {
    __shared__ float4 part[256];
    const int j = find_job(jobs, n_jobs, (int)blockIdx.x);
    const RJob d = jobs[j];
    const int first = j == 0 ? 0 : jobs[j - 1].blk_end;
    const int RL = d.pad, EL = 256 / RL;                 // RL in {1, 2, 4, 8, 16}
    const int el = (int)threadIdx.x % EL, rl = (int)threadIdx.x / EL;
    const long long q = (long long)((int)blockIdx.x - first) * EL + el;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d.n_quad)
        for (int r = rl; r < d.R; r += RL) {
            const float4 v = d.partial[(long long)r * d.n_quad + q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[rl * EL + el] = s;
    doda_sync();
    if (rl == 0 && q < d.n_quad) {
        float4 t = part[el];
        for (int r = 1; r < RL; ++r) {
            const float4 v = part[r * EL + el];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        if (d.accumulate) {
            const float4 old = d.dw[q];
            t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
        }
        d.dw[q] = t;
    }
}

// Fixed-order reduction of the per-chunk partials: 16 element lanes x 16 chunk lanes per block;
// lane r sums chunks r, r+16, ... and the 16 lane sums are added in ascending r (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce(const float *__restrict__ partial, int R,
                                                    long long n_elem, float *__restrict__ dw,
                                                    int accumulate = 0) {
    __shared__ float part[16][17];
    const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long long e = (long long)blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < n_elem)
        for (int r = rl; r < R; r += 16) s += partial[(long long)r * n_elem + e];
    part[rl][el] = s;
    doda_sync();
    if (rl == 0 && e < n_elem) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][el];
        dw[e] = accumulate ? dw[e] + t : t;
    }
}


struct Plan {
    int TA, TB, OGW, n_og, n_tag, n_tbg, R, rows_per_chunk;
};
static const int MULTI_MIN_ROWS = getenv("DODA_WGRAD_MIN_ROWS") ? atoi(getenv("DODA_WGRAD_MIN_ROWS")) : 512;

// multi: the job is one of many in a doda_spconv_wgrad_multi launch — the other layers fill the chip,
// so a layer needs far fewer row chunks (each chunk costs K*ca*cb*4 bytes of partials to write and reduce)
Plan make_plan(int K, int ca, int cb, int n_rows, int elem_bytes, bool multi = false) {
    Plan p;
    const int ta = (ca + 15) / 16, tb = (cb + 15) / 16;
    p.TA = (ta % 2 == 0) ? 2 : 1;
    p.TB = (tb % 2 == 0) ? 2 : 1;
    // 48-channel operands (level 3 of the U-Net): one 3 x 3 tile block gathers every row once instead of nine
    // 1 x 1 blocks gathering a third of it each (and reading the table nine times)
    static const bool no33 = getenv("DODA_WGRAD_NO33") && getenv("DODA_WGRAD_NO33")[0] == '1';
    if (ta == 3 && tb == 3 && elem_bytes == 2 && K > 8 && !no33) { p.TA = 3; p.TB = 3; }
    p.n_tag = ta / p.TA;
    p.n_tbg = tb / p.TB;
    // 2x2 accumulator tiles x 7 offsets would need 112 accumulator registers (1 wave/SIMD): give
    // such blocks 4 offsets per wave and spread the offsets over several block groups instead
    // offsets per wave for 1- and 2-tile blocks (rocprofv3, levels 1 / 3 / 5): bf16 7 -> 4 offsets
    // 54.8 -> 52.2, 39.6 -> 35.3, 19.0 -> 14.3 us (fewer registers, half the partials); fp32 the other
    // way round (91 vs 106 us at level 3): its 16 dY fragment reads per step amortise over more offsets
    p.OGW = (p.TA * p.TB == 4 || elem_bytes == 2) ? 4 : MAX_OGW;
    if (p.TA * p.TB == 9) p.OGW = 2;   // 72 accumulator registers
    if (K <= 8) p.OGW = (p.TA * p.TB == 4) ? 2 : 2;
    p.n_og = div_up(K, 4 * p.OGW);
    const int gy = p.n_tag * p.n_tbg * p.n_og;
    const int rows = n_rows > 0 ? n_rows : 1;
    // measured at M = 600k / 183k (rocprofv3): 1x1 and 2x1 tiles 68 -> 50 us going from 512 to 1024
    // blocks (+5 us of partial reduce); 2x2 tiles are fastest at 512
    // multi (whole U-Net step, wgrad + reduce): 1024 -> 1.86 ms, 512 -> 1.67, 256 -> 1.67, 128 -> 1.99
    // (3 x 3 blocks: a block does nine tiles' worth of work and its chunk writes all K*ca*cb partials)
    static const int t33 = getenv("DODA_WGRAD_T33") ? atoi(getenv("DODA_WGRAD_T33")) : 128;
    const int target = p.TA * p.TB == 9 ? t33 : (multi ? 512 : ((p.TA * p.TB == 4) ? 512 : 1024));
    int R = div_up(target, gy);         // blocks over the whole grid
    // multi: at least MULTI_MIN_ROWS rows per chunk.  Every chunk writes K*ca*cb*4 bytes of partials; with
    // 512 blocks per job the coarse levels (64..112 channels, a few thousand rows) wrote and re-read
    // 5-28 MB per layer for a few hundred rows per chunk, and the launch holds ~40 other jobs to fill the
    // chip with anyway
    const int max_r = multi ? (rows / MULTI_MIN_ROWS > 1 ? rows / MULTI_MIN_ROWS : 1) : div_up(rows, RT);
    if (R > max_r) R = max_r;
    if (R < 1) R = 1;
    p.rows_per_chunk = div_up(div_up(rows, R), RT) * RT;
    p.R = div_up(rows, p.rows_per_chunk);
    (void)K;
    return p;
}

// ---- host side of the many-layer launch -------------------------------------------------------
struct JobPlan {
    Plan p;
    int esz, vok, key;
    int split;         // fp32 job multiplied as bf16 head / tail splits (F32S)
    size_t ws_off;     // partials of this job inside the workspace (unused when R == 1)
    long long n_elem;
};

bool plan_job(const doda_wgrad_job &j, JobPlan *out) {
    if (j.ca <= 0 || j.cb <= 0 || j.K <= 0 || j.K > 4 * MAX_OGW || j.n_rows <= 0 || j.ld < j.n_rows || !j.a ||
        !j.b || !j.tbl || !j.dw || (j.elem_bytes != 2 && j.elem_bytes != 4))
        return false;
    out->esz = j.elem_bytes;
    out->vok = ((size_t)j.ca * j.elem_bytes % 16 == 0) && ((size_t)j.cb * j.elem_bytes % 16 == 0) &&
               ((uintptr_t)j.a % 16 == 0) && ((uintptr_t)j.b % 16 == 0);
    out->p = make_plan(j.K, j.ca, j.cb, j.n_rows, j.elem_bytes, true);
    {   // OPT-IN (see spconv_gather.hip run_gather): fp32 jobs of at least DODA_F32_WGRAD_SPLIT_ROWS rows (0: all of them — one
        // instantiation for every fp32 job keeps the layers of a step in shared launches; unset / -1: none, the exact chain)
        static const long long min_rows = [] { const char *e = getenv("DODA_F32_WGRAD_SPLIT_ROWS"); return e && *e ? atoll(e) : -1ll; }();
        out->split = (j.elem_bytes == 4 && min_rows >= 0 && (long long)j.n_rows >= min_rows) ? 1 : 0;
    }
    out->key = ((((j.elem_bytes * 4 + out->p.TA) * 4 + out->p.TB) * 8 + out->p.OGW) * 2 + out->vok) * 2 + out->split;
    out->n_elem = (long long)j.K * j.ca * j.cb;
    return true;
}

template <class T>
void launch_multi_variant(const Plan &p, int vok, int total_blocks, const WJob *jobs_dev, int n, hipStream_t s) {
    const dim3 grid(total_blocks), block(256);
#define GO(TA, TB, OG)                                                                             \
    do {                                                                                           \
        if (vok) hipLaunchKernelGGL((wgrad_multi_kernel<T, TA, TB, OG, true>), grid, block, 0, s, jobs_dev, n); \
        else hipLaunchKernelGGL((wgrad_multi_kernel<T, TA, TB, OG, false>), grid, block, 0, s, jobs_dev, n); \
    } while (0)
    if (p.OGW == 2) {
        if (p.TA == 3 && p.TB == 3) GO(3, 3, 2);
        else if (p.TA == 1 && p.TB == 1) GO(1, 1, 2);
        else if (p.TA == 2 && p.TB == 1) GO(2, 1, 2);
        else if (p.TA == 1 && p.TB == 2) GO(1, 2, 2);
        else GO(2, 2, 2);
    } else if (p.OGW == 4 && p.TA == 1 && p.TB == 1) GO(1, 1, 4);
    else if (p.OGW == 4 && p.TA == 2 && p.TB == 1) GO(2, 1, 4);
    else if (p.OGW == 4 && p.TA == 1 && p.TB == 2) GO(1, 2, 4);
    else if (p.TA == 1 && p.TB == 1) GO(1, 1, 7);
    else if (p.TA == 2 && p.TB == 1) GO(2, 1, 7);
    else if (p.TA == 1 && p.TB == 2) GO(1, 2, 7);
    else GO(2, 2, 4);
#undef GO
}

// pinned staging for the descriptor upload (one per process, grow-only, reuse guarded by an event)
struct Staging {
    std::mutex mu;
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int ev_dev = -1;   // device the event was created on
    bool pending = false;
};
Staging g_staging;
}  // namespace

// job classes of a multi call
enum { J_SKIP = 0, J_ZERO = 1, J_DENSE = 2, J_PAIRS = 3, J_TILE = 4 };

// rows from which a rulebook's tile jobs take the LDS-staged kernel even when pair lists are at hand (measurement aid:
// DODA_WDMA_MIN_ROWS)
static int wdma_min_rows() {
    static const int v = getenv("DODA_WDMA_MIN_ROWS") ? atoi(getenv("DODA_WDMA_MIN_ROWS")) : 32768;
    return v;
}

static int classify(const doda_wgrad_job &j) {
    if (j.n_rows == 0 && j.dw && j.K > 0 && j.ca > 0 && j.cb > 0)
        return (j.flags & DODA_WGRAD_ACCUMULATE) ? J_SKIP : J_ZERO;
    // a tilebook of the job's table: the LDS-staged kernel (bf16, K = 27; 16 -> 16, and — round 4 — 16 .. 64 channels on
    // either side as 16 x 16 channel blocks over row-strided slices)
    if (j.tilebook && j.tbl && j.elem_bytes == 2 && j.ca % 16 == 0 && j.ca <= 64 && j.cb % 16 == 0 && j.cb <= 64 &&
        j.K == 27 && j.n_rows > 0 && j.a && j.b && j.dw &&
        j.n_a == j.n_rows && j.ld >= j.n_rows && (size_t)j.n_rows * 128 < 0x7ffffff0ull && (size_t)j.K * j.ld * 4 < 0xffffffffull &&
        !(((uintptr_t)j.a | (uintptr_t)j.b | (uintptr_t)j.tilebook) & 15) && doda_wdma::enabled() &&
        // round 4 (block-major chunks: one or two partials per workgroup whatever the number of layers): faster than the pair
        // lists from ~40 k rows up — 8 layers per call: 601 k rows 21.3 / 43.3 us per layer, 152 k rows 7.2 / 13.0, level 2
        // (154 k rows, 32 -> 32 as four blocks) 21.3 / 23.7, 37 k rows 10.3 / 10.1 (tools/wl2.py).  Round 3's schedule (every
        // workgroup walked every layer: a flush per layer and workgroup) lost below 262 k rows.
        (j.n_rows >= wdma_min_rows() || !doda_pairs::eligible(j)))
        return J_TILE;
    // DODA_WGRAD_NO_PAIRS=1 keeps every job on the gather-table kernel (A/B measurements)
    static const bool no_pairs = getenv("DODA_WGRAD_NO_PAIRS") && getenv("DODA_WGRAD_NO_PAIRS")[0] == '1';
    if (doda_pairs::eligible(j) && (!no_pairs || !j.tbl)) return J_PAIRS;
    return J_DENSE;
}

static size_t tile_blocks(const doda_wgrad_job &j) { return (size_t)(j.ca / 16) * (j.cb / 16); }

static bool dense_needs_partial(const JobPlan &jp, const doda_wgrad_job &j) {
    return jp.p.R > 1 || (j.flags & DODA_WGRAD_ACCUMULATE);
}

extern "C" size_t doda_spconv_wgrad_multi_workspace_bytes(const doda_wgrad_job *jobs_h, int32_t n_jobs) {
    if (!jobs_h || n_jobs <= 0) return 0;
    size_t total = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const int cls = classify(jobs_h[k]);
        if (cls == J_PAIRS) { total += doda_pairs::partial_bytes(jobs_h[k]); continue; }
        if (cls == J_TILE) { total += tile_blocks(jobs_h[k]) * align_up(doda_wdma::partial_bytes(jobs_h[k].n_rows), 256); continue; }
        if (cls != J_DENSE) continue;
        JobPlan jp;
        if (!plan_job(jobs_h[k], &jp)) continue;
        if (dense_needs_partial(jp, jobs_h[k])) total += align_up((size_t)jp.p.R * jp.n_elem * 4, 256);
    }
    return total < 256 ? 256 : total;
}

extern "C" size_t doda_spconv_wgrad_multi_desc_bytes(int32_t n_jobs) {
    const size_t per = sizeof(WJob) + sizeof(RJob) > doda_pairs::desc_bytes_per_job()
                           ? sizeof(WJob) + sizeof(RJob) : doda_pairs::desc_bytes_per_job();
    return n_jobs <= 0 ? 0 : align_up((size_t)n_jobs * per + 256, 256);
}

extern "C" int doda_spconv_wgrad_multi(const doda_wgrad_job *jobs_h, int32_t n_jobs, void *ws, size_t ws_bytes,
                                       void *desc_dev, size_t desc_bytes, doda_stream_t stream) {
    if (!jobs_h || n_jobs <= 0 || !ws || !desc_dev) return DODA_ERR_INVALID;
    if (desc_bytes < doda_spconv_wgrad_multi_desc_bytes(n_jobs)) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    std::vector<JobPlan> plans(n_jobs);
    std::vector<int> keys, cls(n_jobs), pair_jobs, tile_jobs;
    size_t off = 0;
    for (int k = 0; k < n_jobs; ++k) {
        plans[k].key = -1;
        cls[k] = classify(jobs_h[k]);
        if (cls[k] == J_ZERO) {
            hipMemsetAsync(jobs_h[k].dw, 0, (size_t)jobs_h[k].K * jobs_h[k].ca * jobs_h[k].cb * 4, s);
            continue;
        }
        if (cls[k] == J_SKIP) continue;
        if (cls[k] == J_PAIRS) { pair_jobs.push_back(k); continue; }
        if (cls[k] == J_TILE) { tile_jobs.push_back(k); continue; }
        if (!plan_job(jobs_h[k], &plans[k])) return DODA_ERR_INVALID;
        plans[k].ws_off = off;
        if (dense_needs_partial(plans[k], jobs_h[k])) off += align_up((size_t)plans[k].p.R * plans[k].n_elem * 4, 256);
        bool seen = false;
        for (int key : keys) seen |= key == plans[k].key;
        if (!seen) keys.push_back(plans[k].key);
    }
    doda_pairs::Prepared prep;
    if (!pair_jobs.empty()) {
        const int st = doda_pairs::prepare(jobs_h, pair_jobs.data(), (int)pair_jobs.size(), (char *)ws, &off, &prep);
        if (st != DODA_OK) return st;
    }
    // tile jobs: one launch per rulebook (jobs sharing table + tilebook): their 16 x 16 channel blocks in queue order, the
    // blocks' partials contiguous behind everything else
    {
        std::vector<char> done(tile_jobs.size(), 0);
        for (size_t q = 0; q < tile_jobs.size(); ++q) {
            if (done[q]) continue;
            const doda_wgrad_job &j0 = jobs_h[tile_jobs[q]];
            std::vector<doda_wdma::Block> blocks;
            for (size_t r = q; r < tile_jobs.size(); ++r) {
                const doda_wgrad_job &j = jobs_h[tile_jobs[r]];
                if (done[r] || j.tilebook != j0.tilebook || j.tbl != j0.tbl || j.n_rows != j0.n_rows || j.ld != j0.ld) continue;
                done[r] = 1;
                const int es = 2;
                for (int ci = 0; ci < j.ca; ci += 16)
                    for (int co = 0; co < j.cb; co += 16)
                        blocks.push_back(doda_wdma::Block{(const char *)j.a + ci * es, (const char *)j.b + co * es,
                                                          j.dw + (size_t)ci * j.cb + co, j.ca * es, j.cb * es, j.ca * j.cb, j.cb,
                                                          (j.flags & DODA_WGRAD_ACCUMULATE) ? 1 : 0});
            }
            const size_t bytes = blocks.size() * align_up(doda_wdma::partial_bytes(j0.n_rows), 256);
            if (ws_bytes < off + bytes) return DODA_ERR_WORKSPACE;
            const int st = doda_wdma::launch(blocks.data(), (int)blocks.size(), j0.tbl, j0.ld, j0.n_rows, j0.tilebook,
                                             (char *)ws + off, s);
            if (st != DODA_OK) return st;
            off += bytes;
        }
    }
    if (ws_bytes < off) return DODA_ERR_WORKSPACE;

    // dense descriptors grouped by kernel variant, then the reductions
    std::vector<WJob> wj;
    std::vector<RJob> rj;
    struct Group { int first, count, blocks, rep; };
    std::vector<Group> groups;
    for (int key : keys) {
        Group g{(int)wj.size(), 0, 0, -1};
        for (int k = 0; k < n_jobs; ++k) {
            if (cls[k] != J_DENSE || plans[k].key != key) continue;
            const doda_wgrad_job &j = jobs_h[k];
            const Plan &p = plans[k].p;
            if (g.rep < 0) g.rep = k;
            WJob d;
            d.a = j.a; d.b = j.b; d.tbl = j.tbl;
            d.out = dense_needs_partial(plans[k], j) ? (float *)((char *)ws + plans[k].ws_off) : j.dw;
            d.ca = j.ca; d.cb = j.cb; d.ld = j.ld; d.K = j.K; d.n_rows = j.n_rows;
            d.rows_per_chunk = p.rows_per_chunk; d.n_tag = p.n_tag; d.n_tbg = p.n_tbg; d.n_og = p.n_og;
            g.blocks += p.R * p.n_tag * p.n_tbg * p.n_og;
            d.blk_end = g.blocks;
            wj.push_back(d);
            ++g.count;
        }
        groups.push_back(g);
    }
    int r_blocks = 0;
    bool scalar_reduce = false;
    for (int k = 0; k < n_jobs; ++k) {
        if (cls[k] != J_DENSE || !dense_needs_partial(plans[k], jobs_h[k])) continue;
        if (plans[k].n_elem % 4 != 0 || (uintptr_t)jobs_h[k].dw % 16 != 0) { scalar_reduce = true; continue; }
        RJob d;
        d.partial = (const float4 *)((char *)ws + plans[k].ws_off);
        d.dw = (float4 *)jobs_h[k].dw;
        d.n_quad = plans[k].n_elem / 4;
        d.R = plans[k].p.R;
        d.accumulate = (jobs_h[k].flags & DODA_WGRAD_ACCUMULATE) ? 1 : 0;
        d.pad = 0;
        int rl = 1;
        while (rl < 16 && rl < d.R) rl *= 2;
        d.pad = rl;                                        // chunk lanes per block (wgrad_reduce_multi)
        r_blocks += (int)div_up(d.n_quad, 256 / rl);
        d.blk_end = r_blocks;
        rj.push_back(d);
    }
    const size_t wbytes = wj.size() * sizeof(WJob), rbytes = rj.size() * sizeof(RJob);
    const size_t pair_off = align_up(wbytes + rbytes, 16), pbytes = prep.desc.size();
    const size_t total_desc = pair_off + pbytes;
    if (total_desc > desc_bytes) return DODA_ERR_WORKSPACE;
    if (total_desc > 0) {
        std::lock_guard<std::mutex> lock(g_staging.mu);
        Staging &st = g_staging;
        if (st.pending) { hipEventSynchronize(st.ev); st.pending = false; }
        if (st.cap < total_desc) {
            if (st.host) hipHostFree(st.host);
            st.cap = align_up(total_desc, 4096) * 2;
            if (hipHostMalloc(&st.host, st.cap, hipHostMallocDefault) != hipSuccess) { st.host = nullptr; st.cap = 0; return DODA_ERR_NOMEM; }
        }
        int cur_dev = 0;
        hipGetDevice(&cur_dev);
        if (st.ev && st.ev_dev != cur_dev) { hipEventDestroy(st.ev); st.ev = nullptr; }   // library used from another device
        if (!st.ev && hipEventCreateWithFlags(&st.ev, hipEventDisableTiming) != hipSuccess) return DODA_ERR_LAUNCH;
        st.ev_dev = cur_dev;
        if (wbytes) memcpy(st.host, wj.data(), wbytes);
        if (rbytes) memcpy((char *)st.host + wbytes, rj.data(), rbytes);
        if (pbytes) memcpy((char *)st.host + pair_off, prep.desc.data(), pbytes);
        if (hipMemcpyAsync(desc_dev, st.host, total_desc, hipMemcpyHostToDevice, s) != hipSuccess) return DODA_ERR_LAUNCH;
        hipEventRecord(st.ev, s);
        st.pending = true;
    }
    const WJob *wj_dev = (const WJob *)desc_dev;
    for (const Group &g : groups) {
        const JobPlan &jp = plans[g.rep];
        if (jp.esz == 4 && jp.split) launch_multi_variant<F32S>(jp.p, jp.vok, g.blocks, wj_dev + g.first, g.count, s);
        else if (jp.esz == 4) launch_multi_variant<F32>(jp.p, jp.vok, g.blocks, wj_dev + g.first, g.count, s);
        else launch_multi_variant<BF16>(jp.p, jp.vok, g.blocks, wj_dev + g.first, g.count, s);
    }
    int st = doda_check_launch();
    if (st != DODA_OK) return st;
    if (!pair_jobs.empty()) {
        st = doda_pairs::launch(prep, (const char *)desc_dev + pair_off, s);
        if (st != DODA_OK) return st;
    }
    if (!rj.empty()) {
        hipLaunchKernelGGL(wgrad_reduce_multi, dim3(r_blocks), dim3(256), 0, s,
                           (const RJob *)((const char *)desc_dev + wbytes), (int)rj.size());
        st = doda_check_launch();
        if (st != DODA_OK) return st;
    }
    if (scalar_reduce)   // odd element counts: the per-layer scalar reduce
        for (int k = 0; k < n_jobs; ++k) {
            if (cls[k] != J_DENSE || !dense_needs_partial(plans[k], jobs_h[k])) continue;
            if (plans[k].n_elem % 4 == 0 && (uintptr_t)jobs_h[k].dw % 16 == 0) continue;
            hipLaunchKernelGGL(wgrad_reduce, dim3(div_up(plans[k].n_elem, 16)), dim3(256), 0, s,
                               (const float *)((char *)ws + plans[k].ws_off), plans[k].p.R, plans[k].n_elem,
                               jobs_h[k].dw, (jobs_h[k].flags & DODA_WGRAD_ACCUMULATE) ? 1 : 0);
        }
    return doda_check_launch();
}
