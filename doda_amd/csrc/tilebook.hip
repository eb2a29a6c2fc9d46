// Tilebook builder: the tile-local form of a SubM gather table (tilebook.hpp) for the LDS-staged
// convolution kernel.  One workgroup per tile of TB_T output rows:
//   1. the tile's K x TB_T table entries are de-duplicated in an LDS hash set (atomicCAS, linear probing);
//   2. the distinct rows are compacted (slot order) and sorted ascending by a bitonic network in LDS —
//      the sorted list is what makes the kernel's row loads coalesce (runs of consecutive rows) and makes
//      the result independent of the order in which the atomics landed (deterministic bytes);
//   3. every entry is replaced by its position in the sorted list (rank stored next to the hash slot).
// A tile with more than TB_UMAX distinct rows only records its count; the convolution kernel then reads
// the dense table for that tile.  spconv has no counterpart (it keeps pair lists and gathers per offset);
// the dense table stays the source of truth and the reference for parity (tests/test_gpu_tile.py).
#include "common.hpp"
#include <stdlib.h>
#include "tilebook.hpp"

#ifdef DODA_TB_DEBUG
__device__ unsigned g_tb_dbg[8 + 2 * 2048];   // [0] hits, [1] tile, [2] U, [3] P, [4] where; then the keys before / after the sort
#endif
namespace {

constexpr int HCAP = 4096;   // hash slots: the insert loop stops adding once TB_UMAX + 256 keys are in
constexpr unsigned EMPTY = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void tilebook_build(const int32_t *__restrict__ tbl, int ld, int n,
                                                      TileBookView v) {
    constexpr int UQ = 2048;          // sort buffer: the power of two above TB_UMAX
    // hash slots and sort buffer back to back: the bitmap form uses both as ONE array of BMW words (196608 bits)
    constexpr int BMW = HCAP + UQ;
    __shared__ unsigned hq[BMW];
    unsigned *const htab = hq, *const uq = hq + HCAP;
    static_assert(TB_UMAX <= UQ, "sort buffer");
    __shared__ unsigned short hrank[BMW];
    __shared__ int cnt;
    __shared__ int wsum[4];
    const int tile = blockIdx.x, tid = threadIdx.x, t0 = tile * TB_T;
    int e[TB_K];
    const bool rok = t0 + tid < n;
#pragma unroll
    for (int o = 0; o < TB_K; ++o) e[o] = rok ? tbl[(long long)o * ld + t0 + tid] : -1;

    // ---- 0. (round 3) the BITMAP form: no hash, no sort -------------------------------------------------------------
    // Rows of a scan arrive in an order with locality (the property the tilebook exists for), so the rows a tile
    // references span a limited range of row numbers: [lo, hi] (bench scene: median 82k, 99 % below 150k — the voxel
    // order follows the scan's surfaces, x-neighbours lie tens of thousands of rows apart).  When that span fits the
    // 196608 bits of the hash + sort arrays, one bit per row — set with LDS atomic ORs, any order, same result — IS the sorted set of distinct rows: a
    // prefix sum over the words' population counts gives every row its rank, the list is the set bits in order, and an
    // entry's local index is prefix[word] + popcount(bits below).  The hash + 55-stage bitonic sort + rank lookup below
    // (90-108 us per level-1 table, LDS-throughput bound) remains for tiles whose rows are scattered.
    {
        int mn = 0x7fffffff, mx = -1;
#pragma unroll
        for (int o = 0; o < TB_K; ++o)
            if (e[o] >= 0) { mn = e[o] < mn ? e[o] : mn; mx = e[o] > mx ? e[o] : mx; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        __shared__ int wmn[4], wmx[4];
        if ((tid & 63) == 0) { wmn[tid >> 6] = mn; wmx[tid >> 6] = mx; }
        doda_sync();
        const int lo = min(min(wmn[0], wmn[1]), min(wmn[2], wmn[3])), hi = max(max(wmx[0], wmx[1]), max(wmx[2], wmx[3]));
        const long long span = (long long)hi - lo + 1;
        if (hi >= 0 && span <= (long long)BMW * 32) {
            const int W = (int)((span + 31) >> 5);              // words in use
            const int wpt = (W + 255) / 256;                    // consecutive words per thread (<= 24)
            for (int k = 0; k < wpt; ++k) { const int w = tid * wpt + k; if (w < W) htab[w] = 0u; }
            doda_sync();
#pragma unroll
            for (int o = 0; o < TB_K; ++o)
                if (e[o] >= 0) atomicOr(&htab[(unsigned)(e[o] - lo) >> 5], 1u << ((unsigned)(e[o] - lo) & 31u));
            doda_sync();
            int c = 0;
            for (int k = 0; k < wpt; ++k) { const int w = tid * wpt + k; if (w < W) c += __popc(htab[w]); }
            const int incl = wave_inclusive_sum(c);
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            doda_sync();
            int base = incl - c;
            for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
            const int U = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tid == 0) {
                v.ucount[tile] = U;
                if (U > TB_CAP64) atomicAdd(&v.n_over[0], 1);
                if (U > TB_LMAX) atomicAdd(&v.n_over[1], 1);
            }
            int32_t *ul = v.ulist + (size_t)tile * TB_UMAX;
            if (U > TB_LMAX) {
                for (int k = tid; k < TB_UMAX; k += 256) ul[k] = -2;
                return;
            }
            // per-word exclusive prefix (hrank) ...
            int run = base;
            for (int k = 0; k < wpt; ++k) {
                const int w = tid * wpt + k;
                if (w >= W) break;
                hrank[w] = (unsigned short)run;
                run += __popc(htab[w]);
            }
            for (int k = U + tid; k < TB_UMAX; k += 256) ul[tb_upos(k)] = -1;
            doda_sync();
            // ... and the list: the set bits in order.  The rows come in a few dense runs (hundreds of consecutive row
            // numbers = ten consecutive FULL words), so the words are dealt out by BYTES, interleaved over the threads —
            // a thread that walked its own consecutive words emitted a whole run alone while the others idled
            for (int it = tid; it < 4 * W; it += 256) {
                const int w = it >> 2, q = it & 3;
                const unsigned word = htab[w];
                unsigned bits = (word >> (8 * q)) & 0xffu;
                if (!bits) continue;
                int k = hrank[w] + __popc(word & ((1u << (8 * q)) - 1u));
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    ul[tb_upos(k++)] = lo + w * 32 + 8 * q + b;
                }
            }
            // the row's ten packed words, one per plane (three 10-bit local indices each; tilebook.hpp)
            uint32_t *li = v.lidx + (size_t)tile * (TB_T * TB_LW) + tb_lpos(tid);
#pragma unroll
            for (int w3 = 0; w3 < TB_LW; ++w3) {
                unsigned word = 0u;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int oo = 2 * (3 * (w3 % 5) + q) + w3 / 5, o = oo < TB_K ? oo : 0;
                    unsigned r = 0u;   // absent (or no such offset): LDS slot 0, the zero row
                    if (oo < TB_K && e[o] >= 0) {
                        const unsigned d = (unsigned)(e[o] - lo), w = d >> 5;
                        r = (unsigned)(hrank[w] + __popc(htab[w] & ((1u << (d & 31u)) - 1u)) + 1);
                    }
                    word |= r << (10 * q);
                }
                li[w3 * TB_T] = word;
            }
            return;
        }
    }

    for (int k = tid; k < HCAP; k += 256) htab[k] = EMPTY;
    if (tid == 0) cnt = 0;
    doda_sync();

    // ---- 1. distinct rows ----  (loops over o stay unrolled: e[] must live in registers)
#pragma unroll
    for (int o = 0; o < TB_K; ++o) {
        if (e[o] < 0) continue;
        if (*(volatile int *)&cnt > TB_LMAX) break;   // overflow: the count is all that is kept
        const unsigned key = (unsigned)e[o];
        unsigned slot = hash_mix(key) & (HCAP - 1);
        for (;;) {
            const unsigned old = atomicCAS(&htab[slot], EMPTY, key);
            if (old == EMPTY) { atomicAdd(&cnt, 1); break; }
            if (old == key) break;
            slot = (slot + 1) & (HCAP - 1);
        }
    }
    doda_sync();
    const int U = cnt;
    if (tid == 0) {
        v.ucount[tile] = U;
        if (U > TB_CAP64) atomicAdd(&v.n_over[0], 1);
        if (U > TB_LMAX) atomicAdd(&v.n_over[1], 1);
    }
    int32_t *ul = v.ulist + (size_t)tile * TB_UMAX;
    if (U > TB_LMAX) {
        for (int k = tid; k < TB_UMAX; k += 256) ul[k] = -2;   // no list: every reader sees the marker in its own entries
        return;
    }

    // ---- 2. compact (slot order), sort ascending ----
    {
        unsigned mine[HCAP / 256];
        int c = 0;
#pragma unroll
        for (int k = 0; k < HCAP / 256; ++k) {
            mine[k] = htab[tid * (HCAP / 256) + k];
            c += mine[k] != EMPTY;
        }
        const int incl = wave_inclusive_sum(c);
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        for (int k = tid; k < UQ; k += 256) uq[k] = EMPTY;
        doda_sync();
        int base = incl - c;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
#pragma unroll
        for (int k = 0; k < HCAP / 256; ++k)
            if (mine[k] != EMPTY) uq[base++] = mine[k];
        doda_sync();
    }
    int P = 2;                    // sort only the power of two that holds the keys (EMPTY pads sort last)
    while (P < U) P <<= 1;
#ifdef DODA_TB_DEBUG
    __shared__ unsigned uq0[2048];
    __shared__ int wP[4], wStages[4];
    for (int k = tid; k < 2048; k += 256) uq0[k] = uq[k];
    if ((tid & 63) == 0) { wP[tid >> 6] = P * 10000 + U; wStages[tid >> 6] = 0; }
    doda_sync();
#endif
    // compare-exchange p of a stage pairs idx = insert-zero-bit(p, j) with idx | j; a workgroup barrier in front of EVERY stage.
    // (Round 3 elided the barriers of the stages with j <= 64 — both elements of a pair then lie in the 128-element chunk the
    // thread's own wave handles, and LDS operations of one wave execute in order — 12 barriers instead of 55.  Round 5 found that
    // sort WRONG about once in 50 000 tiles: a list with one key twice and its neighbour missing, the missing row's local
    // indices then pointing at whatever rank the hash slot held — wrong neighbours, sometimes garbage — in ~1 of 400 builds of a
    // Z-ordered 2 M-voxel table (131 of its tiles take this path; tools/rbdet.py).  The elision was not the cause (next comment),
    // but it rested on the same unchecked assumption about what the compiler waits for, and only tiles whose rows span more than
    // the bitmap form covers come here: the barriers cost nothing measurable.)
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            // (explicit: hipcc emitted this in-loop barrier as a bare s_barrier — the compare-exchange's ds_write_b32 pair sits in a
            // conditionally executed block and its wait-count pass lost them across the back edge — so a wave could arrive with its
            // stores still in flight and the next stage read the old keys: THE bug above, with or without the elision)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            doda_sync();
#ifdef DODA_TB_DEBUG
            if ((tid & 63) == 0) wStages[tid >> 6] += 1;
#endif
            for (int p = tid; p < (P >> 1); p += 256) {
                const int idx = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = idx | j;
                const bool up = (idx & k) == 0;
                const unsigned a = uq[idx], b = uq[ixj];
                if ((a > b) == up) { uq[idx] = b; uq[ixj] = a; }
            }
        }
    }
    doda_sync();
#ifdef DODA_TB_DEBUG
    {
        __shared__ int viol;
        if (tid == 0) viol = 0;
        doda_sync();
        for (int k = tid; k + 1 < U; k += 256)
            if (!(uq[k] < uq[k + 1])) atomicAdd(&viol, 1);
        doda_sync();
        if (viol > 0) {
            __shared__ unsigned first_hit;
            if (tid == 0) first_hit = atomicAdd(&g_tb_dbg[0], 1u);
            doda_sync();
            if (first_hit == 0) {
                if (tid == 0) { g_tb_dbg[1] = (unsigned)tile; g_tb_dbg[2] = (unsigned)U; g_tb_dbg[3] = (unsigned)P; g_tb_dbg[4] = (unsigned)viol; }
                if (tid < 4) { g_tb_dbg[8 + 4096 - 8 + tid] = (unsigned)wP[tid]; g_tb_dbg[8 + 4096 - 4 + tid] = (unsigned)wStages[tid]; }
                for (int k = tid; k < 2048; k += 256) { g_tb_dbg[8 + k] = uq0[k]; g_tb_dbg[8 + 2048 + k] = uq[k]; }
            }
        }
        doda_sync();
    }
#endif
    for (int k = tid; k < TB_UMAX; k += 256) ul[tb_upos(k)] = k < U ? (int32_t)uq[k] : -1;

    // ---- 3. local indices ----
    // position of every key: the sorted list writes each key's rank next to its hash slot, then every
    // table entry is one hash probe (~1.1 LDS reads) + one rank read.  (A binary search per entry — 270 LDS
    // reads per thread — made this phase as expensive as the sort: the builder is LDS-throughput bound.)
    for (int k = tid; k < U; k += 256) {
        const unsigned key = uq[k];
        unsigned slot = hash_mix(key) & (HCAP - 1);
        while (htab[slot] != key) slot = (slot + 1) & (HCAP - 1);
        hrank[slot] = (unsigned short)k;
    }
    doda_sync();
    uint32_t *li = v.lidx + (size_t)tile * (TB_T * TB_LW) + tb_lpos(tid);
#pragma unroll
    for (int w3 = 0; w3 < TB_LW; ++w3) {
        unsigned word = 0u;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int oo = 2 * (3 * (w3 % 5) + q) + w3 / 5, o = oo < TB_K ? oo : 0;
            unsigned r = 0u;   // absent (or no such offset): LDS slot 0, the zero row
            if (oo < TB_K && e[o] >= 0) {
                const unsigned key = (unsigned)e[o];
                unsigned slot = hash_mix(key) & (HCAP - 1);
                while (htab[slot] != key) slot = (slot + 1) & (HCAP - 1);
                r = (unsigned)(hrank[slot] + 1);
            }
            word |= r << (10 * q);
        }
        li[w3 * TB_T] = word;
    }
}

}  // namespace

#ifdef DODA_TB_DEBUG
extern "C" int doda_tilebook_debug(unsigned *host_buf /*[8 + 4096]*/, int reset) {
    if (hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_tb_dbg), sizeof(unsigned) * (8 + 4096)) != hipSuccess) return -1;
    if (reset) { unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_tb_dbg), &z, 4); }
    return 0;
}
#endif
extern "C" int32_t doda_tilebook_tile(void) { return TB_T; }
extern "C" int32_t doda_tilebook_umax(void) { return TB_UMAX; }

extern "C" size_t doda_tilebook_bytes(int32_t n_rows, int32_t K) {
    if (n_rows <= 0 || K != TB_K) return 0;
    return tilebook_bytes_for(n_rows);
}

extern "C" int doda_tilebook_build(const int32_t *tbl, int32_t ld, int32_t K, int32_t n_rows, void *tilebook,
                                   size_t tilebook_bytes, doda_stream_t stream) {
    if (K != TB_K) return DODA_ERR_UNSUPPORTED;
    if (n_rows < 0 || ld < n_rows) return DODA_ERR_INVALID;
    if (n_rows == 0) return DODA_OK;
    if (!tbl || !tilebook || ((uintptr_t)tilebook & 15)) return DODA_ERR_INVALID;
    if (tilebook_bytes < tilebook_bytes_for(n_rows)) return DODA_ERR_WORKSPACE;
    const TileBookView v = tilebook_view(tilebook, n_rows);
    if (hipMemsetAsync(v.n_over, 0, 8, as_stream(stream)) != hipSuccess) return DODA_ERR_LAUNCH;
    hipLaunchKernelGGL(tilebook_build, dim3(v.nt), dim3(256), 0, as_stream(stream), tbl, (int)ld, (int)n_rows, v);
    return doda_check_launch();
}
