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
#include "tilebook.hpp"

namespace {

constexpr int HCAP = 4096;   // hash slots: the insert loop stops adding once TB_UMAX + 256 keys are in
constexpr unsigned EMPTY = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void tilebook_build(const int32_t *__restrict__ tbl, int ld, int n,
                                                      TileBookView v) {
    __shared__ unsigned htab[HCAP];
    constexpr int UQ = 2048;          // sort buffer: the power of two above TB_UMAX
    __shared__ unsigned uq[UQ];
    static_assert(TB_UMAX <= UQ, "sort buffer");
    __shared__ unsigned short hrank[HCAP];
    __shared__ int cnt;
    __shared__ int wsum[4];
    const int tile = blockIdx.x, tid = threadIdx.x, t0 = tile * TB_T;
    for (int k = tid; k < HCAP; k += 256) htab[k] = EMPTY;
    if (tid == 0) cnt = 0;
    __syncthreads();

    int e[TB_K];
    const bool rok = t0 + tid < n;
#pragma unroll
    for (int o = 0; o < TB_K; ++o) e[o] = rok ? tbl[(long long)o * ld + t0 + tid] : -1;

    // ---- 1. distinct rows ----  (loops over o stay unrolled: e[] must live in registers)
#pragma unroll
    for (int o = 0; o < TB_K; ++o) {
        if (e[o] < 0) continue;
        if (*(volatile int *)&cnt > TB_UMAX) break;   // overflow: the count is all that is kept
        const unsigned key = (unsigned)e[o];
        unsigned slot = hash_mix(key) & (HCAP - 1);
        for (;;) {
            const unsigned old = atomicCAS(&htab[slot], EMPTY, key);
            if (old == EMPTY) { atomicAdd(&cnt, 1); break; }
            if (old == key) break;
            slot = (slot + 1) & (HCAP - 1);
        }
    }
    __syncthreads();
    const int U = cnt;
    if (tid == 0) {
        v.ucount[tile] = U;
        if (U > TB_CAP64) atomicAdd(&v.n_over[0], 1);
        if (U > TB_UMAX) atomicAdd(&v.n_over[1], 1);
    }
    int32_t *ul = v.ulist + (size_t)tile * TB_UMAX;
    if (U > TB_UMAX) {
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
        __syncthreads();
        int base = incl - c;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
#pragma unroll
        for (int k = 0; k < HCAP / 256; ++k)
            if (mine[k] != EMPTY) uq[base++] = mine[k];
        __syncthreads();
    }
    int P = 2;                    // sort only the power of two that holds the keys (EMPTY pads sort last)
    while (P < U) P <<= 1;
    // compare-exchange p of a stage pairs idx = insert-zero-bit(p, j) with idx | j.  Thread tid takes
    // p = tid + 256 q, so for j <= 64 both elements lie in the 128-element chunk its own wave handles and
    // LDS operations of one wave execute in order: only stages with j >= 128 (and the first j <= 64 stage
    // after one) need the workgroup barrier — 12 barriers instead of 55 for P = 1024.  (P = 2048: a thread's four
    // pairs p = tid + 256 q still stay inside its wave's chunks for j <= 64.)
    bool crossed = true;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool cross = j >= 128;
            if (cross || crossed) __syncthreads();
            crossed = cross;
            for (int p = tid; p < (P >> 1); p += 256) {
                const int idx = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = idx | j;
                const bool up = (idx & k) == 0;
                const unsigned a = uq[idx], b = uq[ixj];
                if ((a > b) == up) { uq[idx] = b; uq[ixj] = a; }
            }
        }
    }
    __syncthreads();
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
    __syncthreads();
    uint16_t *li = v.lidx + (size_t)tile * TB_K * TB_T + tb_pos(tid);
#pragma unroll
    for (int o = 0; o < TB_K; ++o) {
        unsigned short r = 0;   // absent: LDS slot 0, the zero row
        if (e[o] >= 0) {
            const unsigned key = (unsigned)e[o];
            unsigned slot = hash_mix(key) & (HCAP - 1);
            while (htab[slot] != key) slot = (slot + 1) & (HCAP - 1);
            r = (unsigned short)(hrank[slot] + 1);
        }
        li[o * TB_T] = r;
    }
}

}  // namespace

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
