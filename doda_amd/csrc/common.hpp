// Shared device/host helpers for libdoda_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/doda_hip.h"

#define DODA_WAVE 64

static inline int doda_check_launch() {
    return hipGetLastError() == hipSuccess ? DODA_OK : DODA_ERR_LAUNCH;
}

static inline hipStream_t as_stream(doda_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline uint32_t next_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Cell-id hash table.  One 64-bit word per slot: (cell id : 64 - vbits | value : vbits).  A single
// atomicCAS inserts; because the key sits in the high part, atomicMin on the whole word keeps the
// smallest value among inserts of the same key (first-touch numbering needs exactly that).
// The split is chosen per build (HashFmt): grids below 2^32 cells keep 32 value bits; larger grids
// (1 cm voxels: batch x 2000^3 cells, SURVEY §7 "64-bit packed keys") take key bits from the value,
// which only has to hold a row number — e.g. 2^40 cells with 2^24 rows.
// Capacity is a power of two >= 2 * n (load factor <= 0.5), linear probing.
// ---------------------------------------------------------------------------------------------
#define DODA_HASH_EMPTY 0xFFFFFFFFFFFFFFFFull

typedef unsigned long long cellkey_t;

struct HashFmt {
    int vbits;                  // low bits holding the value
    unsigned long long vmask;   // (1 << vbits) - 1
};

// key / value split for a grid of `cells` cells and values 0 .. n_values-1; false if 64 bits cannot hold both
static inline bool make_hash_fmt(long double cells, unsigned long long n_values, HashFmt *f) {
    int kbits = 1;
    while (kbits < 63 && (long double)(1ull << kbits) < cells + 2.0L) ++kbits;   // all-ones key stays free
    if ((long double)(1ull << kbits) < cells + 2.0L) return false;
    int vb = 64 - kbits;
    if (vb > 32) vb = 32;
    if (n_values > (1ull << vb)) return false;
    f->vbits = vb;
    f->vmask = (1ull << vb) - 1ull;
    return true;
}

__device__ __forceinline__ uint32_t hash_mix(uint32_t k) {
    k ^= k >> 16;
    k *= 0x7feb352du;
    k ^= k >> 15;
    k *= 0x846ca68bu;
    k ^= k >> 16;
    return k;
}

// (A home slot that keeps the eight cells of a z-run in one 64-byte line — mix(k >> 3) * 8 + (k & 7) — was
// measured: the occupied runs cluster, absent neighbours walk through them, and the level-1 probe kernel
// went from 117 to 280 us.)
__device__ __forceinline__ uint32_t hash_key(cellkey_t k) {
    return hash_mix((uint32_t)k ^ ((uint32_t)(k >> 32) * 0x9e3779b1u));
}

// insert (key, val); duplicates of key keep the minimum val.
__device__ __forceinline__ void hash_insert_min(unsigned long long *tab, uint32_t mask, HashFmt f,
                                                cellkey_t key, uint32_t val) {
    const unsigned long long mine = (key << f.vbits) | val;
    uint32_t slot = hash_key(key) & mask;
    for (;;) {
        unsigned long long old = atomicCAS(&tab[slot], DODA_HASH_EMPTY, mine);
        if (old == DODA_HASH_EMPTY) return;
        if ((old >> f.vbits) == key) {
            if (mine < old) atomicMin(&tab[slot], mine);
            return;
        }
        slot = (slot + 1) & mask;
    }
}

// returns the value stored for key, or -1.
__device__ __forceinline__ int hash_find(const unsigned long long *__restrict__ tab,
                                         uint32_t mask, HashFmt f, cellkey_t key) {
    uint32_t slot = hash_key(key) & mask;
    for (;;) {
        unsigned long long cur = tab[slot];
        if (cur == DODA_HASH_EMPTY) return -1;
        if ((cur >> f.vbits) == key) return (int)(uint32_t)(cur & f.vmask);
        slot = (slot + 1) & mask;
    }
}

// ---------------------------------------------------------------------------------------------
// Wave / block primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Workgroup barrier with the LDS wait spelled out.  hipcc's wait-count pass can lose LDS stores that sit in a conditionally executed
// block of a loop whose header is the barrier: it then emits a bare s_barrier, a wave arrives with its ds_write still in flight and
// the others read the old contents (round 5: the tilebook builder's sort, ~1 wrong list in 50 000).  Every barrier of
// this library goes through here (the hand-written `s_waitcnt lgkmcnt(0); s_barrier` pairs of the LDS-DMA kernel aside); the wait is
// free when nothing is pending: conv_tile16 and the bench step measured the same with and without it.
__device__ __forceinline__ void doda_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// number of set bits of a 64-bit ballot mask strictly below this lane
__device__ __forceinline__ int mask_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// inclusive wave scan (sum) over 64 lanes
__device__ __forceinline__ int wave_inclusive_sum(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += n;
    }
    return v;
}

// XCD-aware work mapping (guide T1).  Blocks are dealt round-robin to the 8 XCDs (block b runs on
// XCD b % 8, each with a private 4 MB L2); give every XCD one CONTIGUOUS range of work items so
// that the rows its blocks gather stay in its own L2.  Bijective for any grid size; a different
// hardware placement only costs speed.
__device__ __forceinline__ int xcd_work_item(int bid, int n_items) {
    const int q = n_items >> 3, r = n_items & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Exclusive prefix sum over int32 arrays of any length: three launches, deterministic.
// ws_ints: scratch of at least scan_ws_ints(n) ints.  Optionally writes the grand total.
size_t scan_ws_ints(int n);
int exclusive_scan_i32(const int32_t *in, int32_t *out, int n, int32_t *total_out, int32_t *ws,
                       hipStream_t stream);
