// Rulebook ("gather table") construction for SubMConv3d and SparseConv3d(k2,s2), plus the
// export to spconv-v1.2-format indice pairs.
//
// Replaces spconv v1.2 `get_indice_pairs` (third-party, un-vendored; reference call sites
// model/unet.py:36, model/unet_block.py:26,29,48,70,78).  Upstream builds a dense
// batch*X*Y*Z int32 grid per call (1 GB at 1 cm); here the COO voxel list is
// hashed into a 2x-overprovisioned open-addressing table of 64-bit (cell:32 | row:32) words —
// 16 B per voxel of table, L2-resident at ScanNet sizes — and every ordering that spconv's
// serial CPU loop produces (ascending-input pair lists, first-touch output numbering) is
// obtained from min-reductions and stable prefix sums, never from atomic cursors, so the
// result is deterministic and bit-identical to the serial order.
//
// All kernels are HBM/L2-latency-bound integer work: one lane per voxel, coalesced reads of the
// int4 coordinate rows, coalesced table writes (tbl[o][t] with t across lanes).
#include "common.hpp"
#include <stdlib.h>

namespace {

struct GridDesc {
    int X, Y, Z;  // spatial shape the cell ids are computed in
};

__device__ __forceinline__ cellkey_t cell_id(int b, int x, int y, int z, const GridDesc g) {
    return (cellkey_t)((((long long)b * g.X + x) * g.Y + y) * g.Z + z);
}

// A row whose batch index or coordinates lie outside the declared grid is never entered into the cell map and sees no
// neighbours: its table row is the centre only (SubM) or parent -1 (Down2).  The hash and the direct-address builders
// apply the same test, so which one the workspace size selects does not change the table.
__device__ __forceinline__ bool in_grid(const int4 c, int batch, const GridDesc g) {
    return (unsigned)c.x < (unsigned)batch && (unsigned)c.y < (unsigned)g.X && (unsigned)c.z < (unsigned)g.Y && (unsigned)c.w < (unsigned)g.Z;
}

// ---- SubM ---------------------------------------------------------------------------------
// Also pre-fills column t of the table's mirrored half (offsets first_fill .. K3-1) with -1: the
// probe kernel only writes hits there (saves a memset launch per rulebook).
__global__ __launch_bounds__(256) void subm_insert(const int4 *__restrict__ indices, int m, int batch,
                                                   GridDesc g, unsigned long long *tab,
                                                   uint32_t mask, HashFmt hf, int32_t *__restrict__ nbr, int ld,
                                                   int first_fill, int k3) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int4 c = indices[t];
    if (in_grid(c, batch, g)) hash_insert_min(tab, mask, hf, cell_id(c.x, c.y, c.z, c.w, g), (uint32_t)t);
    for (int o = first_fill; o < k3; ++o) nbr[(long long)o * ld + t] = -1;
}

// One thread per voxel.  A SubM table is its own transpose under offset mirroring (nbr[o][t] = j  <=>
// nbr[K3-1-o][j] = t), so only the first half of the offsets is probed: a hit writes both entries
// (the mirrored one is a scattered 4-byte store into a half that the launcher pre-fills with -1).
// The kernel is bound by the rate of random 8-byte table reads (16 M of them at 600k voxels), not
// by their latency — issuing all first-slot reads before inspecting any changed nothing — so
// halving the reads is what pays.  The first-slot reads are still issued together; only a
// neighbour whose first slot holds a different key walks the probe chain.
template <int KS>
__global__ __launch_bounds__(256) void subm_probe(const int4 *__restrict__ indices, int m, int batch,
                                                  GridDesc g,
                                                  const unsigned long long *__restrict__ tab,
                                                  uint32_t mask, HashFmt hf, int32_t *__restrict__ nbr,
                                                  int ld) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int4 c = indices[t];  // (b, x, y, z)
    constexpr int R = KS / 2, K3 = KS * KS * KS, NP = K3 / 2;   // offsets 0..NP-1 are probed
    const bool own_ok = in_grid(c, batch, g);
    const cellkey_t own = own_ok ? cell_id(c.x, c.y, c.z, c.w, g) : (cellkey_t)0;
    nbr[(long long)NP * ld + t] = t;   // centre
    if (NP == 0) return;
    cellkey_t key[NP > 0 ? NP : 1];
    unsigned long long first[NP > 0 ? NP : 1];
#pragma unroll
    for (int o = 0; o < NP; ++o) {
        const int k0 = o / (KS * KS), k1 = (o / KS) % KS, k2 = o % KS;
        const int x = c.y + k0 - R, y = c.z + k1 - R, z = c.w + k2 - R;
        const bool inb = own_ok && x >= 0 && x < g.X && y >= 0 && y < g.Y && z >= 0 && z < g.Z;
        // out-of-grid neighbours read the voxel's own slot chain head (a valid address) and are
        // discarded below through key == own
        key[o] = inb ? cell_id(c.x, x, y, z, g) : own;
        first[o] = tab[hash_key(key[o]) & mask];
    }
#pragma unroll
    for (int o = 0; o < NP; ++o) {
        int v = -1;
        if (key[o] != own) {
            const unsigned long long cur = first[o];
            if ((cur >> hf.vbits) == key[o]) v = (int)(uint32_t)(cur & hf.vmask);
            else if (cur != DODA_HASH_EMPTY) {   // collision: continue along the chain
                uint32_t slot = (hash_key(key[o]) + 1) & mask;
                for (;;) {
                    const unsigned long long nx = tab[slot];
                    if (nx == DODA_HASH_EMPTY) break;
                    if ((nx >> hf.vbits) == key[o]) { v = (int)(uint32_t)(nx & hf.vmask); break; }
                    slot = (slot + 1) & mask;
                }
            }
        }
        nbr[(long long)o * ld + t] = v;
        if (v >= 0) nbr[(long long)(K3 - 1 - o) * ld + v] = t;
    }
}

// ---- direct-address grid (round 4) ---------------------------------------------------------------
// For grids of up to DODA_RULEBOOK_GRID_MAX_CELLS cells (a batch of 2 cm ScanNet scenes: 16.5 M cells = 66 MB of int32)
// the cell -> row map is a plain array: grid[cell] = row or -1.  A probe is ONE 4-byte read instead of 1.3 eight-byte
// hash reads, and the three z-neighbours of a (dx, dy) pair are adjacent words: the 13 probes of a voxel touch ~5
// 64-byte sectors instead of ~17 — the hash probe ran at the fabric's random-sector rate.  The caller
// opts in by handing over a workspace with room for the grid behind the hash workspace (doda_hip.h); larger grids
// (1 cm scenes: 2^34 cells) keep the hash.  Same results: first-touch (lowest row) wins a cell, as hash_insert_min.
// (round 5: 2^28 cells = 1 GiB of grid — a batch of four 1 cm scenes has 8.5e7 cells; its level-1 probe took 1.1 ms on the
// hash and the memset of the grid costs 45 us.  DODA_RULEBOOK_GRID_MAX_LOG2 moves the limit; callers size the workspace.)
static const long long GRID_MAX_CELLS = [] {
    const char *e = getenv("DODA_RULEBOOK_GRID_MAX_LOG2");
    const int b = e ? atoi(e) : 28;
    return 1ll << (b < 0 ? 0 : b > 30 ? 30 : b);
}();

__global__ __launch_bounds__(256) void subm_grid_insert(const int4 *__restrict__ indices, int m, int batch, GridDesc g,
                                                        int32_t *__restrict__ grid, int32_t *__restrict__ nbr, int ld,
                                                        int first_fill, int k3) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int4 c = indices[t];
    if (in_grid(c, batch, g)) atomicMin((unsigned *)grid + (long long)cell_id(c.x, c.y, c.z, c.w, g), (unsigned)t);
    for (int o = first_fill; o < k3; ++o) nbr[(long long)o * ld + t] = -1;
}

template <int KS>
__global__ __launch_bounds__(256) void subm_grid_probe(const int4 *__restrict__ indices, int m, int batch, GridDesc g,
                                                       const int32_t *__restrict__ grid, int32_t *__restrict__ nbr, int ld) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int4 c = indices[t];  // (b, x, y, z)
    constexpr int R = KS / 2, K3 = KS * KS * KS, NP = K3 / 2;
    nbr[(long long)NP * ld + t] = t;   // centre
    const bool own_ok = in_grid(c, batch, g);
    int v[NP];
#pragma unroll
    for (int o = 0; o < NP; ++o) {
        const int k0 = o / (KS * KS), k1 = (o / KS) % KS, k2 = o % KS;
        const int x = c.y + k0 - R, y = c.z + k1 - R, z = c.w + k2 - R;
        const bool inb = own_ok && x >= 0 && x < g.X && y >= 0 && y < g.Y && z >= 0 && z < g.Z;
        v[o] = inb ? grid[(long long)cell_id(c.x, x, y, z, g)] : -1;
    }
#pragma unroll
    for (int o = 0; o < NP; ++o) {
        nbr[(long long)o * ld + t] = v[o];
        if (v[o] >= 0) nbr[(long long)(K3 - 1 - o) * ld + v[o]] = t;
    }
}

__global__ __launch_bounds__(256) void down2_grid_insert(const int4 *__restrict__ indices, int m, int batch, GridDesc go,
                                                         int32_t *__restrict__ grid, int32_t *__restrict__ off) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    off[j] = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
    const int4 q = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
    if (c.y < 0 || c.z < 0 || c.w < 0 || !in_grid(q, batch, go)) return;
    atomicMin((unsigned *)grid + (long long)cell_id(q.x, q.y, q.z, q.w, go), (unsigned)j);
}

__global__ __launch_bounds__(256) void down2_grid_first(const int4 *__restrict__ indices, int m, int batch, GridDesc go,
                                                        const int32_t *__restrict__ grid, int32_t *__restrict__ firstj,
                                                        int32_t *__restrict__ flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    const int4 q = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
    int f = -1;
    if (!(c.y < 0 || c.z < 0 || c.w < 0) && in_grid(q, batch, go)) f = grid[(long long)cell_id(q.x, q.y, q.z, q.w, go)];
    firstj[j] = f;
    flag[j] = (f == j) ? 1 : 0;
}

// ---- Down2 (kernel 2, stride 2, pad 0) ------------------------------------------------------
__global__ __launch_bounds__(256) void down2_insert(const int4 *__restrict__ indices, int m, int batch,
                                                    GridDesc go, unsigned long long *tab,
                                                    uint32_t mask, HashFmt hf, int32_t *__restrict__ off) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    off[j] = ((c.y & 1) * 2 + (c.z & 1)) * 2 + (c.w & 1);
    const int qx = c.y >> 1, qy = c.z >> 1, qz = c.w >> 1;
    if ((unsigned)c.x >= (unsigned)batch || c.y < 0 || c.z < 0 || c.w < 0 || qx >= go.X || qy >= go.Y || qz >= go.Z) return;
    hash_insert_min(tab, mask, hf, cell_id(c.x, qx, qy, qz, go), (uint32_t)j);
}

__global__ __launch_bounds__(256) void down2_first(const int4 *__restrict__ indices, int m, int batch,
                                                   GridDesc go,
                                                   const unsigned long long *__restrict__ tab,
                                                   uint32_t mask, HashFmt hf, int32_t *__restrict__ firstj,
                                                   int32_t *__restrict__ flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    const int qx = c.y >> 1, qy = c.z >> 1, qz = c.w >> 1;
    int f = -1;
    if (!((unsigned)c.x >= (unsigned)batch || c.y < 0 || c.z < 0 || c.w < 0 || qx >= go.X || qy >= go.Y || qz >= go.Z))
        f = hash_find(tab, mask, hf, cell_id(c.x, qx, qy, qz, go));
    firstj[j] = f;
    flag[j] = (f == j) ? 1 : 0;
}

__global__ __launch_bounds__(256) void down2_finalize(const int4 *__restrict__ indices, int m,
                                                      const int32_t *__restrict__ firstj,
                                                      const int32_t *__restrict__ rank,
                                                      int32_t *__restrict__ parent,
                                                      int4 *__restrict__ out_indices) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int f = firstj[j];
    int q = -1;
    if (f >= 0) {
        q = rank[f];
        if (f == j) {
            const int4 c = indices[j];
            out_indices[q] = make_int4(c.x, c.y >> 1, c.z >> 1, c.w >> 1);
        }
    }
    parent[j] = q;
}

__global__ __launch_bounds__(256) void down2_tables(const int32_t *__restrict__ parent,
                                                    const int32_t *__restrict__ off, int m,
                                                    int32_t *__restrict__ child, int ld_out,
                                                    int32_t *__restrict__ par_off, int ld_in) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int q = parent[j], o = off[j];
#pragma unroll
    for (int k = 0; k < 8; ++k) par_off[(long long)k * ld_in + j] = (k == o) ? q : -1;
    if (q >= 0) child[(long long)o * ld_out + q] = j;
}

// ---- spconv-format pair export ------------------------------------------------------------------
constexpr int PAIR_TILE = 256;

__global__ __launch_bounds__(PAIR_TILE) void pairs_count(const int32_t *__restrict__ tbl, int ld,
                                                         int K, int n_rows, int flip,
                                                         int32_t *__restrict__ counts, int nt) {
    __shared__ int lds[PAIR_TILE / 64];
    const int o = blockIdx.y, src = flip ? K - 1 - o : o;
    const int j = blockIdx.x * PAIR_TILE + threadIdx.x;
    const bool valid = j < n_rows && tbl[(long long)src * ld + j] >= 0;
    const unsigned long long b = __ballot(valid);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = __popcll(b);
    doda_sync();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < PAIR_TILE / 64; ++w) s += lds[w];
        counts[(long long)o * nt + blockIdx.x] = s;
    }
}

// one block per offset: exclusive scan of its nt tile counts, total to pair_num[o]
__global__ __launch_bounds__(256) void pairs_scan(int32_t *counts, int nt, int32_t *pair_num) {
    __shared__ int lds[4];
    int32_t *row = counts + (long long)blockIdx.x * nt;
    int carry = 0;
    for (int start = 0; start < nt; start += 256) {
        const int i = start + threadIdx.x;
        const int v = i < nt ? row[i] : 0;
        const int inc = wave_inclusive_sum(v);
        if (lane_id() == 63) lds[threadIdx.x >> 6] = inc;
        doda_sync();
        int base = 0, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < (int)(threadIdx.x >> 6)) base += lds[w];
            tot += lds[w];
        }
        doda_sync();
        if (i < nt) row[i] = carry + base + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) pair_num[blockIdx.x] = carry;
}

__global__ __launch_bounds__(PAIR_TILE) void pairs_fill(const int32_t *__restrict__ tbl, int ld,
                                                        int K, int n_rows, int flip,
                                                        const int32_t *__restrict__ counts, int nt,
                                                        int32_t *__restrict__ pairs,
                                                        int ld_pairs) {
    __shared__ int lds[PAIR_TILE / 64];
    const int o = blockIdx.y, src = flip ? K - 1 - o : o;
    const int j = blockIdx.x * PAIR_TILE + threadIdx.x;
    const int v = j < n_rows ? tbl[(long long)src * ld + j] : -1;
    const bool valid = v >= 0;
    const unsigned long long b = __ballot(valid);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = __popcll(b);
    doda_sync();
    int base = counts[(long long)o * nt + blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += lds[w];
    if (valid) {
        const int pos = base + mask_rank(b);
        pairs[(long long)o * ld_pairs + pos] = j;                          // [0][o][pos] = in
        pairs[((long long)K + o) * ld_pairs + pos] = v;                    // [1][o][pos] = out
    }
}

// ---- generic geometry (any kernel_size / stride / padding / dilation with K <= 27) -----------
// spconv's getIndicePairsConv numbers the outputs in first-touch order of a serial scan: inputs
// ascending, each input's valid output positions in getValidOutPos order (last axis fastest, from
// the upper bound downwards).  Here every (input j, enumeration rank i) candidate carries the touch
// key j*K + i; a hash of the OUTPUT cells keeps the minimum key per cell (atomicMin on the packed
// word), the entries that hold their cell's minimum are flagged and a prefix sum over the flags in
// key order is the serial numbering.  No atomic cursor anywhere: deterministic and equal to the
// serial order.
struct ConvGeo {
    int k[3], s[3], p[3], d[3];   // kernel, stride, padding, dilation
    int K;
    GridDesc out;                 // output spatial shape
};

// enumeration of getValidOutPos; calls f(rank, ox, oy, oz, offset) for the VALID positions
template <class F>
__device__ __forceinline__ void for_valid_out(const int4 c, const ConvGeo &g, F f) {
    const int pos[3] = {c.y, c.z, c.w};
    int lo[3], up[3], cs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = (pos[a] - (g.k[a] - 1) * g.d[a] - 1 + g.s[a] + g.p[a]) / g.s[a];
        up[a] = (pos[a] + g.p[a]) / g.s[a];
        cs[a] = (up[a] - lo[a]) / g.d[a] + 1;
    }
    const int shape[3] = {g.out.X, g.out.Y, g.out.Z};
    int rank = 0;
    for (int c0 = 0; c0 < cs[0]; ++c0)
        for (int c1 = 0; c1 < cs[1]; ++c1)
            for (int c2 = 0; c2 < cs[2]; ++c2) {
                const int cnt[3] = {c0, c1, c2};
                int v[3], off = 0, mul = 1;
                bool valid = true;
#pragma unroll
                for (int a = 2; a >= 0; --a) {
                    v[a] = up[a] - cnt[a] * g.d[a];
                    valid &= v[a] >= 0 && v[a] <= shape[a] - 1;
                    off += mul * (pos[a] - v[a] * g.s[a] + g.p[a]) / g.d[a];
                    mul *= g.k[a];
                }
                if (valid) {
                    f(rank, v[0], v[1], v[2], off);
                    ++rank;
                }
            }
}

__global__ __launch_bounds__(256) void conv_touch(const int4 *__restrict__ indices, int m, ConvGeo g,
                                                  unsigned long long *tab, uint32_t mask, HashFmt hf) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    for_valid_out(c, g, [&](int rank, int ox, int oy, int oz, int) {
        hash_insert_min(tab, mask, hf, cell_id(c.x, ox, oy, oz, g.out), (uint32_t)j * (uint32_t)g.K + (uint32_t)rank);
    });
}

// flag[j*K + rank] = 1 where that candidate is the first touch of its output cell
__global__ __launch_bounds__(256) void conv_first(const int4 *__restrict__ indices, int m, ConvGeo g,
                                                  const unsigned long long *__restrict__ tab,
                                                  uint32_t mask, HashFmt hf, int32_t *__restrict__ flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    for (int r = 0; r < g.K; ++r) flag[(long long)j * g.K + r] = 0;
    for_valid_out(c, g, [&](int rank, int ox, int oy, int oz, int) {
        const uint32_t mine = (uint32_t)j * (uint32_t)g.K + (uint32_t)rank;
        if ((uint32_t)hash_find(tab, mask, hf, cell_id(c.x, ox, oy, oz, g.out)) == mine) flag[mine] = 1;
    });
}

// the first-touch candidates write their output's coordinates and replace the table value (touch
// key) of their cell by the output id
__global__ __launch_bounds__(256) void conv_number(const int4 *__restrict__ indices, int m, ConvGeo g,
                                                   unsigned long long *tab, uint32_t mask, HashFmt hf,
                                                   const int32_t *__restrict__ flag,
                                                   const int32_t *__restrict__ rank_of,
                                                   int4 *__restrict__ out_indices) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    for_valid_out(c, g, [&](int rank, int ox, int oy, int oz, int) {
        const long long e = (long long)j * g.K + rank;
        if (!flag[e]) return;
        const int id = rank_of[e];
        out_indices[id] = make_int4(c.x, ox, oy, oz);
        const cellkey_t key = cell_id(c.x, ox, oy, oz, g.out);
        uint32_t slot = hash_key(key) & mask;
        while ((tab[slot] >> hf.vbits) != key) slot = (slot + 1) & mask;   // present by construction
        tab[slot] = (key << hf.vbits) | (uint32_t)id;
    });
}

// tbl[off][out] = j and tbl_rev[off][j] = out for every (input, output, offset) triple
__global__ __launch_bounds__(256) void conv_tables(const int4 *__restrict__ indices, int m, ConvGeo g,
                                                   const unsigned long long *__restrict__ tab,
                                                   uint32_t mask, HashFmt hf, int32_t *__restrict__ tbl, int ld_out,
                                                   int32_t *__restrict__ tbl_rev, int ld_in) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int4 c = indices[j];
    for (int o = 0; o < g.K; ++o) tbl_rev[(long long)o * ld_in + j] = -1;
    for_valid_out(c, g, [&](int, int ox, int oy, int oz, int off) {
        const int out = hash_find(tab, mask, hf, cell_id(c.x, ox, oy, oz, g.out));
        tbl_rev[(long long)off * ld_in + j] = out;
        tbl[(long long)off * ld_out + out] = j;
    });
}

// SubM with a non-cubic odd kernel: plain per-offset lookups (offset = row-major kernel index)
__global__ __launch_bounds__(256) void subm_probe_generic(const int4 *__restrict__ indices, int m, int batch,
                                                          GridDesc g, int k0, int k1, int k2,
                                                          const unsigned long long *__restrict__ tab,
                                                          uint32_t mask, HashFmt hf, int32_t *__restrict__ nbr, int ld) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int4 c = indices[t];
    const bool own_ok = in_grid(c, batch, g);
    int o = 0;
    for (int a = 0; a < k0; ++a)
        for (int b = 0; b < k1; ++b)
            for (int d = 0; d < k2; ++d, ++o) {
                const int x = c.y + a - k0 / 2, y = c.z + b - k1 / 2, z = c.w + d - k2 / 2;
                int v = -1;
                if (own_ok && x >= 0 && x < g.X && y >= 0 && y < g.Y && z >= 0 && z < g.Z)
                    v = hash_find(tab, mask, hf, cell_id(c.x, x, y, z, g));
                if (!own_ok && a == k0 / 2 && b == k1 / 2 && d == k2 / 2) v = t;   // centre only
                nbr[(long long)o * ld + t] = v;
            }
}

struct RbWs {
    unsigned long long *tab;
    uint32_t cap;
    int32_t *a, *b, *c, *scan;
    size_t total;
};

RbWs carve(void *ws, int m) {
    RbWs r;
    const uint32_t cap = next_pow2((uint32_t)(2 * (m > 0 ? m : 1) < 1024 ? 1024 : 2 * m));
    size_t off = 0;
    char *p = (char *)ws;
    r.cap = cap;
    r.tab = (unsigned long long *)(p + off);
    off += (size_t)cap * 8;
    const size_t mi = align_up((size_t)(m > 0 ? m : 1) * 4, 256);
    r.a = (int32_t *)(p + off);
    off += mi;
    r.b = (int32_t *)(p + off);
    off += mi;
    r.c = (int32_t *)(p + off);
    off += mi;
    r.scan = (int32_t *)(p + off);
    off += align_up(scan_ws_ints(m) * 4, 256);
    r.total = off;
    return r;
}

// key / value split of the hash words for this grid and value range (values are row numbers, or
// row * K + rank in the generic builder); false = 64 bits cannot hold both (DODA_ERR_GRID_TOO_LARGE)
bool grid_fmt(int batch, int X, int Y, int Z, unsigned long long max_value, HashFmt *hf) {
    if (batch <= 0 || X <= 0 || Y <= 0 || Z <= 0) return false;
    return make_hash_fmt((long double)batch * X * Y * Z, max_value, hf);
}
}  // namespace

// the direct-address grid behind the hash workspace, or null (no room / grid too large / switched off)
static int32_t *dense_grid(void *ws, size_t ws_bytes, size_t hash_total, int batch, const GridDesc g) {
    static const bool off = getenv("DODA_RULEBOOK_GRID") && getenv("DODA_RULEBOOK_GRID")[0] == '0';
    if (off || batch <= 0 || g.X <= 0 || g.Y <= 0 || g.Z <= 0) return nullptr;
    const long double cells = (long double)batch * g.X * g.Y * g.Z;
    if (cells > (long double)GRID_MAX_CELLS) return nullptr;
    const size_t at = align_up(hash_total, 256), need = at + (size_t)cells * 4;
    return ws_bytes >= need ? (int32_t *)((char *)ws + at) : nullptr;
}

extern "C" size_t doda_rulebook_workspace_bytes(int32_t m) { return carve(nullptr, m).total; }

extern "C" int doda_rulebook_subm(const int32_t *indices, int32_t m, const int32_t *shape_h,
                                  int32_t batch, int32_t ksize, int32_t *nbr, int32_t ld,
                                  void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (m < 0 || !shape_h || ld < m) return DODA_ERR_INVALID;
    if (ksize != 1 && ksize != 3) return DODA_ERR_UNSUPPORTED;
    if (m == 0) return DODA_OK;
    if (!indices || !nbr || !ws) return DODA_ERR_INVALID;
    HashFmt hf;
    if (!grid_fmt(batch, shape_h[0], shape_h[1], shape_h[2], (unsigned long long)m, &hf)) return DODA_ERR_GRID_TOO_LARGE;
    const RbWs w = carve(ws, m);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const GridDesc g{shape_h[0], shape_h[1], shape_h[2]};
    const int grid = div_up(m, 256);
    if (ksize == 1) {
        // identity table
        hipLaunchKernelGGL((subm_probe<1>), dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch,
                           g, w.tab, w.cap - 1, hf, nbr, ld);
        return doda_check_launch();
    }
    int32_t *dgrid = dense_grid(ws, ws_bytes, w.total, batch, g);
    if (dgrid) {      // direct-address grid: the caller's workspace has room for it (and the grid is small enough)
        hipMemsetAsync(dgrid, 0xFF, (size_t)batch * g.X * g.Y * g.Z * 4, s);
        hipLaunchKernelGGL(subm_grid_insert, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g, dgrid,
                           nbr, ld, 14, 27);
        hipLaunchKernelGGL((subm_grid_probe<3>), dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g, dgrid,
                           nbr, ld);
        return doda_check_launch();
    }
    hipMemsetAsync(w.tab, 0xFF, (size_t)w.cap * 8, s);
    hipLaunchKernelGGL(subm_insert, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g,
                       w.tab, w.cap - 1, hf, nbr, ld, 14, 27);   // + mirrored half := -1
    hipLaunchKernelGGL((subm_probe<3>), dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g,
                       w.tab, w.cap - 1, hf, nbr, ld);
    return doda_check_launch();
}

extern "C" int doda_rulebook_down2_assign(const int32_t *indices, int32_t m,
                                          const int32_t *shape_h, int32_t batch, int32_t *parent,
                                          int32_t *off, int32_t *out_indices, int32_t *counts_out,
                                          void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (m < 0 || !shape_h || !counts_out) return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    if (m == 0) {
        hipMemsetAsync(counts_out, 0, sizeof(int32_t), s);
        return DODA_OK;
    }
    if (!indices || !parent || !off || !out_indices || !ws) return DODA_ERR_INVALID;
    // spconv get_conv_output_size: (in + 2p - d(k-1) - 1)/s + 1 with k=2,s=2,p=0,d=1
    GridDesc go;
    go.X = (shape_h[0] - 2) / 2 + 1;
    go.Y = (shape_h[1] - 2) / 2 + 1;
    go.Z = (shape_h[2] - 2) / 2 + 1;
    if (shape_h[0] < 2 || shape_h[1] < 2 || shape_h[2] < 2) return DODA_ERR_INVALID;
    HashFmt hf;
    if (!grid_fmt(batch, go.X, go.Y, go.Z, (unsigned long long)m, &hf)) return DODA_ERR_GRID_TOO_LARGE;
    const RbWs w = carve(ws, m);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    const int grid = div_up(m, 256);
    int32_t *dgrid = dense_grid(ws, ws_bytes, w.total, batch, go);
    if (dgrid) {
        hipMemsetAsync(dgrid, 0xFF, (size_t)batch * go.X * go.Y * go.Z * 4, s);
        hipLaunchKernelGGL(down2_grid_insert, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, go, dgrid, off);
        hipLaunchKernelGGL(down2_grid_first, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, go, dgrid,
                           w.a /*firstj*/, w.b /*flag*/);
    } else {
        hipMemsetAsync(w.tab, 0xFF, (size_t)w.cap * 8, s);
        hipLaunchKernelGGL(down2_insert, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, go,
                           w.tab, w.cap - 1, hf, off);
        hipLaunchKernelGGL(down2_first, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, go,
                           w.tab, w.cap - 1, hf, w.a /*firstj*/, w.b /*flag*/);
    }
    int st = exclusive_scan_i32(w.b, w.c /*rank*/, m, counts_out, w.scan, s);
    if (st != DODA_OK) return st;
    hipLaunchKernelGGL(down2_finalize, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, w.a,
                       w.c, parent, (int4 *)out_indices);
    return doda_check_launch();
}

extern "C" int doda_rulebook_down2_tables(const int32_t *parent, const int32_t *off, int32_t m,
                                          int32_t m_out, int32_t *child, int32_t ld_out,
                                          int32_t *par_off, int32_t ld_in, doda_stream_t stream) {
    if (m < 0 || m_out < 0 || ld_out < m_out || ld_in < m) return DODA_ERR_INVALID;
    if (m == 0) return DODA_OK;
    if (!parent || !off || !child || !par_off) return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    if (m_out > 0) hipMemsetAsync(child, 0xFF, (size_t)8 * ld_out * 4, s);
    hipLaunchKernelGGL(down2_tables, dim3(div_up(m, 256)), dim3(256), 0, s, parent, off, m, child,
                       ld_out, par_off, ld_in);
    return doda_check_launch();
}

extern "C" int32_t doda_rulebook_pairs_tile(void) { return PAIR_TILE; }

extern "C" size_t doda_rulebook_pairs_workspace_bytes(int32_t n_rows, int32_t K) {
    const int nt = div_up(n_rows > 0 ? n_rows : 1, PAIR_TILE);
    return align_up((size_t)K * nt * 4, 256);
}

extern "C" int doda_rulebook_pairs(const int32_t *tbl, int32_t ld, int32_t K, int32_t n_rows,
                                   int32_t flip, int32_t *pairs, int32_t ld_pairs,
                                   int32_t *pair_num, void *ws, size_t ws_bytes,
                                   doda_stream_t stream) {
    if (K <= 0 || n_rows < 0 || ld < n_rows || ld_pairs < n_rows || !pair_num)
        return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    if (n_rows == 0) {
        hipMemsetAsync(pair_num, 0, (size_t)K * 4, s);
        return DODA_OK;
    }
    if (!tbl || !pairs || !ws) return DODA_ERR_INVALID;
    if (ws_bytes < doda_rulebook_pairs_workspace_bytes(n_rows, K)) return DODA_ERR_WORKSPACE;
    const int nt = div_up(n_rows, PAIR_TILE);
    int32_t *counts = (int32_t *)ws;
    const int flip_tbl = flip & 1;
    if (!(flip & 2)) hipMemsetAsync(pairs, 0xFF, (size_t)2 * K * ld_pairs * 4, s);   // -1 padding (spconv format)
    hipLaunchKernelGGL(pairs_count, dim3(nt, K), dim3(PAIR_TILE), 0, s, tbl, ld, K, n_rows, flip_tbl,
                       counts, nt);
    hipLaunchKernelGGL(pairs_scan, dim3(K), dim3(256), 0, s, counts, nt, pair_num);
    hipLaunchKernelGGL(pairs_fill, dim3(nt, K), dim3(PAIR_TILE), 0, s, tbl, ld, K, n_rows, flip_tbl,
                       counts, nt, pairs, ld_pairs);
    return doda_check_launch();
}


// ---- generic geometry: host entry points ------------------------------------------------------
namespace {
struct ConvWs {
    unsigned long long *tab;
    uint32_t cap;
    int32_t *flag, *rank, *scan;
    size_t total;
};
ConvWs carve_conv(void *ws, int m, int K) {
    ConvWs r;
    const long long cand = (long long)(m > 0 ? m : 1) * K;
    const uint32_t cap = next_pow2((uint32_t)(2 * cand < 1024 ? 1024 : 2 * cand));
    size_t off = 0;
    char *p = (char *)ws;
    r.cap = cap;
    r.tab = (unsigned long long *)(p + off);
    off += (size_t)cap * 8;
    const size_t ci = align_up((size_t)cand * 4, 256);
    r.flag = (int32_t *)(p + off); off += ci;
    r.rank = (int32_t *)(p + off); off += ci;
    r.scan = (int32_t *)(p + off); off += align_up(scan_ws_ints((int)cand) * 4, 256);
    r.total = off;
    return r;
}
bool make_geo(const int32_t *shape, const int32_t *k, const int32_t *s, const int32_t *p, const int32_t *d,
              ConvGeo *g) {
    g->K = 1;
    int o[3];
    for (int a = 0; a < 3; ++a) {
        if (k[a] < 1 || s[a] < 1 || p[a] < 0 || d[a] < 1 || shape[a] < 1) return false;
        g->k[a] = k[a]; g->s[a] = s[a]; g->p[a] = p[a]; g->d[a] = d[a];
        g->K *= k[a];
        o[a] = (shape[a] + 2 * p[a] - d[a] * (k[a] - 1) - 1) / s[a] + 1;   // spconv get_conv_output_size
        if (o[a] < 1) return false;
    }
    g->out = GridDesc{o[0], o[1], o[2]};
    return true;
}
}  // namespace

extern "C" size_t doda_rulebook_conv_workspace_bytes(int32_t m, int32_t K) {
    if (m < 0 || K < 1 || K > 27 || (long long)m * K > 0x3fffffffll) return 0;
    return carve_conv(nullptr, m, K).total;
}

extern "C" int doda_rulebook_conv_assign(const int32_t *indices, int32_t m, const int32_t *shape_h,
                                         int32_t batch, const int32_t *ksize_h, const int32_t *stride_h,
                                         const int32_t *pad_h, const int32_t *dil_h, int32_t *out_shape_h,
                                         int32_t *count_out, void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (m < 0 || !shape_h || !ksize_h || !stride_h || !pad_h || !dil_h || !out_shape_h || !count_out)
        return DODA_ERR_INVALID;
    ConvGeo g;
    if (!make_geo(shape_h, ksize_h, stride_h, pad_h, dil_h, &g)) return DODA_ERR_INVALID;
    if (g.K > 27 || (long long)m * g.K > 0x3fffffffll) return DODA_ERR_UNSUPPORTED;
    out_shape_h[0] = g.out.X; out_shape_h[1] = g.out.Y; out_shape_h[2] = g.out.Z;
    hipStream_t s = as_stream(stream);
    if (m == 0) { hipMemsetAsync(count_out, 0, sizeof(int32_t), s); return DODA_OK; }
    if (!indices || !ws) return DODA_ERR_INVALID;
    HashFmt hf;   // values are touch keys j * K + rank
    if (!grid_fmt(batch, g.out.X, g.out.Y, g.out.Z, (unsigned long long)m * g.K, &hf)) return DODA_ERR_GRID_TOO_LARGE;
    const ConvWs w = carve_conv(ws, m, g.K);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    const int grid = div_up(m, 256);
    hipMemsetAsync(w.tab, 0xFF, (size_t)w.cap * 8, s);
    hipLaunchKernelGGL(conv_touch, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, g, w.tab, w.cap - 1, hf);
    hipLaunchKernelGGL(conv_first, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, g, w.tab, w.cap - 1, hf,
                       w.flag);
    return exclusive_scan_i32(w.flag, w.rank, m * g.K, count_out, w.scan, s);
}

extern "C" int doda_rulebook_conv_tables(const int32_t *indices, int32_t m, const int32_t *shape_h,
                                         int32_t batch, const int32_t *ksize_h, const int32_t *stride_h,
                                         const int32_t *pad_h, const int32_t *dil_h, int32_t m_out,
                                         int32_t *out_indices, int32_t *tbl, int32_t ld_out,
                                         int32_t *tbl_rev, int32_t ld_in, void *ws, size_t ws_bytes,
                                         doda_stream_t stream) {
    if (m < 0 || m_out < 0 || ld_out < m_out || ld_in < m || !shape_h || !ksize_h || !stride_h || !pad_h || !dil_h)
        return DODA_ERR_INVALID;
    if (m == 0) return DODA_OK;
    ConvGeo g;
    if (!make_geo(shape_h, ksize_h, stride_h, pad_h, dil_h, &g)) return DODA_ERR_INVALID;
    if (g.K > 27) return DODA_ERR_UNSUPPORTED;
    if (!indices || !out_indices || !tbl || !tbl_rev || !ws) return DODA_ERR_INVALID;
    HashFmt hf;   // the split doda_rulebook_conv_assign used for this table
    if (!grid_fmt(batch, g.out.X, g.out.Y, g.out.Z, (unsigned long long)m * g.K, &hf)) return DODA_ERR_GRID_TOO_LARGE;
    const ConvWs w = carve_conv(ws, m, g.K);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const int grid = div_up(m, 256);
    if (m_out > 0) hipMemsetAsync(tbl, 0xFF, (size_t)g.K * ld_out * 4, s);
    hipLaunchKernelGGL(conv_number, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, g, w.tab, w.cap - 1, hf,
                       w.flag, w.rank, (int4 *)out_indices);
    hipLaunchKernelGGL(conv_tables, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, g, w.tab, w.cap - 1, hf,
                       tbl, ld_out, tbl_rev, ld_in);
    return doda_check_launch();
}

// SubM with per-axis odd kernel sizes (K <= 27); the cubic 1 / 3 cases keep doda_rulebook_subm
extern "C" int doda_rulebook_subm_generic(const int32_t *indices, int32_t m, const int32_t *shape_h,
                                          int32_t batch, const int32_t *ksize_h, int32_t *nbr, int32_t ld,
                                          void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (m < 0 || ld < m || !shape_h || !ksize_h) return DODA_ERR_INVALID;
    for (int a = 0; a < 3; ++a)
        if (ksize_h[a] < 1 || ksize_h[a] % 2 == 0) return DODA_ERR_INVALID;
    const int K = ksize_h[0] * ksize_h[1] * ksize_h[2];
    if (K > 27) return DODA_ERR_UNSUPPORTED;
    if (m == 0) return DODA_OK;
    if (!indices || !nbr || !ws) return DODA_ERR_INVALID;
    HashFmt hf;
    if (!grid_fmt(batch, shape_h[0], shape_h[1], shape_h[2], (unsigned long long)m, &hf)) return DODA_ERR_GRID_TOO_LARGE;
    const RbWs w = carve(ws, m);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const GridDesc g{shape_h[0], shape_h[1], shape_h[2]};
    const int grid = div_up(m, 256);
    hipMemsetAsync(w.tab, 0xFF, (size_t)w.cap * 8, s);
    hipLaunchKernelGGL(subm_insert, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g, w.tab,
                       w.cap - 1, hf, nbr, ld, K, K);   // no prefill
    hipLaunchKernelGGL(subm_probe_generic, dim3(grid), dim3(256), 0, s, (const int4 *)indices, m, (int)batch, g,
                       ksize_h[0], ksize_h[1], ksize_h[2], w.tab, w.cap - 1, hf, nbr, ld);
    return doda_check_launch();
}
