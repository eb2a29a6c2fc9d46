// Tile-local form of a SubM gather table ("tilebook"), shared by the builder (tilebook.hip) and the
// LDS-staged kernels (spconv_tile.hip: conv_tile, conv_tile16; spconv_wdma.hip: wgrad_dma16).
//
// The dense table tbl[K][M] makes every (offset, row) slot a vector-memory gather of its own: at level 1
// of DODA's U-Net a 32-row wave issues 28 gather instructions at 37 % lane use, and the kernel is paced by
// the per-CU texture path.  A tile of TB_T consecutive output rows references only
// ~2.2 x TB_T distinct input rows (raster-ordered surface voxels), so the tilebook stores per tile
//   ulist [TB_UMAX]   int32   the distinct input rows, ascending (-1 past the count; every entry -2 when the tile
//                             has more than TB_UMAX of them), stored in the order tb_upos() gives: the 16 bytes a
//                             lane of the LDS-DMA kernel needs for its four row pieces are contiguous
//   lidx  [10][TB_T]  uint32  round 4: ten planes of 256 words.  Plane 5 p + k (p = offset parity, k = 0..4) holds, for every
//                             output row, the local indices of the offsets 2 (3 k + q) + p, q = 0, 1, 2, in bits 10 q .. 10 q + 9;
//                             a local index = 1 + position of tbl[o][t] in ulist, or 0 when absent (LDS slot 0 = the zero
//                             row; the fields of the offsets 27, 28, 29 are 0).  Inside a plane row t sits at word tb_lpos(t):
//                             the four rows a lane of the tile kernels owns (one per 16-row subtile) are adjacent, so ONE 16-byte
//                             LDS read returns three units' worth of indices for all four (the pair kernels take offsets
//                             2 u and 2 u + 1 in a unit: lanes differ in the plane, not in the shift).  40 B per row.
//                             (Rounds 2-3: uint16 [K][TB_T], 54 B per row; the first packed form — nine words per row, row-major
//                             — needed eight 4-byte LDS reads per unit and wave and a per-lane shift.)
//   ucount            int32   number of distinct rows (> TB_LMAX: the list is not kept)
// plus, once per tilebook, n_over int32 [2]: tiles above TB_CAP64 / above TB_UMAX (the caller's safety valve:
// voxel orders without locality overflow everywhere and are better served by the dense-table kernels).
// A kernel stages at most its own capacity (what its LDS budget allows for its row size: tile_cap()) and
// serves a tile above it from the dense table.
// and the kernel loads each distinct row ONCE, coalesced, into LDS and serves all K gathers from there.
#pragma once
#include <stdint.h>
#include <stddef.h>

constexpr int TB_T = 256;        // output rows per tile
constexpr int TB_UMAX = 1024;    // list capacity per tile = the largest kernel capacity (32-byte rows)
constexpr int TB_LMAX = 1023;    // most distinct rows a tile may have and keep its list (local indices are 10 bits: 1 .. 1023)
constexpr int TB_CAP64 = 960;    // kernel capacity for 64-byte rows (2 workgroups per CU)
constexpr int TB_K = 27;
constexpr int TB_LW = 10;        // planes of local indices (32-bit words per output row, three 10-bit indices each)
constexpr int TB_LIDX_BYTES = TB_T * TB_LW * 4;   // 10240 per tile

struct TileBookView {
    int32_t *ulist;     // [nt][TB_UMAX]
    uint32_t *lidx;     // [nt][TB_LW][TB_T]
    int32_t *ucount;    // [nt]
    int32_t *n_over;    // [2]
    int nt;
};

static inline size_t tilebook_bytes_for(long long n_rows) {
    const size_t nt = (size_t)((n_rows + TB_T - 1) / TB_T);
    return nt * ((size_t)TB_UMAX * 4 + (size_t)TB_LIDX_BYTES + 4) + 8;
}

static inline TileBookView tilebook_view(void *base, long long n_rows) {
    TileBookView v;
    v.nt = (int)((n_rows + TB_T - 1) / TB_T);
    char *p = (char *)base;
    v.ulist = (int32_t *)p;
    p += (size_t)v.nt * TB_UMAX * 4;
    v.lidx = (uint32_t *)p;
    p += (size_t)v.nt * TB_LIDX_BYTES;
    v.ucount = (int32_t *)p;
    p += (size_t)v.nt * 4;
    v.n_over = (int32_t *)p;
    return v;
}

// Storage position of list entry e.  The LDS-DMA kernel (spconv_wdma.hip) moves a tile's rows with four DMA
// instructions per wave; lane l of wave w stages, with instruction k, the row of entry (k*8 + w)*32 + (l >> 1).
// Stored at (w*32 + (l >> 1))*4 + k, the four entries of a lane are ONE 16-byte load (8 list instructions per tile
// instead of 40 four-byte ones: these kernels are paced by the number of vector-memory instructions per CU).
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline int tb_upos(int e) { return (((e >> 5) & 7) * 32 + (e & 31)) * 4 + (e >> 8); }
static_assert(TB_UMAX == 1024, "tb_upos permutes exactly 1024 entries");

// Where the local index of (row t of the tile, offset o) lives: plane, bit offset, word inside the plane
#if defined(__HIPCC__)
#define TB_HD __host__ __device__
#else
#define TB_HD
#endif
TB_HD static inline int tb_lplane(int o) { return (o & 1) * 5 + (o >> 1) / 3; }
TB_HD static inline unsigned tb_lshift(int o) { return 10u * (unsigned)((o >> 1) % 3); }
// (row t = 64 w + 16 s + i of the tile: word 64 w + 4 sigma(i) + s, sigma = i with its bits 2 and 3 exchanged — any order of
// the sixteen 16-byte slots serves the tile kernels' 16-byte reads; this one also keeps the weight-gradient kernel's 4-byte
// reads, whose lanes hold rows i and i + 8 in one 32-lane group, off each other's banks)
TB_HD static inline int tb_lsigma(int i) { return (i & 3) | ((i & 4) << 1) | ((i & 8) >> 1); }
TB_HD static inline int tb_lpos(int t) { return (t & 0xC0) | (tb_lsigma(t & 15) << 2) | ((t >> 4) & 3); }
TB_HD static inline unsigned tb_lidx_get(const uint32_t *tile_words, int t, int o) {
    return (tile_words[tb_lplane(o) * TB_T + tb_lpos(t)] >> tb_lshift(o)) & 0x3ffu;
}
#undef TB_HD
