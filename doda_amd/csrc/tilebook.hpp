// Tile-local form of a SubM gather table ("tilebook"), shared by the builder (tilebook.hip) and the
// LDS-staged convolution kernels (spconv_tile.hip: conv_tile, bwd_tile).
//
// The dense table tbl[K][M] makes every (offset, row) slot a vector-memory gather of its own: at level 1
// of DODA's U-Net a 32-row wave issues 28 gather instructions at 37 % lane use, and the kernel is paced by
// the per-CU texture path.  A tile of TB_T consecutive output rows references only
// ~2.2 x TB_T distinct input rows (raster-ordered surface voxels), so the tilebook stores per tile
//   ulist [TB_UMAX]   int32   the distinct input rows, ascending (-1 past the count; every entry -2 when the tile
//                             has more than TB_UMAX of them), stored in the order tb_upos() gives: the 16 bytes a
//                             lane of the LDS-DMA kernel needs for its four row pieces are contiguous
//   lidx  [TB_T][9]   uint32  round 4: per output row t nine words, word w = the local indices of offsets 3w, 3w+1, 3w+2 in
//                             bits 0-9 / 10-19 / 20-29; a local index = 1 + position of tbl[o][t] in ulist, or 0 when absent
//                             (LDS slot 0 = the zero row).  36 B per row (rounds 2-3: uint16 [K][TB_T] = 54 B per row: the
//                             strip was the largest single item of the tile kernels' HBM traffic)
//   ucount            int32   number of distinct rows (> TB_LMAX: the list is not kept)
// plus, once per tilebook, n_over int32 [2]: tiles above TB_CAP64 / above TB_UMAX (the caller's safety valve:
// voxel orders without locality overflow everywhere and are better served by the dense-table kernels).
// A kernel stages at most its own capacity (what its LDS budget allows for its row size: tile_cap()) and
// serves a tile above it from the dense table.
// and the kernel loads each distinct row ONCE, coalesced, into LDS and serves all K gathers from there.
// Rows are stored in tile order (row t at words 9 t .. 9 t + 8): sixteen lanes reading one word of sixteen consecutive rows
// hit sixteen different LDS banks (stride 9 words).
#pragma once
#include <stdint.h>
#include <stddef.h>

constexpr int TB_T = 256;        // output rows per tile
constexpr int TB_UMAX = 1024;    // list capacity per tile = the largest kernel capacity (32-byte rows)
constexpr int TB_LMAX = 1023;    // most distinct rows a tile may have and keep its list (local indices are 10 bits: 1 .. 1023)
constexpr int TB_CAP64 = 960;    // kernel capacity for 64-byte rows (2 workgroups per CU)
constexpr int TB_K = 27;
constexpr int TB_LW = 9;         // 32-bit words of local indices per output row (three 10-bit indices each)
constexpr int TB_LIDX_BYTES = TB_T * TB_LW * 4;   // 9216 per tile

struct TileBookView {
    int32_t *ulist;     // [nt][TB_UMAX]
    uint32_t *lidx;     // [nt][TB_T][TB_LW]
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

// Storage position of list entry e.  The LDS-DMA kernel (spconv_dma.hip) moves a tile's rows with four DMA
// instructions per wave; lane l of wave w stages, with instruction k, the row of entry (k*8 + w)*32 + (l >> 1).
// Stored at (w*32 + (l >> 1))*4 + k, the four entries of a lane are ONE 16-byte load (8 list instructions per tile
// instead of 40 four-byte ones: these kernels are paced by the number of vector-memory instructions per CU).
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline int tb_upos(int e) { return (((e >> 5) & 7) * 32 + (e & 31)) * 4 + (e >> 8); }
static_assert(TB_UMAX == 1024, "tb_upos permutes exactly 1024 entries");

// local index of offset o in a row's nine words
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline unsigned tb_lidx_get(const uint32_t *row_words, int o) { return (row_words[o / 3] >> (10 * (o % 3))) & 0x3ffu; }
