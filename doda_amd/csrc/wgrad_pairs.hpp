// Internal interface between spconv_wgrad.hip (job planning, descriptor upload) and
// spconv_wgrad_pairs.hip (the pair-list weight-gradient kernels).  Not part of the C ABI.
#pragma once
#include "common.hpp"
#include <vector>

namespace doda_pairs {

// true when the job can run on the pair kernel (bf16, 16-channel multiples, pair lists given or
// identity, operands inside the 4 GB hardware range check)
bool eligible(const doda_wgrad_job &j);

// bytes of per-chunk partials the job needs in the caller's workspace (multiple of 256)
size_t partial_bytes(const doda_wgrad_job &j);

struct Prepared {
    std::vector<unsigned char> desc;   // device descriptors, laid out [kernel groups][reduce jobs]
    struct Group { int ta, tb, first, count, blocks; };
    std::vector<Group> groups;         // one launch each; `first` = index of the group's first PJob
    size_t reduce_off = 0;             // byte offset of the reduce descriptors inside desc
    int n_reduce = 0, reduce_blocks = 0;
};

// Plans the eligible jobs (partials carved from ws_base + offsets starting at *ws_off, which is
// advanced).  Returns DODA_OK or an error.
int prepare(const doda_wgrad_job *jobs, const int *which, int n, char *ws_base, size_t *ws_off, Prepared *out);

// desc_dev: device copy of out.desc (already enqueued on `s`)
int launch(const Prepared &p, const void *desc_dev, hipStream_t s);

size_t desc_bytes_per_job();

}  // namespace doda_pairs

// LDS-staged weight gradient over a tilebook (spconv_wdma.hip): bf16 layers of 16 / 32 channels sharing one table
namespace doda_wdma {
bool enabled();
void set_enabled(bool on);
int max_jobs();
size_t partial_bytes(int n_rows);      // workspace per 16 x 16 channel block
// One 16 x 16 channel block of a layer's weight gradient: x / dy point at the block's first channel (bf16), rows x_stride /
// dy_stride bytes apart; the block's corner in the layer's dw [27][ca][cb] and its strides (ldo = ca * cb, ldc = cb).
struct Block { const void *x, *dy; float *dw; int x_stride, dy_stride, ldo, ldc, accumulate; };
int launch(const Block *blocks, int n_blocks, const int32_t *tbl, int ld, int n_rows, const void *tilebook, void *part,
           hipStream_t s);
}  // namespace doda_wdma
