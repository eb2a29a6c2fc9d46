/*
 * doda_hip.h — C ABI of libdoda_hip.so: the MI355X (gfx950) native replacement for the
 * native layer under DODA's sparse-conv U-Net hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes, an explicit HIP stream
 * (passed as void*, i.e. a hipStream_t), and returns an int status (0 = DODA_OK, negative =
 * error, see doda_strerror).  Nothing here exits the process, allocates device memory or
 * synchronises the device unless the comment says so; all buffers are caller-owned (PyTorch's
 * caching allocator on the Python side), workspaces are sized by the *_workspace_bytes queries.
 * Pointers are DEVICE pointers unless the parameter name ends in `_h` (host).
 *
 * Each declaration cites the reference interface (file:line under the DODA checkout) it replaces.
 * spconv v1.2 is an un-vendored third-party dependency of the reference (docs/INSTALL.md:8,26);
 * for those entry points the citation is the reference CALL SITE plus the upstream op name.
 */
#ifndef DODA_HIP_H
#define DODA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DODA_ABI_VERSION 12

#define DODA_OK 0
#define DODA_ERR_INVALID (-1)        /* bad argument (null pointer, negative size, bad mode) */
#define DODA_ERR_LAUNCH (-2)         /* hipGetLastError() != hipSuccess after a launch          */
#define DODA_ERR_GRID_TOO_LARGE (-3) /* bits(batch*X*Y*Z) + bits(rows) > 64: cell id and row number do not fit one hash word */
#define DODA_ERR_UNSUPPORTED (-4)    /* channel count / k outside the compiled range            */
#define DODA_ERR_WORKSPACE (-5)      /* workspace smaller than the *_workspace_bytes answer     */
#define DODA_ERR_NOMEM (-6)          /* host allocation failed (host entry points only)         */

typedef void *doda_stream_t; /* hipStream_t */

int doda_abi_version(void);
const char *doda_strerror(int status);

/* ------------------------------------------------------------------------------------------
 * Voxelisation (pointgroup_ops)
 * ---------------------------------------------------------------------------------------- */

/* HOST, fork-safe, no GPU context.  Replaces PG_OP.voxelize_idx
 * (lib/pointgroup_ops/src/pointgroup_ops_api.cpp:7 -> voxelize/voxelize.cpp:10-31).
 * The reference resizes its output tensors; a C ABI cannot, so the call is split:
 *   doda_voxelize_idx_h      : hashes the points, fills input_map[n], returns M and maxActive
 *                              and an opaque handle holding the per-voxel point lists;
 *   doda_voxelize_idx_fill_h : writes output_coords[M,ncol] and output_map[M,1+maxActive]
 *                              and releases the handle (doda_voxelize_idx_free_h on error paths).
 * coords_h: int64 [n, ncol], ncol = 3 (single batch) or 4 (column 0 = batch index).
 * mode: 0 unique, 1 first point, 2 last point, 3 sum, 4 mean (reference numbering,
 * voxelize.cpp:54,119-152; modes 1/2 follow the CODE, not its comment). */
int doda_voxelize_idx_h(const int64_t *coords_h, int32_t n, int32_t ncol, int32_t batch_size,
                        int32_t mode, int32_t *input_map_h, void **handle, int32_t *n_active,
                        int32_t *max_active);
int doda_voxelize_idx_fill_h(void *handle, const int64_t *coords_h, int64_t *output_coords_h,
                             int32_t *output_map_h);
void doda_voxelize_idx_free_h(void *handle);

/* DEVICE version of the same map (same results as the host call).
 * Stage 1 (assign): input_map[n] and counts_out[2] = {M, maxActive} (device ints the caller
 * copies back before allocating the outputs of stage 2).  Stage 2 (fill): output_coords,
 * output_map.  `ws` must hold doda_voxelize_idx_workspace_bytes(n) bytes and be passed
 * unchanged to both stages. */
size_t doda_voxelize_idx_workspace_bytes(int32_t n);
int doda_voxelize_idx_assign(const int64_t *coords, int32_t n, int32_t ncol, int32_t mode,
                             int32_t *input_map, int32_t *counts_out, void *ws, size_t ws_bytes,
                             doda_stream_t stream);
int doda_voxelize_idx_fill(const int64_t *coords, int32_t n, int32_t ncol, int32_t mode,
                           int32_t n_active, int32_t max_active, int64_t *output_coords,
                           int32_t *output_map, void *ws, size_t ws_bytes, doda_stream_t stream);

/* Replaces PG_OP.voxelize_fp / voxelize_bp (pointgroup_ops_api.cpp:8-9 -> voxelize.cpp:159-181
 * -> voxelize.cu:10-53).  out[r,:] += mult * feats[rule[r,i],:], i = 1..rule[r,0], products
 * rounded before accumulation, in point order; mult = 1/cnt when mode == 4.  `out` must be
 * zero-initialised by the caller as in the reference wrapper (pointgroup_ops.py:59). */
int doda_voxelize_fp(const float *feats, float *out, const int32_t *rules, int32_t mode,
                     int32_t n_active, int32_t max_active, int32_t n_plane, doda_stream_t stream);
int doda_voxelize_bp(const float *d_out, float *d_feats, const int32_t *rules, int32_t mode,
                     int32_t n_active, int32_t max_active, int32_t n_plane, doda_stream_t stream);
/* The network's input rows in one launch (round 5): what model/unet.py:89-94 builds with torch.cat + PG_OP.voxelize_fp
 * (pointgroup_ops.py:44-64) + the cast to the network's feature type, plus the zero channels this library's input layer appends
 * (32-byte rows for the tile kernels).  out[r, 0:c_a+c_b] = pooled row of (feats_a | feats_b) in voxelize_fp's arithmetic (fp32,
 * products rounded before the add, point order), written as fp32 (out_elem_bytes 4) or bf16 round-to-nearest-even (2);
 * out[r, c_a+c_b : c_out] = 0.  `out` [n_active, c_out] is overwritten (no zero-initialisation needed); feats_b may be NULL
 * with c_b = 0.  Not differentiable (the reference's input features carry no gradient). */
int doda_voxelize_fp_rows(const float *feats_a, int32_t c_a, const float *feats_b, int32_t c_b, const int32_t *rules, int32_t mode,
                          int32_t n_active, int32_t max_active, void *out, int32_t c_out, int32_t out_elem_bytes,
                          doda_stream_t stream);
/* PG_OP.point_recover_fp / _bp (pointgroup_ops_api.cpp:10-11 -> voxelize.cpp:184-205): the two
 * kernels above with roles swapped and no averaging. */
int doda_point_recover_fp(const float *feats, float *out, const int32_t *rules, int32_t n_active,
                          int32_t max_active, int32_t n_plane, doda_stream_t stream);
int doda_point_recover_bp(const float *d_out, float *d_feats, const int32_t *rules,
                          int32_t n_active, int32_t max_active, int32_t n_plane,
                          doda_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse-convolution rulebooks  (spconv v1.2 `ops.get_indice_pairs` ->
 * torch.ops.spconv.get_indice_pairs; reference call sites model/unet.py:36,
 * model/unet_block.py:26,29,48,70,78)
 *
 * Native rulebook format ("gather table"): int32 tbl[K][ld]; tbl[o][t] = row of the INPUT
 * feature matrix that output row t reads for kernel offset o, or -1.  Offsets are numbered
 * row-major over (k0,k1,k2) exactly as spconv numbers them.
 * ---------------------------------------------------------------------------------------- */
size_t doda_rulebook_workspace_bytes(int32_t m);
/* ABI 7 (round 4).  doda_rulebook_subm (ksize 3) and doda_rulebook_down2_assign look voxels up through a 64-bit hash table
 * inside `ws`.  A caller that hands over a LARGER workspace — align256(doda_rulebook_workspace_bytes(m)) + 4 * cells bytes,
 * cells = batch * X * Y * Z of the grid the call looks up in (the input shape for subm, the output shape (s - 2) / 2 + 1
 * for down2), cells <= 2^28 (ABI 9; 2^26 before; DODA_RULEBOOK_GRID_MAX_LOG2) — gets a direct-address grid instead (grid[cell] = row
 * or -1: one 4-byte read per probe, the three z-neighbours adjacent; a batch of 2 cm scenes is 16.5 M cells = 66 MB, of four 1 cm
 * scenes 85 M cells = 340 MB).  Same tables, same first-touch numbering; larger grids and the minimum workspace keep the hash.
 * DODA_RULEBOOK_GRID=0 disables the grid. */

/* SubMConv3d, odd cubic kernel `ksize` (1 or 3), stride 1, padding ksize/2, dilation 1.
 * indices: int32 [m,4] = (batch, x, y, z).  nbr: int32 [ksize^3][ld], ld >= m.
 * nbr[o][t] = id of the active voxel at p_t + k(o) - ksize/2, or -1 (out of shape / inactive). */
int doda_rulebook_subm(const int32_t *indices, int32_t m, const int32_t *shape_h, int32_t batch,
                       int32_t ksize, int32_t *nbr, int32_t ld, void *ws, size_t ws_bytes,
                       doda_stream_t stream);

/* SparseConv3d kernel 2 stride 2 padding 0 (unet_block.py:70), stage 1: parent[j] = output row
 * of input j (or -1 when its cell lies outside the output shape (s-2)/2+1), off[j] = kernel
 * offset ((x&1)*2+(y&1))*2+(z&1), out_indices[q] for q < M_out, counts_out[0] = M_out (device).
 * Output rows are numbered in first-touch order over ascending input index, which is the order
 * spconv's CPU path produces. */
int doda_rulebook_down2_assign(const int32_t *indices, int32_t m, const int32_t *shape_h,
                               int32_t batch, int32_t *parent, int32_t *off,
                               int32_t *out_indices, int32_t *counts_out, void *ws,
                               size_t ws_bytes, doda_stream_t stream);
/* stage 2: child[8][ld_out] (child[o][q] = fine row with parent q and offset o, or -1) and
 * par_off[8][ld_in] (par_off[o][j] = parent[j] if off[j]==o else -1). */
int doda_rulebook_down2_tables(const int32_t *parent, const int32_t *off, int32_t m,
                               int32_t m_out, int32_t *child, int32_t ld_out, int32_t *par_off,
                               int32_t ld_in, doda_stream_t stream);

/* Generic geometry (spconv get_indice_pairs for any kernel_size / stride / padding / dilation with
 * K = k0*k1*k2 <= 27; SURVEY §8f rank 4 — DODA itself only instantiates SubM k1/k3 and k2 s2).
 * Outputs are numbered in the first-touch order of spconv's serial scan (inputs ascending, each
 * input's valid output positions in getValidOutPos order).  Two stages around one size read-back,
 * sharing `ws` (doda_rulebook_conv_workspace_bytes(m, K)), which must not be touched in between:
 *   assign: out_shape_h (host, 3 ints) and *count_out (device) = number of outputs;
 *   tables: out_indices [m_out][4], tbl[K][ld_out] (tbl[o][t] = input row feeding output t through
 *           kernel offset o, row-major offset index) and tbl_rev[K][ld_in] (the output that input j
 *           feeds through offset o) — the same pair of tables as child / par_off above, so forward,
 *           inverse and gradients run through the same doda_spconv_* calls.
 * doda_rulebook_subm_generic: SubM with per-axis odd kernel sizes (nbr[o][t], row-major offsets). */
size_t doda_rulebook_conv_workspace_bytes(int32_t m, int32_t K);
int doda_rulebook_conv_assign(const int32_t *indices, int32_t m, const int32_t *shape_h, int32_t batch,
                              const int32_t *ksize_h, const int32_t *stride_h, const int32_t *pad_h,
                              const int32_t *dil_h, int32_t *out_shape_h, int32_t *count_out, void *ws,
                              size_t ws_bytes, doda_stream_t stream);
int doda_rulebook_conv_tables(const int32_t *indices, int32_t m, const int32_t *shape_h, int32_t batch,
                              const int32_t *ksize_h, const int32_t *stride_h, const int32_t *pad_h,
                              const int32_t *dil_h, int32_t m_out, int32_t *out_indices, int32_t *tbl,
                              int32_t ld_out, int32_t *tbl_rev, int32_t ld_in, void *ws, size_t ws_bytes,
                              doda_stream_t stream);
int doda_rulebook_subm_generic(const int32_t *indices, int32_t m, const int32_t *shape_h, int32_t batch,
                               const int32_t *ksize_h, int32_t *nbr, int32_t ld, void *ws,
                               size_t ws_bytes, doda_stream_t stream);

/* Export a gather table as spconv-v1.2-format indice pairs: pairs int32 [2][K][ld_pairs]
 * (-1 padded), pair_num int32 [K].  List o holds (in = j, out = tbl[src(o)][j]) for ascending j
 * with tbl[src(o)][j] >= 0, src(o) = K-1-o when `flip & 1` (SubM: in/out roles are mirrored) else o
 * (down2: pass par_off).  Ascending-j order is the order of spconv's CPU path.  `flip & 2`: entries
 * past pair_num[o] are left unwritten instead of -1 (lists for doda_spconv_wgrad_pairs_bf16, which
 * is bounded by pair_num: saves the 8*K*ld-byte fill).
 * On return `ws` starts with the SEGMENT PREFIX of the lists, int32 [K][nt] with
 * nt = ceil(n_rows / doda_rulebook_pairs_tile()): entry [o][t] = number of pairs of list o with
 * j < t * tile (the stable compaction's own tile prefix); doda_spconv_wgrad_pairs_bf16 reads it. */
int32_t doda_rulebook_pairs_tile(void);
size_t doda_rulebook_pairs_workspace_bytes(int32_t n_rows, int32_t K);
int doda_rulebook_pairs(const int32_t *tbl, int32_t ld, int32_t K, int32_t n_rows, int32_t flip,
                        int32_t *pairs, int32_t ld_pairs, int32_t *pair_num, void *ws,
                        size_t ws_bytes, doda_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse-convolution arithmetic (spconv v1.2 torch.ops.spconv.indice_conv /
 * indice_conv_backward behind Fsp.indice_subm_conv / indice_conv / indice_inverse_conv;
 * reference call sites as above).  All output-stationary: every output row is written once,
 * no atomics, deterministic.
 *
 *   y[t, :] = sum_o  x[tbl[o][t], :] . B_o          t < n_out,  x: [n_in, kc],  y: [n_out, nc]
 *
 * w_layout 0: B_o = W[o]        with w stored [K][kc][nc]           (forward)
 * w_layout 1: B_o = W[o]^T      with w stored [K][nc][kc]           (data-grad of down2/inverse)
 * w_layout 2: B_o = W[K-1-o]^T  with w stored [K][nc][kc]           (data-grad of SubM)
 *
 * `ws` (doda_spconv_gather_workspace_bytes(K,kc,nc, sizeof element)) receives the weights
 * re-laid-out in MFMA fragment order by a pre-pack kernel inside the call.  K <= 27.
 * ---------------------------------------------------------------------------------------- */
size_t doda_spconv_gather_workspace_bytes(int32_t K, int32_t kc, int32_t nc, int32_t elem_bytes);

/* Pre-packing many weight tensors in ONE launch (weights change once per optimizer step; a U-Net
 * step otherwise issues 2 x 71 pack launches).  A descriptor is 8 bytes `w` (device, fp32),
 * 8 bytes `out` (device, doda_spconv_gather_workspace_bytes each), then int32 K, kc, nc, layout
 * (0..2), elem_bytes (2|4) and three int32 the planner fills (doda_spconv_pack_desc_bytes() bytes
 * in all).  doda_spconv_pack_plan_h completes the host descriptors and the inclusive block prefix;
 * the caller uploads both and calls doda_spconv_pack_multi.  A gather call then passes the packed
 * buffer as `w` with w_layout | 0x100 (and may pass ws = NULL).  The fragment order (16-channel,
 * 32-channel "wide", or offset-"pair" for 16-channel bf16 layers) is chosen by the library from
 * (K, kc, elem_bytes) alone, identically in the planner and in the gather; a gather call that
 * cannot use the order its buffer was packed in (unaligned / > 2 GB features, which take the
 * generic kernel) returns DODA_ERR_UNSUPPORTED and the caller passes the fp32 weights instead. */
size_t doda_spconv_pack_desc_bytes(void);
int doda_spconv_pack_plan_h(void *descs_h, int32_t n_desc, int32_t *blk_end_h, int32_t *total_blocks);
int doda_spconv_pack_multi(const void *descs_dev, const int32_t *blk_end_dev, int32_t n_desc,
                           int32_t total_blocks, doda_stream_t stream);

/* bf16 feature storage (BASELINE config 2): x / y are bf16 bit patterns; weights arrive fp32 and
 * are rounded to bf16 by the pre-pack; fp32 accumulate; y rounded to bf16 (RNE) once, or kept
 * fp32 when y_is_f32 (y then points to float [n_out, nc]; used by the point head's logits).
 *
 * ABI 7: doda_spconv_gather_ex is the ONE gather entry point (epi == NULL: the plain convolution); the per-dtype
 * doda_spconv_gather_{f32,bf16} / _gather_add_* of ABI <= 6 were its special cases and are gone.  With `residual`
 * y = (sum_o x[tbl[o][t]] . B_o) + res, res: [n_out, nc] in the dtype of y — the `output.features +=
 * identity.features` of the reference's ResidualBlock (model/unet_block.py:36); the sum is taken in fp32 before the
 * single rounding of y. */

/* Gather with epilogue options: the conv fused with its neighbours in the layer graph (reference
 * model/unet_block.py:23-37,46-49,67-79: BatchNorm1d -> ReLU -> conv [-> + identity]).
 *   residual : y = conv + residual ([n_out, nc], dtype of y), as doda_spconv_gather_add_*.
 *   stats    : BatchNorm statistics accumulated while the output rows are stored — the separate read
 *              pass over y (forward) or over dy and x (backward) of a BatchNorm disappears:
 *     bn_x == NULL (forward call; the BatchNorm FOLLOWS the conv):
 *              stats[p][0][c] = sum_t y[t,c], stats[p][1][c] = sum_t y[t,c]^2 over the rows of
 *              workgroup tile p, y as stored (after its bf16 rounding);
 *     bn_x != NULL (data-grad call; y = d loss / d z with z = [relu](bn(bn_x)), the BatchNorm PRECEDES
 *              the conv): stats[p][0][c] = sum dz, stats[p][1][c] = sum dz * xhat, with
 *              xhat = (bn_x - mean) * invstd and dz = y * [gamma * xhat + beta > 0] when bn_relu;
 *              bn_mean / bn_invstd / bn_gamma / bn_beta: float [nc] each.
 *     `stats` holds doda_spconv_stats_capacity(n_out) rows of 2*nc floats; *stats_rows_h (HOST) receives
 *     the number of rows written (known when the call returns: it depends on the tile the library
 *     picks, not on device data).  Feed them to doda_bn_relu_fwd_stats / doda_bn_relu_bwd_stats.
 * Statistics need the fast kernel (kc, nc multiples of 4, 16-byte aligned, < 2 GB): otherwise
 * DODA_ERR_UNSUPPORTED and the caller runs the plain call plus a standalone BatchNorm. */
typedef struct doda_conv_epilogue {
    const void *residual;
    float *stats;
    int32_t *stats_rows_h;
    const void *bn_x;
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
    int32_t bn_relu;
    int32_t tilebook_rows;   /* ABI 3: rows the tilebook was built for (must equal n_out) */
    const void *tilebook;    /* ABI 3: doda_tilebook_build of `tbl`, or NULL */
    int32_t residual_bcast;   /* ABI 6: `residual` is ONE row [nc] (dtype of y) added to every output row — a bias (the Linear
                               * head of reference model/unet.py:64 as a gather-GEMM); dense-table kernels only */
    double *stats_totals;     /* ABI 9: with `stats`: DODA_STATS_TOTALS_DOUBLES(nc) doubles — DODA_STATS_SLOTS x 2 x nc / 4 groups of 16
                               * (a 128-byte line per four channels; the first four doubles of a group are used) —, ZERO on entry (or holding the sums of
                               * other calls over the same rows), to which the kernel's workgroups ADD their (sum, sum of squares)
                               * with fp64 atomics instead of writing rows into `stats` (which must still be non-NULL: it selects
                               * the statistics epilogue, nothing is written there).  Feed them to doda_bn_relu_fwd_totals /
                               * doda_bn_relu_bwd_totals: BatchNorm in ONE launch, no reduction launch in front of it.  Every addend
                               * is an fp32 value: a sum of a few thousand is exact in fp64 unless the magnitudes span more than
                               * ~2^16, and then differs between runs by an ulp of fp64 — the statistics are rounded to fp32. */
    int32_t x_ld, y_ld, residual_ld, bn_x_ld;   /* ABI 11: row strides in ELEMENTS of x / y / residual / bn_x (0 = dense: kc, nc, nc, nc;
                               * multiples of 4): a column slice of a wider matrix — one half of a U-Net level's concatenation
                               * (reference model/unet_block.py:89-93: the two producing convs write the halves, no torch.cat
                               * launch) — is read / written in place.  Dense-table fast kernel only (else DODA_ERR_UNSUPPORTED) */
    const struct doda_conv_prologue *prologue;   /* ABI 11: a BatchNorm folded into the gather (below), or NULL */
} doda_conv_epilogue;
/* ABI 11.  The BatchNorm1d(+ReLU) in FRONT of a convolution folded into that convolution's gather — the BatchNorm launch of
 * reference model/unet_block.py:23-30,46-49,67-79 (BatchNorm1d -> ReLU -> conv, 65 pairs per forward pass of model/unet.py:42-45)
 * disappears, forward and backward.  Where it pays: layers of a few thousand rows and fewer (U-Net levels 4-7), where a BatchNorm
 * sweep is a launch-floor kernel; every gathered row is transformed once per kernel offset that reads it.
 *   kind 1, forward: y = sum_o xn[tbl[o][t]] . B_o with xn = [relu]((x - mean) * invstd * gamma + beta) rounded to the storage
 *       type.  Training: mean / invstd come from `totals` (the fp64 sums the conv that PRODUCED x accumulated:
 *       doda_conv_epilogue.stats_totals; `totals_b` / `c_a`: x is a channel concatenation [a | b] of two producers), every
 *       workgroup derives them itself; the launch writes mean / invstd, updates running_mean / running_var (momentum, unbiased
 *       variance) and num_batches_tracked once.  totals == NULL: evaluation mode, the running statistics are used as they are.
 *       `side` [rows, kc] (row stride side_ld) receives xn — the operand of the layer's weight gradient.
 *   kind 2, backward: x = dz, the data gradient a later conv produced for the BatchNorm's OUTPUT with the sums (sum dz', sum dz' xhat)
 *       in `totals` (its data-grad statistics epilogue, bn_x = aux); the gathered rows are
 *       du = gamma invstd ([gamma xhat + beta > 0] dz - mean(dz') - xhat mean(dz' xhat)), xhat = (aux - mean) invstd — the gradient of the
 *       BatchNorm's INPUT — and feed the data gradient of the conv in front of that BatchNorm; kind 3: du + add (the skip
 *       connection's gradient, reference model/unet_block.py:36).  `side` receives du; dgamma / dbeta are written (accumulate:
 *       added to).  aux / add: [rows, kc] with row strides aux_ld / add_ld.
 * Arithmetic per element = doda_bn_relu_fwd_totals / doda_bn_relu_bwd_totals (same operations, same order: same bits).  Needs
 * 16-byte rows pieces: bf16 with kc >= 32, kc % 8 == 0, or fp32; kc <= 256; same dtype in and out. */
typedef struct doda_conv_prologue {
    int32_t kind, relu;
    int32_t rows;            /* rows of x, aux, add, side (= n_in of the call) */
    int32_t c_a;             /* kind 1: channels covered by `totals` when totals_b != NULL */
    const double *totals, *totals_b;
    float eps, momentum;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    float *mean, *invstd;    /* kind 1: written in training mode (unused in evaluation mode); kind >= 2: read */
    void *side;
    int32_t side_ld;
    int32_t aux_ld, add_ld;
    int32_t accumulate;
    const void *aux, *add;
    float *dgamma, *dbeta;
} doda_conv_prologue;
#define DODA_STATS_SLOTS 8
#define DODA_STATS_TOTALS_DOUBLES(nc) ((size_t)DODA_STATS_SLOTS * 2 * 16 * ((size_t)(nc) / 4))
/* ABI 3.  Tile-local form of a SubM gather table ("tilebook") for the LDS-staged convolution kernel:
 * per tile of doda_tilebook_tile() consecutive output rows, the sorted list of DISTINCT input rows the
 * tile's K x tile table entries reference and, per entry, its position in that list.  The kernel loads
 * every distinct row once into LDS and serves all K gathers of the tile from there (replaces spconv's
 * per-offset sparse_gather kernels the same way the dense table does — reference call sites
 * model/unet_block.py:26,29,48 —, with ~1/3 of the vector-memory instructions).  Built once per
 * rulebook, used by every forward and data-grad call of its layers through doda_conv_epilogue.tilebook
 * when the features are bf16 with 16 input channels and K == 27; other calls ignore it.  Results equal
 * the dense-table path up to the fp32 summation order over offsets.  K must be 27 (else bytes == 0 /
 * DODA_ERR_UNSUPPORTED); `tilebook` 16-byte aligned. */
int32_t doda_tilebook_tile(void);
int32_t doda_tilebook_umax(void);   /* list slots per tile (1024; a tile whose neighbourhood has more than 1023 distinct rows is
                                     * an overflow tile).  Layout: ulist int32 [nt][umax]; lidx uint32 [nt][10][tile] — ten planes
                                     * of packed 10-bit local indices (1 + the position of tbl[o][row] in ulist, 0 = no neighbour):
                                     * offset o of row t is bits 10 ((o / 2) % 3) .. + 9 of word ((t & 0xC0) | (sigma(t & 15) << 2)
                                     * | ((t >> 4) & 3)), sigma = bits 2 and 3 exchanged, of plane 5 (o & 1) + (o / 2) / 3; ucount int32 [nt]; n_over int32 [2] (the
                                     * last eight bytes) = tiles whose neighbourhood exceeds the 64-byte-row / 32-byte-row staging
                                     * capacity (they are served from the dense table: a caller whose voxel order has no locality
                                     * reads n_over and stops building tilebooks).  56 bytes per output row (ABI 7; 70 with the
                                     * uint16 strip before) */
size_t doda_tilebook_bytes(int32_t n_rows, int32_t K);
int doda_tilebook_build(const int32_t *tbl, int32_t ld, int32_t K, int32_t n_rows, void *tilebook,
                        size_t tilebook_bytes, doda_stream_t stream);
/* ABI 7.  A/B switches of the kernel selection (measurements and parity tests; all default to 1; they replace the
 * doda_spconv_set_*_kernel functions of ABI <= 6).  doda_set_option returns DODA_ERR_INVALID for an unknown option,
 * doda_get_option -1. */
#define DODA_OPT_TILE_KERNEL 1   /* 0: ignore tilebooks in doda_spconv_gather_ex (dense-table kernels only) */
#define DODA_OPT_WLDS_KERNEL 2   /* 0: the 48 -> 48 channel layers stay on the streaming-weights kernel */
#define DODA_OPT_WDMA_KERNEL 3   /* 0: weight-gradient jobs ignore their tilebook (pair-list / gather-table kernels) */
#define DODA_OPT_TILE_PIPELINE 4 /* 0: 16 -> 16 layers stay on conv_tile whatever the tile count (default 1: tables of >= 769 tiles —
                                  * DODA_TILE16_MIN_TILES — take the cross-tile pipelined conv_tile16; results are bit-identical) */
#define DODA_OPT_TILE_DUAL 5     /* 0: 32-output-channel tile layers take one channel block per pass over the units (default 1:
                                  * both in one pass; results are bit-identical) */
#define DODA_OPT_CONV_UP 6       /* 0: K <= 8 gathers with 32 input channels and fewer input than output rows (inverse convolution
                                  * forward, strided convolution data gradient) stay on the offset-by-offset kernel (default 1:
                                  * conv_up32, one gather per output row) */
#define DODA_OPT_PRE_FWD_ROWS 7  /* ABI 11, not a 0 / 1 switch: doda_layers_run folds a BNFWD op of at most this many rows into the next
                                  * convolution's gather (0: never; default 16384 or DODA_PRE_FWD_ROWS) */
#define DODA_OPT_PRE_BWD_ROWS 8  /* ABI 11: the same for BNBWD ops (default 0 or DODA_PRE_BWD_ROWS) */
int doda_set_option(int32_t option, int32_t value);
int32_t doda_get_option(int32_t option);
size_t doda_spconv_stats_capacity(int32_t n_out);
int doda_spconv_gather_ex(const void *x, int32_t n_in, int32_t kc, int32_t elem_bytes, const float *w,
                          int32_t nc, const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out, void *y,
                          int32_t y_is_f32, int32_t w_layout, void *ws, size_t ws_bytes,
                          const doda_conv_epilogue *epi, doda_stream_t stream);

/* Weight gradient  dw[o][i][j] = sum_t a[tbl[o][t], i] * b[t, j],  a: [*, ca], b: [n_rows, cb], dw: fp32 [K][ca][cb]
 * (spconv indice_conv_backward's per-offset `Xg^T . dYg`).  ABI 7: doda_spconv_wgrad_multi is the ONE entry point (a
 * single layer is a call with one job); the per-dtype single-layer calls and doda_spconv_wgrad_pairs_bf16 of ABI <= 6
 * were its special cases and are gone.  Three kernels behind a job, chosen by the library:
 *   gather table : every dtype and channel count, K <= 28; per-row-chunk partials, fixed-order reduce;
 *   pair lists   : bf16, ca % 16 == 0, cb % 16 == 0, spconv-format lists of PRESENT pairs only
 *                  dw[o] (+)= sum_{p < pair_num[o]} a[pair_in[o*ld + p]]^T b[pair_out[o*ld + p]]; pair_num (device) and
 *                  pair_seg [K][seg_nt] come from doda_rulebook_pairs (pair_seg[o][t] = pairs of list o whose `in` row
 *                  lies below tile t, 256 rows per tile).  A workgroup takes one range of `in` rows and walks, one
 *                  offset per wave, the matching segment of every list: a and b stream from HBM once.  pair_num ==
 *                  NULL and pair_seg == NULL with K == 1: the list holds exactly ld pairs in row order (the 1x1
 *                  convolution).  Rows reach MFMA k-order through a one-hot MFMA, no LDS round trip;
 *   tilebook     : bf16, ca and cb multiples of 16 up to 64, K == 27, rulebooks of >= 32 768 rows: LDS-staged, as 16 x 16
 *                  channel blocks over row-strided slices of a and b (see `tilebook` below).
 * All deterministic (per-chunk / per-workgroup partials in ws, fixed-order reduce). */

/* Weight gradients of MANY layers in one call (one launch per kernel variant + one reduce launch
 * instead of two launches per layer; the coarse levels' small grids run concurrently).  The weight
 * gradient of a layer does not feed the rest of the backward pass, so a caller can queue the jobs
 * while back-propagating and issue them together before the optimizer step.  `jobs_h` is a host
 * array; `ws` (doda_spconv_wgrad_multi_workspace_bytes) holds the partials, `desc_dev`
 * (doda_spconv_wgrad_multi_desc_bytes) receives the device descriptors.  Results equal the per-layer
 * calls up to the summation order of the partial reduce (both deterministic). */
typedef struct doda_wgrad_job {
    const void *a;        /* [n_a, ca]    features gathered through tbl / pair_in (fp32 or bf16) */
    const void *b;        /* [n_rows, cb] output gradient, same dtype */
    const int32_t *tbl;   /* [K][ld] gather table (may be NULL when the pair lists are given and usable) */
    float *dw;            /* [K][ca][cb] fp32, overwritten (or accumulated into: DODA_WGRAD_ACCUMULATE) */
    int32_t ca, cb, ld, K, n_rows, elem_bytes;
    /* ABI 2.  Optional spconv-format pair lists of the same rulebook (doda_rulebook_pairs):
     * list o = pairs p < pair_num[o] of (a row pair_in[o*pair_ld+p], b row pair_out[o*pair_ld+p]).
     * bf16 jobs with ca % 16 == 0 and cb % 16 == 0 then run the pair kernel (only PRESENT pairs are
     * walked); other jobs use tbl. */
    const int32_t *pair_in, *pair_out, *pair_num;   /* pair_num NULL (K == 1): the list holds pair_ld pairs */
    int32_t pair_ld;      /* leading dimension of pair_in / pair_out */
    int32_t n_a;          /* rows of a (bounds the hardware range check of the pair kernel) */
    int32_t flags;        /* DODA_WGRAD_* */
    int32_t reserved;
    const int32_t *pair_seg;     /* [K][pair_seg_nt] segment prefix of the lists (see doda_rulebook_pairs), with */
    int32_t pair_seg_nt;         /* pair_num; both NULL / 0 for the full identity lists of a 1x1 convolution   */
    int32_t reserved2;
    /* ABI 4.  Optional tilebook of `tbl` (doda_tilebook_build, built for n_rows rows): bf16 jobs with K == 27 and 16 .. 64
     * channels on either side (ABI 7; ABI <= 6: ca == cb == 16) then run the LDS-staged kernel (spconv_wdma.hip: per 16 x 16
     * channel block the tile's distinct a rows and its b rows are staged once per 256 rows by LDS-DMA instead of being
     * gathered per pair); jobs of one call that share a tilebook share a launch, and the SUMMATION ORDER of a job depends on
     * the other tile jobs of its call (workgroups take contiguous chunks of the call's (block, tile) list): results are
     * deterministic per call, equal across call shapes to fp32 rounding. */
    const void *tilebook;
} doda_wgrad_job;
#define DODA_WGRAD_ACCUMULATE 1   /* dw += result (second backward pass into an existing .grad) */
size_t doda_spconv_wgrad_multi_workspace_bytes(const doda_wgrad_job *jobs_h, int32_t n_jobs);
size_t doda_spconv_wgrad_multi_desc_bytes(int32_t n_jobs);
int doda_spconv_wgrad_multi(const doda_wgrad_job *jobs_h, int32_t n_jobs, void *ws, size_t ws_bytes,
                            void *desc_dev, size_t desc_bytes, doda_stream_t stream);

/* Indice-pair max pooling (spconv v1.2 indice_maxpool / indice_maxpool_backward; unused by
 * DODA, named by north_star).  y[t,c] = max_o x[tbl[o][t],c] over present o (0 if none);
 * dx[tbl[o][t],c] += dy[t,c] where x[tbl[o][t],c] == y[t,c]. */
int doda_maxpool_fwd_f32(const float *x, int32_t c, const int32_t *tbl, int32_t ld, int32_t K,
                         int32_t n_out, float *y, doda_stream_t stream);
int doda_maxpool_bwd_f32(const float *x, const float *y, const float *dy, int32_t c,
                         const int32_t *tbl, int32_t ld, int32_t K, int32_t n_out, float *dx,
                         doda_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused BatchNorm1d(+ReLU) on [m, c] features (the BN -> ReLU pairs that
 * precede every conv, reference model/unet.py:28,42-45, model/unet_block.py:23-30,46-49,67-79;
 * upstream these are torch.nn.BatchNorm1d + ReLU applied to `.features` by SparseSequential).
 * x / y / dy / dx: fp32 (elem_bytes 4) or bf16 (2); statistics and parameters fp32; c % 4 == 0.
 * training != 0: batch statistics (biased variance, eps inside the sqrt) are computed, written to
 * save_mean / save_invstd and folded into running_mean / running_var (momentum, unbiased variance)
 * when those are non-null, and *num_batches_tracked (int64, nullable) is incremented; training == 0: save_mean / save_invstd must already hold the running
 * mean and 1/sqrt(running_var + eps).  Backward recomputes the ReLU mask from x.
 * ---------------------------------------------------------------------------------------- */
size_t doda_bn_workspace_bytes(int32_t m, int32_t c);
int doda_bn_relu_fwd(const void *x, int32_t m, int32_t c, int32_t elem_bytes, float eps,
                     float momentum, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, int64_t *num_batches_tracked, int32_t training,
                     int32_t relu, void *y, float *save_mean, float *save_invstd, void *ws,
                     size_t ws_bytes, doda_stream_t stream);
int doda_bn_relu_bwd(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes,
                     const float *save_mean, const float *save_invstd, const float *gamma,
                     const float *beta, int32_t relu, void *dx, float *dgamma, float *dbeta,
                     void *ws, size_t ws_bytes, doda_stream_t stream);
/* Same, plus a second gradient of x summed into dx: dx = BN-backward(dy) + add (dtype of x).
 * In a pre-activation residual block (model/unet_block.py:23-37) x feeds both the first BatchNorm and
 * the skip connection; autograd would add the two gradients with an extra elementwise pass.
 * ABI 7: `add` is ROW-STRIDED — row r of the second gradient starts at add + r * add_ld elements (add_ld >= c, a
 * multiple of 4; `add` aligned to 4 elements; add_ld == c: dense) — a column slice of a wider matrix, i.e. the gradient
 * torch.cat's backward hands to one of its inputs (the U-Net's skip connection, reference model/unet_block.py:93): no
 * copy into a dense matrix, no separate accumulation kernel. */
int doda_bn_relu_bwd_add(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes,
                         const float *save_mean, const float *save_invstd, const float *gamma,
                         const float *beta, int32_t relu, const void *add, int32_t add_ld, void *dx,
                         float *dgamma, float *dbeta, void *ws, size_t ws_bytes, doda_stream_t stream);

/* BatchNorm(+ReLU) whose statistics pass already happened in a conv epilogue (doda_spconv_gather_ex):
 * `stats` = [stats_rows][2][c] partial sums.  Forward: (sum x, sum x^2) -> mean / invstd / running
 * statistics (fp64 combine), then y = [relu]((x - mean) * invstd * gamma + beta): two launches instead
 * of three, x read once instead of twice.  Backward: (sum dz, sum dz*xhat) -> dgamma, dbeta and
 * dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)) [+ add]: dy and x read once instead of
 * twice.  coef_ws: 3*c floats of scratch. */
int doda_bn_relu_fwd_stats(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const float *stats,
                           int32_t stats_rows, float eps, float momentum, const float *gamma,
                           const float *beta, float *running_mean, float *running_var,
                           int64_t *num_batches_tracked, int32_t relu, void *y, float *save_mean,
                           float *save_invstd, doda_stream_t stream);
int doda_bn_relu_bwd_stats(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes,
                           const float *stats, int32_t stats_rows, const float *save_mean,
                           const float *save_invstd, const float *gamma, const float *beta, int32_t relu,
                           const void *add, int32_t add_ld, void *dx, float *dgamma, float *dbeta,
                           float *coef_ws, doda_stream_t stream);   /* add / add_ld as doda_bn_relu_bwd_add; add may be NULL */
/* ABI 9.  The same two operators over TOTALS (doda_conv_epilogue.stats_totals) — reference torch.nn.BatchNorm1d + ReLU applied by
 * SparseSequential (model/unet_block.py:23-30,46-49), ONE launch each: every workgroup of the sweep derives the per-channel
 * vectors from the DODA_STATS_TOTALS_DOUBLES(c) totals itself.  totals_b / c_a (forward): the columns [c_a, c) of x come from a second
 * producer with its own totals (DODA_STATS_TOTALS_DOUBLES(c - c_a)) — the U-Net level's concatenation (unet_block.py:93); NULL: one
 * producer.  c <= 256. */
int doda_bn_relu_fwd_totals(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const double *totals,
                            const double *totals_b, int32_t c_a, float eps, float momentum, const float *gamma,
                            const float *beta, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                            int32_t relu, void *y, float *save_mean, float *save_invstd, doda_stream_t stream);
int doda_bn_relu_bwd_totals(const void *x, const void *dy, int32_t m, int32_t c, int32_t elem_bytes, const double *totals,
                            const float *save_mean, const float *save_invstd, const float *gamma, const float *beta,
                            int32_t relu, const void *add, int32_t add_ld, void *dx, float *dgamma, float *dbeta,
                            doda_stream_t stream);
/* ABI 6.  The apply half alone: y = relu?((x - mean) * invstd * gamma + beta) with given per-channel vectors (e.g. from two
 * doda_bn_fwd_final calls over the two halves of a channel concatenation, whose statistics rows come from the two convs
 * that produced the halves — reference model/unet_block.py:93 followed by :23). */
int doda_bn_relu_apply(const void *x, int32_t m, int32_t c, int32_t elem_bytes, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int32_t relu, void *y, doda_stream_t stream);
/* ABI 6.  The reduction half of doda_bn_relu_fwd_stats alone (training mode): save_mean / save_invstd [c] from the
 * partial rows, running statistics and num_batches_tracked updated (any of the three may be NULL) — with
 * doda_bn_relu_apply: the BatchNorm after a channel concatenation, fed by the statistics rows of its two halves. */
int doda_bn_fwd_final(const float *stats, int32_t stats_rows, int32_t m, int32_t c, float eps, float momentum,
                      float *running_mean, float *running_var, int64_t *num_batches_tracked, float *save_mean,
                      float *save_invstd, doda_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Neighbour queries
 * ---------------------------------------------------------------------------------------- */

/* Replaces pointops2_cuda.knnquery_cuda (lib/pointops2/src/pointops_api.cpp:13 ->
 * knnquery/knnquery_cuda.cpp:8-17 -> knnquery_cuda_kernel.cu:65-116).  offset/new_offset are
 * the END offsets per batch item (length nbatch; the wrapper strips the leading 0,
 * pointops2.py:66).  idx int32 [m,nsample], dist2 f32 [m,nsample] ascending, tie order = the
 * reference's max-heap + heap-sort order.  nsample <= 100 (reference array bound). */
int doda_knnquery(int32_t m, int32_t nsample, const float *xyz, const float *new_xyz,
                  const int32_t *offset, const int32_t *new_offset, int32_t nbatch, int32_t *idx,
                  float *dist2, doda_stream_t stream);

/* Replaces PG_OP.knn_batch (pointgroup_ops_api.cpp:26 -> knn/knn.cpp:8-19 -> knn.cu:7-73):
 * for each of the n points of `xyz`, its k nearest in query_xyz[offsets[b]:offsets[b+1]),
 * insertion order, strict '<' (first seen wins ties).  k <= 40. */
int doda_knn_batch(int32_t n, int32_t m, int32_t k, const float *xyz, const float *query_xyz,
                   const int32_t *batch_idxs, const int32_t *query_batch_offsets, int32_t *idx,
                   doda_stream_t stream);

/* Replaces PG_OP.ballquery_batch_p (pointgroup_ops_api.cpp:13 -> bfs_cluster.cpp:15-25 ->
 * bfs_cluster.cu:15-90).  Deterministic two-pass form: start_len[i] = (exclusive prefix of
 * counts, count) in point order, neighbours ascending; at most 1000 neighbours per point and
 * nothing written at or beyond n*mean_active, as in the reference.  *total_h receives the sum
 * of counts; like the reference's blocking cudaMemcpy this call synchronises `stream`.
 * ws: doda_ballquery_workspace_bytes(n). */
size_t doda_ballquery_workspace_bytes(int32_t n);
int doda_ballquery_batch_p(int32_t n, int32_t mean_active, float radius, const float *xyz,
                           const int32_t *batch_idxs, const int32_t *batch_offsets, int32_t *idx,
                           int32_t *start_len, int32_t *total_h, void *ws, size_t ws_bytes,
                           doda_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Cross-entropy with ignore_index on the point logits (reference model/unet.py:107-108,196:
 * nn.CrossEntropyLoss(ignore_index) after the Linear head): mean over the non-ignored points.
 * fwd: lse float [n] (saved for bwd), out float [2] = {mean loss, number of valid points};
 * bwd: dlogits = (softmax - onehot) * grad[0] / n_valid, zero rows for ignored points.  c <= 64.
 * ---------------------------------------------------------------------------------------- */
size_t doda_cross_entropy_workspace_bytes(int32_t n);
int doda_cross_entropy_fwd(const float *logits, const int64_t *labels, int32_t n, int32_t c,
                           int64_t ignore_index, float *lse, float *out, void *ws, size_t ws_bytes,
                           doda_stream_t stream);
int doda_cross_entropy_bwd(const float *logits, const int64_t *labels, const float *lse, const float *out,
                           const float *grad, int32_t n, int32_t c, int64_t ignore_index, float *dlogits,
                           doda_stream_t stream);

/* ABI 11.  The point head and its loss without the point-level score matrix (csrc/head.hip) — reference model/unet.py:62-64
 * (feats_pt = out.features[p2v]; scores = Linear(feats_pt)) + :107-108,196 (CrossEntropyLoss(ignore_index)).  Every point of a
 * voxel reads the same feature row, so its logits are its voxel's: loss = (1 / n_valid) sum_v [cnt_v lse(z_v) - sum_{p in v} z_v[y_p]],
 * z_v = W f_v + b — two sweeps over the VOXEL rows and their point lists instead of five over a [points, classes] fp32 matrix.
 *   feats [m, c] bf16 / fp32 (c == 16), weight fp32 [n_cls, c] (rounded to bf16 for bf16 features, as the gather-GEMM's pre-pack
 *   does), bias fp32 [n_cls] or NULL, v2p int32 [m, v2p_ld] = (count, point ids ...) per voxel (doda_voxelize_idx's output_map:
 *   every point in exactly one row), labels int64 [points].
 *   fwd: out float [2] = {mean loss over the valid points, n_valid}; pred int32 [m] = argmax class per voxel (or NULL);
 *        partial_ws: float [2 * doda_head_ce_blocks(m)] scratch.
 *   bwd: d_feats [m, c] and dz [m, n_cls] in the features' storage type (dz = d loss / d z_v summed over the voxel's points: the
 *        operand of the weight gradient dW = dz^T feats through doda_spconv_wgrad_multi over an identity table);
 *        db_partial float [doda_head_ce_blocks(m)][n_cls]: the caller adds the rows (fixed order) for d bias.  grad float [1]. */
/*   doda_head_dw_bf16: the weight gradient of that head from bf16 feats [m, 16] and dz [m, n_cls] (voxels as the MFMA k dimension):
 *        partial float [doda_head_dw_blocks(m)][32][16] — rows = classes (zero past n_cls), the caller adds the blocks (fixed order);
 *        fp32 features take doda_spconv_wgrad_multi over an identity table instead. */
int32_t doda_head_dw_blocks(int32_t m);
int doda_head_dw_bf16(const void *feats, const void *dz, int32_t m, int32_t c, int32_t n_cls, float *partial, int32_t n_blocks,
                      doda_stream_t stream);
int32_t doda_head_ce_blocks(int32_t m);
int doda_head_ce_fwd(const void *feats, int32_t m, int32_t c, int32_t elem_bytes, const float *weight, const float *bias,
                     int32_t n_cls, const int32_t *v2p, int32_t v2p_ld, const int64_t *labels, int64_t ignore_index, float *out,
                     int32_t *pred, float *partial_ws, int32_t n_blocks, doda_stream_t stream);
int doda_head_ce_bwd(const void *feats, int32_t m, int32_t c, int32_t elem_bytes, const float *weight, const float *bias,
                     int32_t n_cls, const int32_t *v2p, int32_t v2p_ld, const int64_t *labels, int64_t ignore_index, const float *out,
                     const float *grad, void *d_feats, void *dz, float *db_partial, int32_t n_blocks, doda_stream_t stream);

/* ABI 6.  Glue of the head and the input layer (csrc/glue.hip).
 * doda_cast_colsum_f32_bf16: y = bf16(x) (round to nearest even, as torch's cast) and partial[b][c] = sum of the rows
 *   workgroup b swept — the Linear head's backward (reference model/unet.py:64) needs the score gradient as a bf16 GEMM
 *   operand AND summed over the points (d_bias): one pass instead of a cast and a reduce kernel.  c <= 64; `partial`:
 *   float [doda_cast_colsum_blocks(n, c)][c]; the caller adds the rows (fixed order: deterministic).
 * doda_pad_channels: y[r, :c_in] = x[r, :], y[r, c_in:c_out] = 0 (the xyz input layer's rows padded to 4 / 16 channels,
 *   spconv/conv.py): torch's constant_pad_nd = a fill + a strided copy. */
int32_t doda_cast_colsum_blocks(int64_t n, int32_t c);
int doda_cast_colsum_f32_bf16(const float *x, int64_t n, int32_t c, uint16_t *y, float *partial, int32_t n_blocks,
                              doda_stream_t stream);
int doda_pad_channels(const void *x, int64_t n, int32_t c_in, int32_t c_out, int32_t elem_bytes, void *y,
                      doda_stream_t stream);

/* ---- optimizer step -------------------------------------------------------------------------------
 * SGD over all parameter tensors of a network in one launch (replaces the five multi-tensor launches of
 * torch.optim.SGD(fused=True) the reference's trainer would issue per step: util/common_utils.py:196-215,
 * tool/train.py:255-262 optimizer.step()).  torch's update rule term by term:
 *   g' = g (+ weight_decay * p);  buf = first_step ? g' : momentum * buf + (1 - dampening) * g';
 *   g'' = nesterov ? g' + momentum * buf : buf  (g' when momentum == 0);  p -= lr * g''
 * with the hyper-parameters as doubles.  p, g, buf: fp32 device arrays of n elements (buf may be NULL
 * when momentum == 0); first_step: the buffer holds no value yet.  desc_dev: device scratch of
 * doda_sgd_multi_desc_bytes(n_tensors) bytes, owned by the call until the stream has passed it. */
typedef struct {
    float *p;
    const float *g;
    float *buf;
    int64_t n;
    int32_t first_step;
    int32_t reserved;
} doda_sgd_tensor;
size_t doda_sgd_multi_desc_bytes(int32_t n_tensors);
int doda_sgd_multi(const doda_sgd_tensor *tensors_h, int32_t n_tensors, double lr, double momentum,
                   double dampening, double weight_decay, int32_t nesterov, int32_t maximize, void *desc_dev,
                   size_t desc_bytes, doda_stream_t stream);

/* ABI 12.  Segmentation meters — reference util/common_utils.py:233-246 intersectionAndUnionGPU (per iteration through update_meter,
 * :249; tool/test.py:82): hist[0][c] += valid points with pred == label == c, hist[1][c] += valid points predicted c, hist[2][c] +=
 * valid points labelled c (union = hist[1] + hist[2] - hist[0]); valid: label != ignore_index and 0 <= label < k; a point's
 * prediction is preds[p2v ? p2v[p] : p] (int32, or int64 with preds_are_int64) clamped to [0, k - 1].  hist: int64 [3][k],
 * ACCUMULATED (zero it once); k <= 256.  One launch, integer atomics: exact and order-independent. */
int doda_seg_meters(const void *preds, int32_t preds_are_int64, const int32_t *p2v, const int64_t *labels, int32_t n, int32_t k,
                    int64_t ignore_index, int64_t *hist, doda_stream_t stream);

/* ---- ABI 11 (ABI 12: any level — a GEMM op may carry its table's tilebook): U-Net levels as an op list of per-layer launches (csrc/layers.hip) ----------------------------------------
 * The levels of DODA's U-Net (ABI 11: the deep ones; ABI 12: any, from the root UBlock down) — reference model/unet_block.py:55-100 (UBlock: blocks -> strided conv -> UBlock -> inverse conv
 * -> concatenation -> blocks_tail) with model/unet_block.py:9-37 inside (ResidualBlock: BatchNorm1d -> ReLU -> SubMConv3d, twice,
 * + skip) — are described by the caller as a HOST array of ops and issued by doda_layers_run as whole-chip launches, back to back,
 * with nothing of the caller's (interpreter, autograd nodes, allocations) in between; the arithmetic per layer is that of
 * doda_spconv_gather_ex (fp32 accumulate, one rounding at the store) and of the BatchNorm sweeps over fp64 totals.  (ABI 8-10 also
 * had a persistent single-XCD executor walking the same list, doda_coarse_run: removed in ABI 11 — it lost to the per-layer
 * kernels at every batch size once those were issued from inside the library; DESIGN.md.)
 *
 * All feature tensors are [rows, c] in ONE storage type per call (bf16 or fp32) with a row stride in elements (`*_ld`: a column
 * slice of a wider matrix — the halves of a concatenation — is addressed in place); channel counts are multiples of 8 (bf16) / 4
 * (fp32), at most 256.  Every statistics array (`stats`, `stats_b`) is DODA_STATS_TOTALS_DOUBLES(c) doubles of fp64 TOTALS
 * (doda_conv_epilogue.stats_totals), zero before the op that accumulates into them.  `n_part` must be 0.
 *
 * DODA_CX_GEMM   y[t, :] = sum_o x[tbl[o][t], :] . B_o (+ res[t, :])   t < rows; `w` = the fragment-packed weights of
 *                doda_spconv_pack_multi for (K, c_in, c_out, elem_bytes) in the op's direction; tbl int32 [K][tbl_ld]
 *                (DODA_CX_F_IDENTITY: K = 1 over an identity table — the 1x1 convolution).
 *                aux == NULL (forward): stats += (sum y, sum y^2), y as stored.
 *                aux != NULL (data gradient; aux = input of the BatchNorm in FRONT of the conv, mean / invstd / gamma / beta its
 *                vectors): stats += (sum dz, sum dz * xhat), dz = y * [gamma * xhat + beta > 0] (the mask only with
 *                DODA_CX_F_RELU), xhat = (aux - mean) * invstd; y itself is stored UNMASKED.   stats may be NULL.
 * DODA_CX_BNFWD  y = [relu]((x - mean) * invstd * gamma + beta) over c_in channels.  DODA_CX_F_TRAINING: mean / invstd from the
 *                totals `stats` (first c_split channels) and `stats_b` (the rest: x is a concatenation), written to mean /
 *                invstd; running_mean / running_var / nbt updated when given.  Otherwise the running statistics are used.
 * DODA_CX_BNBWD  dx = gamma * invstd * ([gamma * xhat + beta > 0] dz - mean(dz') - xhat * mean(dz' * xhat)) (+ res): x = dz (unmasked; the
 *                op masks with DODA_CX_F_RELU and needs `beta`), aux = the BatchNorm's input, stats = the GEMM's backward totals;
 *                columns < c_split go to y, the rest to y2 (the two halves of a concatenation's gradient as two dense tensors);
 *                dgamma / dbeta written (or, with DODA_CX_F_ACCUM, added to).
 * DODA_CX_STATS  stats += (sum x, sum x^2).
 * A BatchNorm op whose output only feeds the NEXT op's gather is folded into that convolution (doda_conv_prologue): BNFWD ; GEMM ->
 * one launch, BNBWD ; GEMM -> one launch, where the BatchNorm has at most DODA_PRE_FWD_ROWS (16384) / DODA_PRE_BWD_ROWS (0: the backward
 * fold is built and tested but measured slower on the GPU) rows (environment; doda_set_option).  The folded and unfolded forms of a
 * list give the same bits.  DODA_CX_F_BARRIER is accepted and ignored (stream order). */
#define DODA_CX_GEMM 1
#define DODA_CX_BNFWD 2
#define DODA_CX_BNBWD 3
#define DODA_CX_STATS 4
#define DODA_CX_F_BARRIER 1
#define DODA_CX_F_IDENTITY 2
#define DODA_CX_F_RELU 4
#define DODA_CX_F_TRAINING 8
#define DODA_CX_F_ACCUM 16
typedef struct doda_cx_op {
    int32_t kind, flags;
    int32_t rows;            /* output rows (GEMM) / rows of the tensor */
    int32_t rows_in;         /* GEMM: rows of x */
    int32_t c_in, c_out;     /* GEMM: channels of x / y; other kinds: c_in = channels */
    int32_t K, tbl_ld;
    int32_t x_ld, y_ld, res_ld, aux_ld, y2_ld;
    int32_t n_part;          /* must be 0 (ABI 8-10: the executor's partial-row count) */
    int32_t c_split;
    int32_t reserved;
    float eps, momentum;
    const void *x;
    const void *w;
    const int32_t *tbl;
    void *y;
    void *y2;
    const void *res;
    const void *aux;
    float *stats;            /* fp64 totals (doda_conv_epilogue.stats_totals), typed float * for ABI 8 compatibility */
    const float *stats_b;
    const float *gamma, *beta;
    float *mean, *invstd;
    float *running_mean, *running_var;
    int64_t *nbt;
    float *dgamma, *dbeta;
    const void *tilebook;    /* ABI 12, GEMM: the tilebook of `tbl` (doda_tilebook_build over rows == this op's `rows`), or NULL */
} doda_cx_op;
/* *n_launches_h (optional, HOST) receives the number of kernel launches issued. */
int doda_layers_run(const doda_cx_op *ops_h, int32_t n_ops, int32_t elem_bytes, int32_t *n_launches_h, doda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DODA_HIP_H */
