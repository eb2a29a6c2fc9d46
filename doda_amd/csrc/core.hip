// ABI bookkeeping + the deterministic multi-block exclusive scan used by the rulebook builders.
#include "common.hpp"

extern "C" int doda_abi_version(void) { return DODA_ABI_VERSION; }

extern "C" const char *doda_strerror(int status) {
    switch (status) {
        case DODA_OK: return "ok";
        case DODA_ERR_INVALID: return "invalid argument";
        case DODA_ERR_LAUNCH: return "HIP kernel launch failed";
        case DODA_ERR_GRID_TOO_LARGE: return "cell id (batch*X*Y*Z) and row number do not fit one 64-bit hash word";
        case DODA_ERR_UNSUPPORTED: return "size outside the compiled range";
        case DODA_ERR_WORKSPACE: return "workspace too small";
        case DODA_ERR_NOMEM: return "host allocation failed";
        default: return "unknown doda status";
    }
}

// ------------------------------------------------------------------------------------------
namespace {
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_sum(int v, int *total, int *lds /*[4]*/) {
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    int inc = wave_inclusive_sum(v);
    if (lane == 63) lds[wid] = inc;
    doda_sync();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        int s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    doda_sync();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums(const int32_t *__restrict__ in, int n,
                                                             int32_t *__restrict__ tile_sum) {
    __shared__ int lds[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    block_exclusive_sum(s, &tot, lds);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of the tile sums, grand total to total_out
__global__ __launch_bounds__(SCAN_BLOCK) void scan_partials(int32_t *tile_sum, int n_tiles,
                                                            int32_t *total_out) {
    __shared__ int lds[4];
    int carry = 0;
    for (int start = 0; start < n_tiles; start += SCAN_BLOCK) {
        int i = start + threadIdx.x;
        int v = i < n_tiles ? tile_sum[i] : 0;
        int tot;
        int ex = block_exclusive_sum(v, &tot, lds);
        if (i < n_tiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const int32_t *__restrict__ in,
                                                         int32_t *__restrict__ out, int n,
                                                         const int32_t *__restrict__ tile_off) {
    __shared__ int lds[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int run = block_exclusive_sum(s, &tot, lds) + tile_off[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}
}  // namespace

size_t scan_ws_ints(int n) { return (size_t)div_up(n > 0 ? n : 1, SCAN_TILE) + 8; }

int exclusive_scan_i32(const int32_t *in, int32_t *out, int n, int32_t *total_out, int32_t *ws,
                       hipStream_t stream) {
    if (n <= 0) {
        if (total_out) hipMemsetAsync(total_out, 0, sizeof(int32_t), stream);
        return DODA_OK;
    }
    const int tiles = div_up(n, SCAN_TILE);
    hipLaunchKernelGGL(scan_tile_sums, dim3(tiles), dim3(SCAN_BLOCK), 0, stream, in, n, ws);
    hipLaunchKernelGGL(scan_partials, dim3(1), dim3(SCAN_BLOCK), 0, stream, ws, tiles, total_out);
    hipLaunchKernelGGL(scan_apply, dim3(tiles), dim3(SCAN_BLOCK), 0, stream, in, out, n, ws);
    return doda_check_launch();
}
