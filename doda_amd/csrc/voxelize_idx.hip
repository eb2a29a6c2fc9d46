// Point -> voxel index maps: PG_OP.voxelize_idx, host and device versions.
//
// Reference semantics (lib/pointgroup_ops/src/voxelize/voxelize.cpp:10-155): voxel ids are
// handed out in first-occurrence order over the point list; coordinates are compared after
// narrowing int64 -> int32 (voxelize.cpp:73,90: `p[j] = coords[j]` with Point = array<Int>);
// per-voxel point lists are ascending; modes 3/4 pad rows to 1+maxActive with -1, modes 0/1/2
// emit rows [1, p] with p = only / first (`front()`) / last (`back()`) point of the voxel.
// The result does not depend on the hash container, so the google::dense_hash_map of the
// reference is replaced by a flat open-addressing table (host) / an index-only CAS table whose
// slots converge to the smallest point index of the voxel (device).
#include "common.hpp"
#include <new>
#include <vector>

// ================================= host =================================================
namespace {
struct HostVox {
    int32_t n = 0, ncol = 0, mode = 0, n_active = 0, max_active = 1;
    std::vector<int32_t> vid;    // voxel of each point
    std::vector<int32_t> count;  // points per voxel
};

inline uint64_t mix64(uint64_t z) {
    z ^= z >> 33; z *= 0xff51afd7ed558ccdull;
    z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull;
    z ^= z >> 33;
    return z;
}

struct Key4 { int32_t b, x, y, z; };
inline bool operator==(const Key4 &a, const Key4 &c) { return a.b == c.b && a.x == c.x && a.y == c.y && a.z == c.z; }
inline uint64_t hash4(const Key4 &k) {
    return mix64(((uint64_t)(uint32_t)k.b << 32 | (uint32_t)k.x) ^ mix64((uint64_t)(uint32_t)k.y << 32 | (uint32_t)k.z));
}
}  // namespace

extern "C" int doda_voxelize_idx_h(const int64_t *coords, int32_t n, int32_t ncol,
                                   int32_t batch_size, int32_t mode, int32_t *input_map,
                                   void **handle, int32_t *n_active, int32_t *max_active) {
    (void)batch_size;  // only an initial container size in the reference (voxelize.cpp:68,95-97)
    if (!handle || !n_active || !max_active || n < 0 || (ncol != 3 && ncol != 4) || mode < 0 || mode > 4)
        return DODA_ERR_INVALID;
    if (n > 0 && (!coords || !input_map)) return DODA_ERR_INVALID;
    HostVox *h = new (std::nothrow) HostVox();
    if (!h) return DODA_ERR_NOMEM;
    try {
        h->n = n; h->ncol = ncol; h->mode = mode;
        h->vid.resize(n);
        uint64_t cap = 1024;
        while (cap < 2ull * (uint64_t)n) cap <<= 1;
        std::vector<Key4> keys;  // key of each voxel id
        keys.reserve(n / 2 + 1);
        std::vector<int32_t> slot_vid(cap, -1);  // open-addressing table: slot -> voxel id
        const uint64_t mask = cap - 1;
        for (int32_t i = 0; i < n; ++i) {
            const int64_t *c = coords + (int64_t)i * ncol;
            Key4 k;
            if (ncol == 4) { k.b = (int32_t)c[0]; k.x = (int32_t)c[1]; k.y = (int32_t)c[2]; k.z = (int32_t)c[3]; }
            else { k.b = 0; k.x = (int32_t)c[0]; k.y = (int32_t)c[1]; k.z = (int32_t)c[2]; }
            uint64_t s = hash4(k) & mask;
            int32_t v;
            for (;;) {
                v = slot_vid[s];
                if (v < 0) {
                    v = h->n_active++;
                    slot_vid[s] = v;
                    keys.push_back(k);
                    h->count.push_back(0);
                    break;
                }
                if (keys[v] == k) break;
                s = (s + 1) & mask;
            }
            h->vid[i] = v;
            h->count[v]++;
            input_map[i] = v;
        }
        h->max_active = 1;
        if (mode == 3 || mode == 4)
            for (int32_t cnt : h->count) if (cnt > h->max_active) h->max_active = cnt;
        if (mode == 0)
            for (int32_t cnt : h->count) if (cnt != 1) { delete h; return DODA_ERR_INVALID; }
    } catch (const std::bad_alloc &) {
        delete h;
        return DODA_ERR_NOMEM;
    }
    *handle = h;
    *n_active = h->n_active;
    *max_active = h->max_active;
    return DODA_OK;
}

extern "C" int doda_voxelize_idx_fill_h(void *handle, const int64_t *coords,
                                        int64_t *output_coords, int32_t *output_map) {
    HostVox *h = (HostVox *)handle;
    if (!h) return DODA_ERR_INVALID;
    const int32_t M = h->n_active, W = h->max_active + 1;
    if (M > 0 && (!coords || !output_coords || !output_map)) { delete h; return DODA_ERR_INVALID; }
    for (int64_t e = 0; e < (int64_t)M * W; ++e) output_map[e] = -1;
    if (h->mode == 3 || h->mode == 4) {
        for (int32_t v = 0; v < M; ++v) output_map[(int64_t)v * W] = 0;
        for (int32_t i = 0; i < h->n; ++i) {  // ascending i => ascending lists
            int32_t *row = output_map + (int64_t)h->vid[i] * W;
            row[1 + row[0]++] = i;
        }
    } else {
        for (int32_t v = 0; v < M; ++v) output_map[(int64_t)v * W] = 1;
        for (int32_t i = 0; i < h->n; ++i) {
            int32_t *row = output_map + (int64_t)h->vid[i] * W;
            if (h->mode == 2 || row[1] < 0) row[1] = i;  // mode 2 keeps the last, 0/1 the first
        }
    }
    for (int32_t v = 0; v < M; ++v) {
        const int64_t *src = coords + (int64_t)output_map[(int64_t)v * W + 1] * h->ncol;
        for (int j = 0; j < h->ncol; ++j) output_coords[(int64_t)v * h->ncol + j] = src[j];
    }
    delete h;
    return DODA_OK;
}

extern "C" void doda_voxelize_idx_free_h(void *handle) { delete (HostVox *)handle; }

// ================================= device ===============================================
namespace {
constexpr int VX_EMPTY = 0x7fffffff;

struct VoxWs {
    int32_t *tab; uint32_t cap;
    int32_t *first, *flag, *rank, *count, *cursor, *last, *scan;
    size_t total;
};

VoxWs vox_carve(void *ws, int n) {
    VoxWs w;
    const int nn = n > 0 ? n : 1;
    w.cap = next_pow2((uint32_t)(2 * nn < 1024 ? 1024 : 2 * nn));
    char *p = (char *)ws;
    size_t off = 0;
    w.tab = (int32_t *)(p + off); off += (size_t)w.cap * 4;
    const size_t ni = align_up((size_t)nn * 4, 256);
    w.first = (int32_t *)(p + off); off += ni;
    w.flag = (int32_t *)(p + off); off += ni;
    w.rank = (int32_t *)(p + off); off += ni;
    w.count = (int32_t *)(p + off); off += ni;
    w.cursor = (int32_t *)(p + off); off += ni;
    w.last = (int32_t *)(p + off); off += ni;
    w.scan = (int32_t *)(p + off); off += align_up(scan_ws_ints(nn) * 4, 256);
    w.total = off;
    return w;
}

__device__ __forceinline__ void load_key(const int64_t *__restrict__ coords, int i, int ncol,
                                         int &b, int &x, int &y, int &z) {
    const int64_t *c = coords + (long long)i * ncol;
    if (ncol == 4) { b = (int)c[0]; x = (int)c[1]; y = (int)c[2]; z = (int)c[3]; }
    else { b = 0; x = (int)c[0]; y = (int)c[1]; z = (int)c[2]; }
}

__device__ __forceinline__ uint32_t key_hash(int b, int x, int y, int z) {
    return hash_mix((uint32_t)x * 73856093u ^ hash_mix((uint32_t)y * 19349663u ^ hash_mix((uint32_t)z * 83492791u ^ (uint32_t)b)));
}

__global__ __launch_bounds__(256) void vox_insert(const int64_t *__restrict__ coords, int n,
                                                  int ncol, int32_t *tab, uint32_t mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b, x, y, z;
    load_key(coords, i, ncol, b, x, y, z);
    uint32_t s = key_hash(b, x, y, z) & mask;
    for (;;) {
        const int old = atomicCAS(&tab[s], VX_EMPTY, i);
        if (old == VX_EMPTY) return;
        int ob, ox, oy, oz;
        load_key(coords, old, ncol, ob, ox, oy, oz);
        if (ob == b && ox == x && oy == y && oz == z) {
            atomicMin(&tab[s], i);  // slot keeps the smallest index of this voxel
            return;
        }
        s = (s + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void vox_first(const int64_t *__restrict__ coords, int n,
                                                 int ncol, const int32_t *__restrict__ tab,
                                                 uint32_t mask, int32_t *__restrict__ first,
                                                 int32_t *__restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b, x, y, z;
    load_key(coords, i, ncol, b, x, y, z);
    uint32_t s = key_hash(b, x, y, z) & mask;
    int f;
    for (;;) {
        f = tab[s];
        int ob, ox, oy, oz;
        load_key(coords, f, ncol, ob, ox, oy, oz);
        if (ob == b && ox == x && oy == y && oz == z) break;
        s = (s + 1) & mask;
    }
    first[i] = f;
    flag[i] = f == i;
}

__global__ __launch_bounds__(256) void vox_assign(int n, const int32_t *__restrict__ first,
                                                  const int32_t *__restrict__ rank,
                                                  int32_t *__restrict__ input_map,
                                                  int32_t *__restrict__ count,
                                                  int32_t *__restrict__ last) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = rank[first[i]];
    input_map[i] = v;
    atomicAdd(&count[v], 1);
    atomicMax(&last[v], i);
}

__global__ __launch_bounds__(256) void vox_max(const int32_t *__restrict__ count,
                                               const int32_t *__restrict__ n_active_d, int mode,
                                               int32_t *__restrict__ counts_out) {
    const int m = *n_active_d;
    int mx = 1;
    if (mode == 3 || mode == 4)
        for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < m; v += gridDim.x * blockDim.x)
            mx = count[v] > mx ? count[v] : mx;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
    }
    // one atomic per workgroup, and only when it would raise the value (2048 waves on one address took 25 us)
    __shared__ int wmx[4];
    if (lane_id() == 0) wmx[threadIdx.x >> 6] = mx;
    doda_sync();
    if (threadIdx.x == 0) {
        const int a = wmx[0] > wmx[1] ? wmx[0] : wmx[1], b = wmx[2] > wmx[3] ? wmx[2] : wmx[3];
        const int bm = a > b ? a : b;
        if (bm > __hip_atomic_load(&counts_out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&counts_out[1], bm);
    }
}

// n_active / max_active are the caller's (doda_voxelize_idx_fill): every kernel below bounds its reads of the workspace
// arrays (n entries) and its writes of the outputs (n_active rows of W) by them, so a size that disagrees with what
// doda_voxelize_idx_assign counted cannot touch memory outside the two outputs — an underestimate loses the voxels and
// points that do not fit, an overestimate leaves trailing rows empty (count 0 / index -1, coordinates 0).
__global__ __launch_bounds__(256) void vox_rows_init(int n, int m, int W, int mode,
                                                     const int32_t *__restrict__ count,
                                                     int32_t *__restrict__ output_map) {
    const long long total = (long long)m * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(e / W), c = (int)(e - (long long)v * W);
        const int cnt = v < n ? count[v] : 0;   // 0 for rows past the voxels stage 1 found
        output_map[e] = c == 0 ? ((mode == 3 || mode == 4) ? (cnt < W ? cnt : W - 1) : (cnt > 0 ? 1 : 0)) : -1;
    }
}

__global__ __launch_bounds__(256) void vox_scatter(int n, int m, int W, int mode,
                                                   const int32_t *__restrict__ first,
                                                   const int32_t *__restrict__ rank,
                                                   const int32_t *__restrict__ last,
                                                   int32_t *__restrict__ cursor,
                                                   int32_t *__restrict__ output_map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = rank[first[i]];  // voxel of point i (kept in the workspace by stage 1)
    if (v >= m) return;
    int32_t *row = output_map + (long long)v * W;
    if (mode == 3 || mode == 4) {
        const int slot = atomicAdd(&cursor[v], 1);
        if (slot < W - 1) row[1 + slot] = i;  // any order; vox_finish sorts the row
    } else if (first[i] == i) {
        row[1] = (mode == 2) ? last[v] : i;
    }
}

// ascending point order inside each row + the voxel's coordinate row
__global__ __launch_bounds__(256) void vox_finish(int m, int W, int ncol, int mode,
                                                  const int64_t *__restrict__ coords,
                                                  int32_t *__restrict__ output_map,
                                                  int64_t *__restrict__ output_coords) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m) return;
    int32_t *row = output_map + (long long)v * W;
    if (mode == 3 || mode == 4) {
        const int cnt = row[0];
        for (int a = 2; a <= cnt; ++a) {
            const int key = row[a];
            int b = a - 1;
            while (b >= 1 && row[b] > key) { row[b + 1] = row[b]; --b; }
            row[b + 1] = key;
        }
    }
    const int p = row[1];
    for (int j = 0; j < ncol; ++j) output_coords[(long long)v * ncol + j] = p >= 0 ? coords[(long long)p * ncol + j] : 0;
}
}  // namespace

extern "C" size_t doda_voxelize_idx_workspace_bytes(int32_t n) { return vox_carve(nullptr, n).total; }

extern "C" int doda_voxelize_idx_assign(const int64_t *coords, int32_t n, int32_t ncol,
                                        int32_t mode, int32_t *input_map, int32_t *counts_out,
                                        void *ws, size_t ws_bytes, doda_stream_t stream) {
    if (n < 0 || (ncol != 3 && ncol != 4) || mode < 0 || mode > 4 || !counts_out) return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    hipMemsetAsync(counts_out, 0, 2 * sizeof(int32_t), s);
    if (n == 0) return DODA_OK;
    if (!coords || !input_map || !ws) return DODA_ERR_INVALID;
    const VoxWs w = vox_carve(ws, n);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    const int grid = div_up(n, 256);
    hipMemsetD32Async((hipDeviceptr_t)w.tab, VX_EMPTY, w.cap, s);
    hipMemsetAsync(w.count, 0, (size_t)((char *)w.last - (char *)w.count) + (size_t)n * 4, s);   // count, cursor, last: adjacent (vox_carve)
    hipLaunchKernelGGL(vox_insert, dim3(grid), dim3(256), 0, s, coords, n, ncol, w.tab, w.cap - 1);
    hipLaunchKernelGGL(vox_first, dim3(grid), dim3(256), 0, s, coords, n, ncol, w.tab, w.cap - 1,
                       w.first, w.flag);
    int st = exclusive_scan_i32(w.flag, w.rank, n, counts_out, w.scan, s);
    if (st != DODA_OK) return st;
    hipLaunchKernelGGL(vox_assign, dim3(grid), dim3(256), 0, s, n, w.first, w.rank, input_map,
                       w.count, w.last);
    hipLaunchKernelGGL(vox_max, dim3(div_up(n, 256) < 512 ? div_up(n, 256) : 512), dim3(256), 0, s,
                       w.count, counts_out, mode, counts_out);
    return doda_check_launch();
}

extern "C" int doda_voxelize_idx_fill(const int64_t *coords, int32_t n, int32_t ncol, int32_t mode,
                                      int32_t n_active, int32_t max_active,
                                      int64_t *output_coords, int32_t *output_map, void *ws,
                                      size_t ws_bytes, doda_stream_t stream) {
    if (n < 0 || n_active < 0 || max_active < 1 || (ncol != 3 && ncol != 4)) return DODA_ERR_INVALID;
    if (n == 0 || n_active == 0) return DODA_OK;
    if (!coords || !output_coords || !output_map || !ws) return DODA_ERR_INVALID;
    const VoxWs w = vox_carve(ws, n);
    if (ws_bytes < w.total) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const int W = max_active + 1;
    const long long total = (long long)n_active * W;
    const int g0 = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(vox_rows_init, dim3(g0), dim3(256), 0, s, n, n_active, W, mode, w.count,
                       output_map);
    hipLaunchKernelGGL(vox_scatter, dim3(div_up(n, 256)), dim3(256), 0, s, n, n_active, W, mode, w.first,
                       w.rank, w.last, w.cursor, output_map);
    hipLaunchKernelGGL(vox_finish, dim3(div_up(n_active, 256)), dim3(256), 0, s, n_active, W, ncol,
                       mode, coords, output_map, output_coords);
    return doda_check_launch();
}
