// Point head + loss without the point-level score matrix (ABI 11: doda_head_ce_fwd / _bwd).
//
// reference model/unet.py:62-64,107-108,196: feats_pt = out.features[p2v]; scores = Linear(feats_pt); loss =
// CrossEntropyLoss(ignore_index)(scores, labels) — a [N points, n_cls] fp32 matrix (62 MB at 781 k points x 20 classes) that the
// per-layer path writes and re-reads five times (gather-GEMM, loss forward, loss backward, cast + column sums, the two
// gradient GEMMs: ~210 us of a 4.8 ms step).  Every point of a voxel reads the SAME feature row, so its logits are its voxel's:
// with z_v = W f_v + b,  lse_v = log sum exp z_v  and the voxel's point list (v2p_map, reference lib/pointgroup_ops voxelize_idx)
//     loss   = (1 / n_valid) sum_v [ cnt_v lse_v - sum_{p in v, valid} z_v[y_p] ]
//     dz_v   = (g / n_valid) [ cnt_v softmax(z_v) - hist_v ]          (hist_v[k] = valid points of v labelled k)
//     df_v   = W^T dz_v,   dW = sum_v dz_v f_v^T,   db = sum_v dz_v
// i.e. two sweeps over the VOXEL rows (19 MB of features + the point lists + the labels): one thread per voxel, the weights
// in LDS, logits recomputed instead of stored.  dz_v is written once at voxel level (the operand of the library's weight-gradient
// kernel over an identity table); db as per-workgroup rows the caller adds (fixed order: deterministic).  The per-voxel argmax
// (the training meters' prediction, reference tool/train.py accuracy) rides in the forward sweep.
// bf16 features: the weights are rounded to bf16 first, as the gather-GEMM's pre-pack does; accumulation in fp32.
#include "common.hpp"
#include "spconv_common.hpp"

namespace {
constexpr int HD_BLOCK = 256;
constexpr int HD_MAX_C = 32;      // feature channels (DODA: 16)
constexpr int HD_MAX_K = 32;      // classes

template <int ESZ> struct HdRow;
template <> struct HdRow<2> {
    static __device__ __forceinline__ void load4(const void *p, float (&v)[4]) {
        const u32x2 r = *reinterpret_cast<const u32x2 *>(p);
        v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
        v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(void *p, float v) { *reinterpret_cast<unsigned short *>(p) = f2bf(v); }
    static __device__ __forceinline__ float load(const void *p) { return __uint_as_float((unsigned)*reinterpret_cast<const unsigned short *>(p) << 16); }
    static __device__ __forceinline__ float wround(float w) { return __uint_as_float((unsigned)f2bf(w) << 16); }
};
template <> struct HdRow<4> {
    static __device__ __forceinline__ void load4(const void *p, float (&v)[4]) {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    static __device__ __forceinline__ void store(void *p, float v) { *reinterpret_cast<float *>(p) = v; }
    static __device__ __forceinline__ float load(const void *p) { return *reinterpret_cast<const float *>(p); }
    static __device__ __forceinline__ float wround(float w) { return w; }
};

template <int ESZ, int C>
__device__ __forceinline__ void hd_load_row(const void *feats, long long v, float (&f)[C]) {
#pragma unroll
    for (int q = 0; q < C; q += 4) {
        float t[4];
        HdRow<ESZ>::load4((const char *)feats + ((size_t)v * C + q) * ESZ, t);
        f[q] = t[0]; f[q + 1] = t[1]; f[q + 2] = t[2]; f[q + 3] = t[3];
    }
}
template <int C>
__device__ __forceinline__ float hd_logit(const float (*w)[HD_MAX_C], const float *b, int k, const float (&f)[C]) {
    float z = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) z = __builtin_fmaf(w[k][c], f[c], z);
    return z + b[k];
}
template <int ESZ>
__device__ __forceinline__ void hd_stage_weights(const float *__restrict__ weight, const float *__restrict__ bias, int n_cls, int c,
                                                 float (*w)[HD_MAX_C], float *b) {
    for (int e = threadIdx.x; e < n_cls * c; e += HD_BLOCK) w[e / c][e % c] = HdRow<ESZ>::wround(weight[e]);
    for (int k = threadIdx.x; k < n_cls; k += HD_BLOCK) b[k] = bias ? bias[k] : 0.f;
    doda_sync();
}

// NK: the class count rounded up to a multiple of four (compile time): the voxel's logits live in registers — computed once per
// sweep (three recomputations per class, with the class count a run-time bound, took 31 us forward / 121 us backward at 601 k
// voxels) — with the padding classes at -inf.
template <int C, int NK>
__device__ __forceinline__ void hd_logits(const float (*w)[HD_MAX_C], const float *b, int n_cls, const float (&f)[C], float (&z)[NK], float &mx, int &arg) {
    mx = -INFINITY;
    arg = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) t = __builtin_fmaf(w[k][c], f[c], t);     // (fused: the build's -ffp-contract=off would make it two instructions)
        t = k < n_cls ? t + b[k] : -INFINITY;
        z[k] = t;
        if (t > mx) { mx = t; arg = k; }
        if ((k & 3) == 3) asm volatile("" ::: "memory");
    }
}

// forward: per-workgroup (loss sum, valid count) partials; pred[v] = argmax_k z_v[k]
template <int ESZ, int C, int NK>
__global__ __launch_bounds__(HD_BLOCK, 4) void head_ce_fwd(const void *__restrict__ feats, int m, const float *__restrict__ weight,
                                                        const float *__restrict__ bias, int n_cls, const int32_t *__restrict__ v2p,
                                                        int v2p_ld, const long long *__restrict__ labels, long long ignore_index,
                                                        float *__restrict__ partial, int32_t *__restrict__ pred) {
    __shared__ float w[HD_MAX_K][HD_MAX_C], b[HD_MAX_K];
    __shared__ float red[2][HD_BLOCK / 64];
    hd_stage_weights<ESZ>(weight, bias, n_cls, C, w, b);
    float loss = 0.f, cnt = 0.f;
#pragma unroll 1
    for (long long v = (long long)blockIdx.x * HD_BLOCK + threadIdx.x; v < m; v += (long long)gridDim.x * HD_BLOCK) {
        asm volatile("" ::: "memory");      // (the 320 staged weights stay in LDS: hoisted out of this loop they cost 256 VGPRs + spills)
        float f[C], z[NK], mx;
        int arg;
        hd_load_row<ESZ, C>(feats, v, f);
        const int32_t *row = v2p + v * v2p_ld;
        const int np = row[0];
        hd_logits<C, NK>(w, b, n_cls, f, z, mx, arg);
        if (pred) pred[v] = arg;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) s += __expf(z[k] - mx);
        const float lse = mx + __logf(s);
#pragma unroll 1
        for (int i = 0; i < np; ++i) {
            const long long lab = labels[row[1 + i]];
            if (lab != ignore_index && lab >= 0 && lab < n_cls) {
                float zy = 0.f;
#pragma unroll
                for (int k = 0; k < NK; ++k) zy = (int)lab == k ? z[k] : zy;
                loss += lse - zy;
                cnt += 1.f;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { loss += __shfl_xor(loss, d, 64); cnt += __shfl_xor(cnt, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
    doda_sync();
    if (threadIdx.x == 0) {
        float a = 0.f, c2 = 0.f;
        for (int q = 0; q < HD_BLOCK / 64; ++q) { a += red[0][q]; c2 += red[1][q]; }
        partial[2 * blockIdx.x] = a;
        partial[2 * blockIdx.x + 1] = c2;
    }
}

// out[0] = sum(loss) / max(n_valid, 1), out[1] = n_valid (fixed order, fp64).  1024 threads: two or three partial rows each — one
// round trip instead of the 37 dependent ones a single wave needed over 2349 rows (11 us)
__global__ __launch_bounds__(1024) void head_ce_final(const float *__restrict__ partial, int nblocks, float *__restrict__ out) {
    __shared__ double red[2][16];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 1024) { a += (double)partial[2 * k]; b += (double)partial[2 * k + 1]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    doda_sync();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int q = 0; q < 16; ++q) { a += red[0][q]; b += red[1][q]; }
        out[0] = (float)(a / (b > 1.0 ? b : 1.0));
        out[1] = (float)b;
    }
}

// backward: d_feats [m, C], dz [m, n_cls] (storage type of the features), db partial rows [blocks][n_cls].
// The per-class values of a thread's voxel (exp(z - max), then dz) and its running column sums live in LDS COLUMNS
// (sz[k][thread], sdb[k][thread]: conflict-free) instead of register arrays: the class loops stay rolled, a point's label indexes
// its class directly (dz[label] -= scale), and the kernel needs ~50 VGPRs — the register form (three 20-element arrays per thread,
// loops fully unrolled) compiled to 256 VGPRs + spills and ran at 120 us.
template <int ESZ, int C, int NK>
__global__ __launch_bounds__(HD_BLOCK) void head_ce_bwd(const void *__restrict__ feats, int m, const float *__restrict__ weight,
                                                        const float *__restrict__ bias, int n_cls, const int32_t *__restrict__ v2p,
                                                        int v2p_ld, const long long *__restrict__ labels, long long ignore_index,
                                                        const float *__restrict__ out, const float *__restrict__ grad,
                                                        void *__restrict__ d_feats, void *__restrict__ dz_out, float *__restrict__ db_partial) {
    __shared__ __attribute__((aligned(16))) float w[HD_MAX_K][HD_MAX_C], b[HD_MAX_K];
    __shared__ float sz[NK][HD_BLOCK];
    hd_stage_weights<ESZ>(weight, bias, n_cls, C, w, b);
    const float nv = out[1] > 1.f ? out[1] : 1.f;
    const float scale = grad[0] / nv;
    const int tid = threadIdx.x;
    const long long v = (long long)blockIdx.x * HD_BLOCK + tid;
#pragma unroll 1
    for (int k = 0; k < NK; ++k) sz[k][tid] = 0.f;      // (a thread past the last voxel contributes zeros to the column sums)
    if (v < m) {
        float f[C];
        hd_load_row<ESZ, C>(feats, v, f);
        const int32_t *row = v2p + v * v2p_ld;
        const int np = row[0];
        float mx = -INFINITY;
#pragma unroll 2
        for (int k = 0; k < n_cls; ++k) {
            const float z = hd_logit<C>(w, b, k, f);
            sz[k][tid] = z;
            mx = fmaxf(mx, z);
        }
        float s = 0.f;
#pragma unroll 2
        for (int k = 0; k < n_cls; ++k) {
            const float e = __expf(sz[k][tid] - mx);
            sz[k][tid] = e;
            s += e;
        }
        // valid points: their count now, their classes after the softmax term
        float cntv = 0.f;
#pragma unroll 1
        for (int i = 0; i < np; ++i) {
            const long long lab = labels[row[1 + i]];
            cntv += (lab != ignore_index && lab >= 0 && lab < n_cls) ? 1.f : 0.f;
        }
        const float a = cntv * scale / s;        // dz_k = a exp(z_k - mx) - scale hist_k
#pragma unroll 2
        for (int k = 0; k < n_cls; ++k) sz[k][tid] *= a;
#pragma unroll 1
        for (int i = 0; i < np; ++i) {
            const long long lab = labels[row[1 + i]];
            if (lab != ignore_index && lab >= 0 && lab < n_cls) sz[(int)lab][tid] -= scale;
        }
        float df[C];
#pragma unroll
        for (int c = 0; c < C; ++c) df[c] = 0.f;
        char *dzr = (char *)dz_out + (size_t)v * n_cls * ESZ;
#pragma unroll 1
        for (int k = 0; k < n_cls; k += 2) {      // two classes per step: one 4- / 8-byte store of dz
            const float g0 = sz[k][tid], g1 = k + 1 < n_cls ? sz[k + 1][tid] : 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) df[c] = __builtin_fmaf(g0, w[k][c], df[c]);
            if (k + 1 < n_cls) {
#pragma unroll
                for (int c = 0; c < C; ++c) df[c] = __builtin_fmaf(g1, w[k + 1][c], df[c]);
                if constexpr (ESZ == 2)
                    *reinterpret_cast<unsigned *>(dzr + (size_t)k * 2) = (unsigned)f2bf(g0) | ((unsigned)f2bf(g1) << 16);
                else
                    *reinterpret_cast<f32x2 *>(dzr + (size_t)k * 4) = (f32x2){g0, g1};
            } else HdRow<ESZ>::store(dzr + (size_t)k * ESZ, g0);
        }
        char *dfr = (char *)d_feats + (size_t)v * C * ESZ;
        if constexpr (ESZ == 2) {
#pragma unroll
            for (int q = 0; q < C; q += 8) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(df[q + 2 * j]) | ((unsigned)f2bf(df[q + 2 * j + 1]) << 16);
                *reinterpret_cast<u32x4 *>(dfr + (size_t)q * 2) = o;
            }
        } else {
#pragma unroll
            for (int q = 0; q < C; q += 4) *reinterpret_cast<f32x4 *>(dfr + (size_t)q * 4) = (f32x4){df[q], df[q + 1], df[q + 2], df[q + 3]};
        }
    }
    // column sums of the workgroup's 256 voxels: 8 threads per class over 32 columns each, then the eight in order (fixed order)
    doda_sync();
    {
        const int k = tid >> 3, part = tid & 7;
        float t = 0.f;
        if (k < n_cls) {
            for (int q = 0; q < 32; ++q) t += sz[k][part * 32 + q];
        }
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        if (k < n_cls && part == 0) db_partial[(size_t)blockIdx.x * n_cls + k] = t;
    }
}

// dW of the head from the voxel-level score gradient, bf16: partial[wg][32 classes][16 channels] = sum over the workgroup's voxels of
// dz_v[k] f_v[c].  Voxels are the MFMA's k dimension: a wave parks 64 voxels' dz / feature rows in LDS class- / channel-major
// (lane = voxel writes a column), so that lane (row i, group g) of v_mfma_f32_16x16x32_bf16 reads its eight consecutive voxels of
// class / channel i with one 16-byte read; 4 MFMAs per 64 voxels, accumulators across the wave's chunks, the four waves summed in
// order.  Replaces the gather-table weight-gradient kernel over an identity table (50 us for this shape: 20 is not a multiple of
// its 16-channel blocks).
__global__ __launch_bounds__(HD_BLOCK) void head_dw_bf16(const unsigned short *__restrict__ feats, const unsigned short *__restrict__ dz,
                                                         int m, int n_cls, float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned short sdz[4][32][64 + 8], sf[4][16][64 + 8];   // (+8: rows 144 bytes apart, 16-byte aligned)
    __shared__ f32x4 red[3][2][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k = n_cls; k < 32; ++k) sdz[wid][k][lane] = 0;                  // padding classes: zero rows, written once
    const long long n_chunks = ((long long)m + 63) / 64;
    for (long long ch = (long long)blockIdx.x * 4 + wid; ch < n_chunks; ch += (long long)gridDim.x * 4) {
        const long long v = ch * 64 + lane;
        if (v < m) {
            const u32x4 f0 = *reinterpret_cast<const u32x4 *>(feats + v * 16), f1 = *reinterpret_cast<const u32x4 *>(feats + v * 16 + 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sf[wid][2 * q][lane] = (unsigned short)(f0[q] & 0xffffu); sf[wid][2 * q + 1][lane] = (unsigned short)(f0[q] >> 16);
                sf[wid][8 + 2 * q][lane] = (unsigned short)(f1[q] & 0xffffu); sf[wid][9 + 2 * q][lane] = (unsigned short)(f1[q] >> 16);
            }
            const unsigned short *r = dz + v * n_cls;
            for (int k = 0; k + 1 < n_cls; k += 2) {      // (rows of n_cls bf16: 4-byte aligned when n_cls is even; odd: element by element)
                if ((n_cls & 1) == 0) {
                    const unsigned p = *reinterpret_cast<const unsigned *>(r + k);
                    sdz[wid][k][lane] = (unsigned short)(p & 0xffffu); sdz[wid][k + 1][lane] = (unsigned short)(p >> 16);
                } else { sdz[wid][k][lane] = r[k]; sdz[wid][k + 1][lane] = r[k + 1]; }
            }
            if (n_cls & 1) sdz[wid][n_cls - 1][lane] = r[n_cls - 1];
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) sf[wid][c][lane] = 0;
            for (int k = 0; k < n_cls; ++k) sdz[wid][k][lane] = 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (wave-private LDS: its own writes are visible to its own reads after the wait)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const u32x4 bv = *reinterpret_cast<const u32x4 *>(&sf[wid][i][32 * st + 8 * g]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 av = *reinterpret_cast<const u32x4 *>(&sdz[wid][16 * t + i][32 * st + 8 * g]);
                mma_bf16_k32(acc[t], av, bv);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // D[row = class 16 t + 4 g + r][col = channel i]; the four waves in order
    if (wid > 0) { red[wid - 1][0][lane] = acc[0]; red[wid - 1][1][lane] = acc[1]; }
    doda_sync();
    if (wid == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 a = acc[t];
#pragma unroll
            for (int q = 0; q < 3; ++q) a += red[q][t][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) partial[(size_t)blockIdx.x * 512 + (size_t)(16 * t + 4 * g + r) * 16 + i] = a[r];
        }
    }
}

inline int hd_blocks(int m) {
    const int nb = div_up(m, HD_BLOCK);    // one voxel per thread: three dependent global reads per voxel (point list -> point ids ->
    return nb < 1 ? 1 : nb;                // labels) want every wave slot of the chip filled (1024 fatter workgroups: 24 / 53 us)
}
// class counts as compile-time multiples of four (DODA: 20 ScanNet / 13 S3DIS / 11 common classes); others: DODA_ERR_UNSUPPORTED
#define HD_DISPATCH(KERNEL, ...)                                                                                   \
    do {                                                                                                           \
        const int nk = (n_cls + 3) / 4 * 4;                                                                        \
        if (elem_bytes == 2) {                                                                                     \
            if (nk <= 12) hipLaunchKernelGGL((KERNEL<2, 16, 12>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);       \
            else if (nk <= 16) hipLaunchKernelGGL((KERNEL<2, 16, 16>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);  \
            else if (nk <= 20) hipLaunchKernelGGL((KERNEL<2, 16, 20>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);  \
            else hipLaunchKernelGGL((KERNEL<2, 16, 32>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);                \
        } else {                                                                                                   \
            if (nk <= 12) hipLaunchKernelGGL((KERNEL<4, 16, 12>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);       \
            else if (nk <= 16) hipLaunchKernelGGL((KERNEL<4, 16, 16>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);  \
            else if (nk <= 20) hipLaunchKernelGGL((KERNEL<4, 16, 20>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);  \
            else hipLaunchKernelGGL((KERNEL<4, 16, 32>), dim3(n_blocks), dim3(HD_BLOCK), 0, s, __VA_ARGS__);                \
        }                                                                                                          \
    } while (0)
inline bool hd_bad(const void *feats, int m, int c, int esz, const float *weight, int n_cls, const int32_t *v2p, int v2p_ld,
                   const int64_t *labels) {
    return m < 0 || (esz != 2 && esz != 4) || n_cls <= 0 || v2p_ld < 1 || !feats || !weight || !v2p || !labels;
}
}  // namespace

extern "C" int32_t doda_head_ce_blocks(int32_t m) { return hd_blocks(m > 0 ? m : 1); }

extern "C" int doda_head_ce_fwd(const void *feats, int32_t m, int32_t c, int32_t elem_bytes, const float *weight, const float *bias,
                                int32_t n_cls, const int32_t *v2p, int32_t v2p_ld, const int64_t *labels, int64_t ignore_index,
                                float *out, int32_t *pred, float *partial_ws, int32_t n_blocks, doda_stream_t stream) {
    if (!out) return DODA_ERR_INVALID;
    hipStream_t s = as_stream(stream);
    if (m == 0) { (void)hipMemsetAsync(out, 0, 8, s); return DODA_OK; }
    if (hd_bad(feats, m, c, elem_bytes, weight, n_cls, v2p, v2p_ld, labels) || !partial_ws) return DODA_ERR_INVALID;
    if (c != 16 || n_cls > 32) return DODA_ERR_UNSUPPORTED;     // (DODA's head: 16 channels, <= 20 classes; others take the matrix path)
    if (n_blocks != hd_blocks(m)) return DODA_ERR_WORKSPACE;
    HD_DISPATCH(head_ce_fwd, feats, m, weight, bias, n_cls, v2p, v2p_ld, (const long long *)labels, (long long)ignore_index, partial_ws, pred);
    hipLaunchKernelGGL(head_ce_final, dim3(1), dim3(1024), 0, s, (const float *)partial_ws, n_blocks, out);
    return doda_check_launch();
}

extern "C" int doda_head_ce_bwd(const void *feats, int32_t m, int32_t c, int32_t elem_bytes, const float *weight, const float *bias,
                                int32_t n_cls, const int32_t *v2p, int32_t v2p_ld, const int64_t *labels, int64_t ignore_index,
                                const float *out, const float *grad, void *d_feats, void *dz, float *db_partial, int32_t n_blocks,
                                doda_stream_t stream) {
    if (m == 0) return DODA_OK;
    if (hd_bad(feats, m, c, elem_bytes, weight, n_cls, v2p, v2p_ld, labels) || !out || !grad || !d_feats || !dz || !db_partial)
        return DODA_ERR_INVALID;
    if (c != 16 || n_cls > 32) return DODA_ERR_UNSUPPORTED;
    if (n_blocks != hd_blocks(m)) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    HD_DISPATCH(head_ce_bwd, feats, m, weight, bias, n_cls, v2p, v2p_ld, (const long long *)labels, (long long)ignore_index, out, grad,
                d_feats, dz, db_partial);
    return doda_check_launch();
}

extern "C" int32_t doda_head_dw_blocks(int32_t m) {
    int nb = div_up(div_up(m > 0 ? m : 1, 64), 4);
    if (nb > 1024) nb = 1024;
    return nb < 1 ? 1 : nb;
}

extern "C" int doda_head_dw_bf16(const void *feats, const void *dz, int32_t m, int32_t c, int32_t n_cls, float *partial, int32_t n_blocks,
                                 doda_stream_t stream) {
    if (m < 0 || !partial) return DODA_ERR_INVALID;
    if (c != 16 || n_cls < 1 || n_cls > 32) return DODA_ERR_UNSUPPORTED;
    if (n_blocks != doda_head_dw_blocks(m)) return DODA_ERR_WORKSPACE;
    if (m > 0 && (!feats || !dz || ((uintptr_t)feats & 15) || ((uintptr_t)dz & 3))) return DODA_ERR_INVALID;
    hipLaunchKernelGGL(head_dw_bf16, dim3(n_blocks), dim3(HD_BLOCK), 0, as_stream(stream), (const unsigned short *)feats,
                       (const unsigned short *)dz, m, n_cls, partial);
    return doda_check_launch();
}
