// SGD update of every parameter of a network in ONE launch.
//
// The reference trains with torch.optim.SGD (util/common_utils.py:196-215, tool/train.py:235-268); the
// U-Net has 281 parameter tensors, most of them BatchNorm vectors of 16..224 floats.  torch's fused
// multi-tensor path packs at most ~110 tensor addresses into one kernel-argument block, so the step is
// five launches of ~30-50 us each; here the per-tensor descriptors live in device memory (uploaded with the
// launch: the gradient tensors are new allocations every step) and one grid covers all tensors.
// The arithmetic is torch's FusedSgdMathFunctor term by term (hyper-parameters are doubles there, so the
// mixed products round once, from double): bit-identical parameters and momentum buffers.
#include "common.hpp"

#include <mutex>
#include <string.h>

namespace {

constexpr int SGD_BLOCK = 256;
constexpr int SGD_PER_THREAD = 8;                       // two 16-byte vectors per thread
constexpr int SGD_TILE = SGD_BLOCK * SGD_PER_THREAD;    // elements per block

struct SgdDesc {            // device descriptor (32 bytes)
    float *p;
    const float *g;
    float *buf;             // momentum buffer or null
    int n;                  // elements
    int blk_end;            // inclusive prefix of blocks; bit 31 of `n` is not used
};

// torch's kernel is compiled with floating-point contraction, this library without (the voxel / kNN
// kernels must round every operation like the reference's): the fused multiply-adds are spelled out.
// The new buffer value is a double there; only the Nesterov term sees it unrounded (tools/sgdprobe.py).
__device__ __forceinline__ void sgd_one(float &p, float g, float *buf, bool first, double lr, double momentum,
                                        double dampening, double wd, bool nesterov, bool maximize) {
    if (maximize) g = -g;
    if (wd != 0.0) g = (float)fma(wd, (double)p, (double)g);
    if (buf) {
        const double b = first ? (double)g : fma(momentum, (double)*buf, (1.0 - dampening) * (double)g);
        *buf = (float)b;
        g = nesterov ? (float)fma(momentum, b, (double)g) : (float)b;
    }
    p = (float)fma(-lr, (double)g, (double)p);
}

__global__ __launch_bounds__(SGD_BLOCK) void sgd_multi(const SgdDesc *__restrict__ descs, const int *__restrict__ first,
                                                       int n_desc, double lr, double momentum, double dampening,
                                                       double wd, int nesterov, int maximize) {
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {   // first descriptor whose inclusive end exceeds this block
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x < descs[mid].blk_end) hi = mid; else lo = mid + 1;
    }
    const SgdDesc d = descs[lo];
    const bool is_first = first[lo] != 0;
    const int blk = (int)blockIdx.x - (lo == 0 ? 0 : descs[lo - 1].blk_end);
    const long long base = (long long)blk * SGD_TILE;
    const bool vec = (((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.buf) & 15) == 0;
#pragma unroll
    for (int k = 0; k < SGD_PER_THREAD / 4; ++k) {
        const long long e = base + ((long long)k * SGD_BLOCK + threadIdx.x) * 4;
        if (e >= d.n) break;
        if (vec && e + 3 < d.n) {
            float4 p = *reinterpret_cast<float4 *>(d.p + e);
            const float4 g = *reinterpret_cast<const float4 *>(d.g + e);
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.buf && !is_first) b = *reinterpret_cast<const float4 *>(d.buf + e);
            sgd_one(p.x, g.x, d.buf ? &b.x : nullptr, is_first, lr, momentum, dampening, wd, nesterov, maximize);
            sgd_one(p.y, g.y, d.buf ? &b.y : nullptr, is_first, lr, momentum, dampening, wd, nesterov, maximize);
            sgd_one(p.z, g.z, d.buf ? &b.z : nullptr, is_first, lr, momentum, dampening, wd, nesterov, maximize);
            sgd_one(p.w, g.w, d.buf ? &b.w : nullptr, is_first, lr, momentum, dampening, wd, nesterov, maximize);
            *reinterpret_cast<float4 *>(d.p + e) = p;
            if (d.buf) *reinterpret_cast<float4 *>(d.buf + e) = b;
        } else {
            for (int q = 0; q < 4 && e + q < d.n; ++q) {
                float p = d.p[e + q];
                float b = (d.buf && !is_first) ? d.buf[e + q] : 0.f;
                sgd_one(p, d.g[e + q], d.buf ? &b : nullptr, is_first, lr, momentum, dampening, wd, nesterov, maximize);
                d.p[e + q] = p;
                if (d.buf) d.buf[e + q] = b;
            }
        }
    }
}

// pinned staging ring for the descriptor upload (grow-only slots, reuse of a slot guarded by its event)
struct Slot {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    int ev_dev = -1;    // device the event was created on (an event can only be recorded on that device's streams)
    bool pending = false;
};
std::mutex g_mu;
Slot g_slot[4];
unsigned g_next = 0;

}  // namespace

extern "C" size_t doda_sgd_multi_desc_bytes(int32_t n_tensors) {
    if (n_tensors <= 0) return 0;
    return align_up((size_t)n_tensors * sizeof(SgdDesc), 16) + align_up((size_t)n_tensors * sizeof(int), 16);
}

extern "C" int doda_sgd_multi(const doda_sgd_tensor *t, int32_t n_tensors, double lr, double momentum, double dampening,
                              double weight_decay, int32_t nesterov, int32_t maximize, void *desc_dev,
                              size_t desc_bytes, doda_stream_t stream) {
    if (n_tensors == 0) return DODA_OK;
    if (n_tensors < 0 || !t || !desc_dev) return DODA_ERR_INVALID;
    const size_t need = doda_sgd_multi_desc_bytes(n_tensors);
    if (desc_bytes < need) return DODA_ERR_WORKSPACE;
    hipStream_t s = as_stream(stream);
    const size_t first_off = align_up((size_t)n_tensors * sizeof(SgdDesc), 16);
    long long blocks = 0;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        Slot &sl = g_slot[g_next++ & 3u];
        if (sl.pending) { hipEventSynchronize(sl.ev); sl.pending = false; }
        if (sl.cap < need) {
            if (sl.host) hipHostFree(sl.host);
            sl.cap = align_up(need, 4096) * 2;
            if (hipHostMalloc(&sl.host, sl.cap, hipHostMallocDefault) != hipSuccess) { sl.host = nullptr; sl.cap = 0; return DODA_ERR_NOMEM; }
        }
        int cur_dev = 0;
        hipGetDevice(&cur_dev);
        if (sl.ev && sl.ev_dev != cur_dev) { hipEventDestroy(sl.ev); sl.ev = nullptr; }   // library used from another device
        if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return DODA_ERR_LAUNCH;
        sl.ev_dev = cur_dev;
        SgdDesc *d = (SgdDesc *)sl.host;
        int *first = (int *)((char *)sl.host + first_off);
        for (int k = 0; k < n_tensors; ++k) {
            if (!t[k].p || !t[k].g || t[k].n < 0 || t[k].n > 0x7fffffffll) return DODA_ERR_INVALID;
            if (momentum != 0.0 && !t[k].buf) return DODA_ERR_INVALID;
            blocks += (t[k].n + SGD_TILE - 1) / SGD_TILE;
            if (blocks > 0x7fffffffll) return DODA_ERR_UNSUPPORTED;
            d[k].p = t[k].p; d[k].g = t[k].g; d[k].buf = momentum != 0.0 ? t[k].buf : nullptr;
            d[k].n = (int)t[k].n; d[k].blk_end = (int)blocks;
            first[k] = t[k].first_step ? 1 : 0;
        }
        if (blocks == 0) return DODA_OK;
        if (hipMemcpyAsync(desc_dev, sl.host, need, hipMemcpyHostToDevice, s) != hipSuccess) return DODA_ERR_LAUNCH;
        hipEventRecord(sl.ev, s);
        sl.pending = true;
    }
    hipLaunchKernelGGL(sgd_multi, dim3((unsigned)blocks), dim3(SGD_BLOCK), 0, s, (const SgdDesc *)desc_dev,
                       (const int *)((const char *)desc_dev + first_off), n_tensors, lr, momentum, dampening,
                       weight_decay, nesterov, maximize);
    return doda_check_launch();
}
