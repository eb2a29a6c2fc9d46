// probe: what one phase boundary of a persistent coarse-level executor costs on MI355X.
// G participating workgroups (those whose block id has (b & 7) < XCDS of an 8 * G / XCDS grid: observed placement b % 8 =
// XCD) run PH phases; per phase every workgroup publishes WR bytes with 16-byte sc1 (write-through) stores, drains them,
// arrives on one counter (relaxed agent atomic), polls it with relaxed sc1 loads, then gathers RD bytes of OTHER
// workgroups' fresh data with 16-byte sc1 loads and checks every word.  Prints us per phase.
//   hipcc --offload-arch=gfx950 -O3 -o xcdbar xcdbar.hip && ./xcdbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target, unsigned *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned *ctr, unsigned *err, unsigned *xcc, char *buf, int G, int xcds, int phases,
                                              int wr_bytes, int rd_bytes, int slot_bytes, unsigned *bad) {
    if ((int)(blockIdx.x & 7) >= xcds) return;
    const int me = (blockIdx.x >> 3) * xcds + (blockIdx.x & 7);
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[me] = id;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (unsigned)(2 * G * slot_bytes), 0x00020000);
    unsigned nbad = 0;
    for (int p = 0; p < phases; ++p) {
        const unsigned base = (unsigned)((p & 1) * G * slot_bytes);   // ping-pong so a fast writer never overwrites a slow reader's data
        for (int off = threadIdx.x * 16; off < wr_bytes; off += THREADS * 16) {
            const unsigned v = (unsigned)(p * 65536 + me * 1024 + (off >> 4));
            const u32x4 d = {v, v + 1, v + 2, v + 3};
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, base + (unsigned)(me * slot_bytes + off), 0, 16);
        }
        grid_barrier(ctr, (unsigned)(p + 1) * (unsigned)G, err);
        // gather: chunk j of the read set comes from workgroup (me + 1 + j) % G, a pseudo-random 16-byte piece of its slot
        for (int j = threadIdx.x; j * 16 < rd_bytes; j += THREADS) {
            const int src = (me + 1 + j) % G;
            const int piece = (int)((unsigned)(j * 2654435761u) >> 8) % (wr_bytes >> 4);
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(src * slot_bytes + piece * 16), 0, 16);
            const unsigned v = (unsigned)(p * 65536 + src * 1024 + piece);
            nbad += (d[0] != v) + (d[1] != v + 1) + (d[2] != v + 2) + (d[3] != v + 3);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const int GMAX = 256, SLOT = 64 * 1024;
    unsigned *ctr, *err, *xcc, *bad; char *buf;
    (void)hipMalloc(&ctr, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&bad, 4); (void)hipMalloc(&xcc, GMAX * 4);
    (void)hipMalloc(&buf, (size_t)2 * GMAX * SLOT);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int phases = 200;
    struct Cfg { int G, xcds, wr, rd; };
    const Cfg cfgs[] = {{32, 1, 16, 0}, {32, 1, 8192, 0}, {32, 1, 8192, 16384}, {32, 1, 8192, 131072}, {32, 1, 32768, 131072},
                        {64, 2, 16, 0}, {64, 2, 8192, 131072}, {128, 4, 16, 0}, {128, 4, 8192, 131072},
                        {256, 8, 16, 0}, {256, 8, 8192, 16384}, {256, 8, 8192, 131072}, {16, 1, 16, 0}, {8, 1, 16, 0}};
    for (const Cfg &c : cfgs) {
        float best = 1e9f;
        unsigned herr = 0, hbad = 0;
        std::vector<unsigned> hx(GMAX, 99);
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(ctr, 0, 4); (void)hipMemset(err, 0, 4); (void)hipMemset(bad, 0, 4);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k<512>, dim3(8 * c.G / c.xcds), dim3(512), 0, 0, ctr, err, xcc, buf, c.G, c.xcds, phases, c.wr, c.rd, SLOT, bad);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hx.data(), xcc, c.G * 4, hipMemcpyDeviceToHost);
        unsigned mask = 0; for (int i = 0; i < c.G; ++i) mask |= 1u << (hx[i] & 31);
        printf("G %3d on %d XCD(s) [xcc mask %02x]  publish %6d B  gather %6d B per WG : %6.2f us per phase   timeout %u  bad words %u\n",
               c.G, c.xcds, mask, c.wr, c.rd, best * 1e3f / phases, herr, hbad);
    }
    return 0;
}
