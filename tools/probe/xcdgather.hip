// probe: one layer boundary of a coarse-level executor with the executor's own access pattern.
// N rows of RB bytes; G workgroups on XCDS XCDs own N / G consecutive rows each.  Per phase a workgroup rewrites its rows
// (16-byte stores), passes the grid barrier and then gathers NBR neighbour rows per own row (rows near its own: the voxel
// order has locality), 16 bytes per lane, contiguous lanes covering a row, LOADS_IN_FLIGHT loads issued before the first
// is consumed.  Protocol 0: sc1 stores + sc1 loads (no fences).  Protocol 1: plain stores + agent release fence by one lane,
// agent acquire fence by one lane after the barrier, plain loads (L1 may serve the re-reads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PROTO>
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target, unsigned *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (PROTO == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { *err = 1; break; }
        }
        if (PROTO == 1 || PROTO == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}


template <int PROTO, int THREADS, int INFL>
__global__ __launch_bounds__(THREADS) void k(unsigned *ctr, unsigned *err, char *buf, int N, int RB, int G, int xcds, int phases,
                                              int nbr, unsigned *bad) {
    if ((int)(blockIdx.x & 7) >= xcds) return;
    const int me = (blockIdx.x >> 3) * xcds + (blockIdx.x & 7);
    const int per = (N + G - 1) / G, r0 = me * per, r1 = min(N, r0 + per);
    const int cpr = RB / 16;   // 16-byte chunks per row
    const unsigned half = (unsigned)N * (unsigned)RB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 2u * half, 0x00020000);
    // protocol 2 (SAME-XCD ONLY): plain stores (write-through L1 -> the shared L2, lines stay there), drained, no fence; sc1
    // loads (bypass L1, served by L2).  Protocol 3 (same XCD only): plain stores, acquire fence (L1 invalidate), plain loads.
    constexpr int AUX = PROTO == 0 ? 16 : 0;
    constexpr int AUXL = (PROTO == 0 || PROTO == 2) ? 16 : 0;
    unsigned nbad = 0;
    const int deltas[16] = {-1, 1, -9, 9, -10, 10, -11, 11, -83, 83, -84, 84, -85, 85, -2, 2};
    for (int p = 0; p < phases; ++p) {
        const unsigned base = (p & 1) ? half : 0u;
        for (int i = threadIdx.x; i < (r1 - r0) * cpr; i += THREADS) {
            const int row = r0 + i / cpr, c = i % cpr;
            const unsigned v = (unsigned)(p * 1000003 + row * 16 + c);
            const u32x4 d = {v, v ^ 1u, v ^ 2u, v ^ 3u};
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, base + (unsigned)row * (unsigned)RB + (unsigned)c * 16u, 0, AUX);
        }
        grid_barrier<PROTO>(ctr, (unsigned)(p + 1) * (unsigned)G, err);
        const int items = (r1 - r0) * nbr * cpr;
        for (int i0 = threadIdx.x; i0 < items; i0 += THREADS * INFL) {
            u32x4 d[INFL];
            unsigned want[INFL];
#pragma unroll
            for (int q = 0; q < INFL; ++q) {
                const int i = i0 + q * THREADS;
                const int c = i % cpr, t = i / cpr, row = r0 + t / nbr, kk = t % nbr;
                int src = row + deltas[kk & 15] * (1 + (kk >> 4));
                src = ((src % N) + N) % N;
                want[q] = (unsigned)(p * 1000003 + src * 16 + c);
                d[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, i < items ? base + (unsigned)src * (unsigned)RB + (unsigned)c * 16u : 0x80000000u, 0, AUXL);
            }
#pragma unroll
            for (int q = 0; q < INFL; ++q)
                if (i0 + q * THREADS < items)
                    nbad += (d[q][0] != want[q]) + (d[q][1] != (want[q] ^ 1u)) + (d[q][2] != (want[q] ^ 2u)) + (d[q][3] != (want[q] ^ 3u));
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned *ctr, *err, *bad; char *buf;
    (void)hipMalloc(&ctr, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&buf, (size_t)2 * 65536 * 256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int phases = 200;
    struct Cfg { int proto, N, RB, G, xcds, nbr, threads, infl; };
    const Cfg cfgs[] = {
        {0, 1920, 160, 32, 1, 12, 512, 8}, {0, 1920, 160, 32, 1, 12, 1024, 8}, {0, 1920, 160, 32, 1, 12, 512, 16}, {0, 1920, 160, 32, 1, 12, 1024, 4},
        {0, 1920, 160, 64, 1, 12, 512, 8}, {0, 1920, 160, 64, 1, 12, 256, 8}, {0, 1920, 160, 64, 1, 12, 256, 16}, {0, 1920, 160, 128, 1, 12, 256, 8},
        {0, 1920, 160, 16, 1, 12, 1024, 8}, {0, 1920, 160, 8, 1, 12, 1024, 8},
        {2, 1920, 160, 32, 1, 12, 1024, 8}, {2, 1920, 160, 64, 1, 12, 512, 8},
        {0, 8400, 128, 32, 1, 12, 1024, 8}, {0, 8400, 128, 64, 1, 12, 512, 8}, {0, 420, 192, 32, 1, 12, 1024, 8}, {0, 420, 192, 16, 1, 12, 1024, 8}, {0, 80, 224, 8, 1, 12, 1024, 8}};
    for (const Cfg &c : cfgs) {
        float best = 1e9f;
        unsigned herr = 0, hbad = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(ctr, 0, 4); (void)hipMemset(err, 0, 4); (void)hipMemset(bad, 0, 4);
            (void)hipEventRecord(a);
            const dim3 grid(8 * c.G / c.xcds);
#define RUN(P, T, I) hipLaunchKernelGGL((k<P, T, I>), grid, dim3(T), 0, 0, ctr, err, buf, c.N, c.RB, c.G, c.xcds, phases, c.nbr, bad)
            if (c.threads == 512 && c.infl == 8) { if (c.proto == 0) RUN(0, 512, 8); else RUN(2, 512, 8); }
            else if (c.threads == 1024 && c.infl == 8) { if (c.proto == 0) RUN(0, 1024, 8); else RUN(2, 1024, 8); }
            else if (c.threads == 512 && c.infl == 16) { if (c.proto == 0) RUN(0, 512, 16); else RUN(2, 512, 16); }
            else if (c.threads == 256 && c.infl == 8) { if (c.proto == 0) RUN(0, 256, 8); else RUN(2, 256, 8); }
            else if (c.threads == 256 && c.infl == 16) { if (c.proto == 0) RUN(0, 256, 16); else RUN(2, 256, 16); }
            else { if (c.proto == 0) RUN(0, 1024, 4); else RUN(2, 1024, 4); }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        const double gathered = (double)c.N * c.nbr * c.RB;
        printf("%s  thr %4d infl %2d  N %5d x %3d B  G %3d on %d XCD  %2d nbrs (%.2f MB gathered) : %6.2f us per phase   timeout %u  bad %u\n",
               c.proto == 0 ? "sc1 st/ld   " : c.proto == 1 ? "plain+fences" : c.proto == 2 ? "plain st/sc1 ld" : "plain st/acq/plain ld", c.threads, c.infl, c.N, c.RB, c.G, c.xcds, c.nbr, gathered / 1e6, best * 1e3f / phases, herr, hbad);
    }
    return 0;
}
