// probe: how many gather instructions per cycle can a gfx950 CU retire?  Each wave issues `buffer_load_dwordx4`
// (or x2) instructions back to back, 8 in flight, with a chosen lane -> address pattern over a 19 MB feature
// matrix (600k rows of 32 B; L2 / Infinity-Cache resident after the first pass):
//   0 coalesced        lane l reads 16 B at (chunk * 64 + l) * 16                        (1 KB contiguous)
//   1 rows32 local     lanes (2r, 2r+1) read the halves of row base + perm[r] (rows within +-64 of the base)
//   2 rows32 random    ... of a random row of the whole matrix
//   3 rows32 local, 37 % of the rows present (other lanes masked out of EXEC, as conv_fast does)
//   4 rows32 local, 37 % present, absent lanes read an out-of-range offset instead of being masked
//   5 rows64 local     4 lanes per 64-byte row (the 32-channel layers)
//   6 rows16 local x2  dwordx2: 4 lanes per 32-byte row (8 B each)
//   7 same line        every lane reads row 0 (all hits, one line)
// Output: ns per instruction per CU and bytes/s, for 4 and 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o gatherrate gatherrate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 make_rsrc(const void *p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32) & 0xffffu; r[2] = bytes; r[3] = 0x00020000u;
    return r;
}

template <int PAT>
__global__ __launch_bounds__(256) void k(const unsigned char *feat, unsigned feat_bytes, const int *rnd, int n_rows,
                                         int iters, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const u32x4 rs = make_rsrc(feat, feat_bytes);
    unsigned acc = 0;
    // per-lane offset recipe
    const int r2 = lane >> 1, h2 = lane & 1, r4 = lane >> 2, q4 = lane & 3;
    const unsigned seed = (unsigned)wave * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
        unsigned off[8];
        unsigned long long mask[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned chunk = (unsigned)(it * 8 + j);
            const unsigned base_row = (unsigned)(((unsigned long long)wave * 977u + chunk * 31u) % (unsigned)(n_rows - 256));
            mask[j] = ~0ull;
            if (PAT == 0) off[j] = ((base_row * 2u) + lane) * 16u;
            else if (PAT == 1 || PAT == 3 || PAT == 4) {
                const unsigned row = base_row + (unsigned)((r2 * 37 + chunk * 5) & 127);
                off[j] = row * 32u + h2 * 16u;
                if (PAT != 1) {
                    const bool present = ((r2 * 2654435761u + chunk * 40503u + seed) >> 16) % 100u < 37u;
                    if (PAT == 3) mask[j] = __ballot(present || lane == 0);
                    else if (!present) off[j] = 0x80000000u;
                }
            } else if (PAT == 2) off[j] = (unsigned)rnd[(wave * 64 + chunk * 32 + r2) & 0xfffff] * 32u + h2 * 16u;
            else if (PAT == 5) off[j] = (base_row + (unsigned)((r4 * 37 + chunk * 5) & 127)) * 64u + q4 * 16u;
            else if (PAT == 6) off[j] = (base_row + (unsigned)((r4 * 37 + chunk * 5) & 127)) * 32u + q4 * 8u;
            else off[j] = h2 * 16u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (PAT == 6) {
                u32x2 t;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(t) : "v"(off[j]), "s"(rs));
                v[j] = (u32x4){t[0], t[1], 0u, 0u};
            } else if (PAT == 3) {
                unsigned long long keep;
                v[j] = (u32x4){0u, 0u, 0u, 0u};
                asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\ts_mov_b64 exec, %1"
                             : "+v"(v[j]), "=&s"(keep) : "v"(off[j]), "s"(rs), "s"(mask[j]));
            } else {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[j]) : "v"(off[j]), "s"(rs));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("" : "+v"(v[j]));
            acc ^= v[j][0] ^ v[j][3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT>
static void run(const char *name, const unsigned char *feat, unsigned bytes, const int *rnd, int n_rows, unsigned *sink,
                double lanes_frac, double bytes_per_lane) {
    for (int wps = 4; wps <= 8; wps += 4) {
        const int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD on 256 CUs
        const int iters = 200;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), 0, 0, feat, bytes, rnd, n_rows, 20, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), 0, 0, feat, bytes, rnd, n_rows, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double instr = (double)blocks * 4 * iters * 8;
        const double per_cu_ns = ms * 1e6 / (instr / 256.0);
        printf("%-28s %d waves/SIMD: %7.2f ns per instruction per CU (%5.1f cycles @2.4 GHz), %6.2f TB/s of useful bytes\n", name,
               wps, per_cu_ns, per_cu_ns * 2.4, instr * 64 * lanes_frac * bytes_per_lane / (ms * 1e-3) / 1e12);
    }
}

int main() {
    const int n_rows = 600000;
    const unsigned bytes = (unsigned)n_rows * 64u;   // 64 B per row so that pattern 5 stays in range
    unsigned char *feat; int *rnd; unsigned *sink;
    hipMalloc((void **)&feat, bytes); hipMemset(feat, 1, bytes);
    std::vector<int> h(1 << 20);
    srand(1);
    for (auto &x : h) x = (int)(((long long)rand() * 32768 + rand()) % n_rows);
    hipMalloc((void **)&rnd, h.size() * 4); hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc((void **)&sink, 64);
    run<0>("coalesced 16 B/lane", feat, bytes, rnd, n_rows, sink, 1.0, 16);
    run<1>("32-B rows, local", feat, bytes, rnd, n_rows, sink, 1.0, 16);
    run<2>("32-B rows, random", feat, bytes, rnd, n_rows, sink, 1.0, 16);
    run<3>("32-B rows, 37 %, EXEC mask", feat, bytes, rnd, n_rows, sink, 0.37, 16);
    run<4>("32-B rows, 37 %, OOB lanes", feat, bytes, rnd, n_rows, sink, 0.37, 16);
    run<5>("64-B rows, local", feat, bytes, rnd, n_rows, sink, 1.0, 16);
    run<6>("32-B rows as 4 x 8 B", feat, bytes, rnd, n_rows, sink, 1.0, 8);
    run<7>("one line", feat, bytes, rnd, n_rows, sink, 1.0, 16);
    return 0;
}
