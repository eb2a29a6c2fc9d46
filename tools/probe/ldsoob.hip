// probe: (1) does a ds_read_b128 whose address lies past the workgroup's LDS allocation return zeros, and (2) does such a lane stay
// out of the bank arbitration?  The tile kernels read a "zero row" for every absent (offset, row) slot of the rulebook — 54 % of
// the lanes at level 1 — and those reads conflict with every real row on the zero row's banks.
// Lane (g, i): row r(lane) x 32 B + (g & 1) x 16 B, as conv_tile16's operand fetch.  Patterns:
//   0 consecutive rows, none absent        1 consecutive, half absent -> row 0      2 consecutive, half absent -> out of range
//   3 random rows, none absent             4 random, half absent -> row 0           5 random, half absent -> out of range
//   hipcc --offload-arch=gfx950 -O3 -o ldsoob.bin ldsoob.hip && ./ldsoob.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char rows[1024 * 32];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    for (int e = tid; e < 1024 * 8; e += 256) reinterpret_cast<unsigned *>(rows)[e] = e < 8 ? 0u : (unsigned)e;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)rows, half = (unsigned)(g & 1) * 16u;
    unsigned acc = 0, oob_or = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned h = (unsigned)(lane * 2654435761u + (unsigned)(it * 8 + q) * 40503u);
            unsigned row = (P >= 3) ? 1u + ((h >> 7) % 1023u) : 1u + (unsigned)((it * 8 + q) * 16 + i + 300 * (g >> 1)) % 1023u;
            const bool absent = (P % 3 != 0) && (((h >> 3) & 1u) != 0u);
            unsigned a = base + row * 32u + half;
            if (absent) a = (P % 3 == 1) ? base + half : (0xffffffe0u | half);
            asm volatile("ds_read_b128 %0, %1" : "=v"(r[q]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc += r[q][0] ^ r[q][3];
            if (P % 3 == 2) {
                const unsigned h = (unsigned)(lane * 2654435761u + (unsigned)(it * 8 + q) * 40503u);
                if ((h >> 3) & 1u) oob_or |= r[q][0] | r[q][1] | r[q][2] | r[q][3];
            }
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
    if (oob_or) atomicOr(out + 1024 * 256, oob_or);
}

template <int P> void run(unsigned *d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(d + 1024 * 256, 0, 4);
    hipLaunchKernelGGL((k<P>), dim3(1024), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<P>), dim3(1024), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned flag = 0; hipMemcpy(&flag, d + 1024 * 256, 4, hipMemcpyDeviceToHost);
    printf("pattern %d: %8.1f us  (%.2f ns per wave-read per CU)  nonzero bits read out of range: 0x%x\n", P, ms * 1e3f,
           ms * 1e6f / ((double)iters * 8 * 4 * 1024 / 256), flag);
}
int main() {
    unsigned *d; hipMalloc(&d, (1024 * 256 + 1) * 4);
    const int it = 2000;
    run<0>(d, it); run<1>(d, it); run<2>(d, it); run<3>(d, it); run<4>(d, it); run<5>(d, it);
    return 0;
}
