// probe: LDS throughput of the wgrad transpose round trip (2 x ds_write_b128 per lane-row, then
// 4 x ds_read_b64_tr_b16) for several row layouts of a 64-row x 16-channel bf16 tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// byte address of (row, byte-in-row 0..31) for layout L
template <int L> __device__ __forceinline__ unsigned addr_of(int row, int byte) {
    if (L == 0) return row * 48 + byte;                       // padded rows (current)
    if (L == 1) return row * 32 + byte;                       // dense
    if (L == 2) return (row * 32 + byte) ^ (((row >> 3) & 1) << 7);        // swap 128-B halves of odd row octets
    if (L == 3) return (row * 32 + byte) ^ (((row >> 2) & 3) << 4) ;       // xor 16-B chunk with row/4
    if (L == 4) return row * 40 + byte;                       // 8-byte pad (b128 stores misaligned -> two b64)
    if (L == 5) return (row * 32 + byte) ^ (((row >> 3) & 7) << 5);        // xor 32-B slot with row/8
    if (L == 7) return (byte >> 4) * (1024 + 128) + row * 16 + (byte & 15);   // 16-byte chunk planes
    if (L == 8) return (byte >> 4) * (1024 + 64) + row * 16 + (byte & 15);
    if (L == 9) return ((byte >> 4) * 1024 + row * 16 + (byte & 15)) ^ (((row >> 3) & 1) << 7);
    return row * 64 + byte;                                   // L == 6: 64-B rows
}

template <int L, int MODE>   // MODE 0: write+read, 1: write only, 2: read only
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[4][64 * 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    unsigned char* my = tile[wid];
    u32x4 v0 = {(unsigned)lane, 1u, 2u, 3u}, v1 = {4u, 5u, 6u, (unsigned)lane};
    unsigned acc = 0;
    const unsigned base = (unsigned)(uintptr_t)my;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2) {
            if (L == 4) {
                *reinterpret_cast<u32x2*>(my + addr_of<L>(lane, 0)) = (u32x2){v0[0], v0[1]};
                *reinterpret_cast<u32x2*>(my + addr_of<L>(lane, 8)) = (u32x2){v0[2], v0[3]};
                *reinterpret_cast<u32x2*>(my + addr_of<L>(lane, 16)) = (u32x2){v1[0], v1[1]};
                *reinterpret_cast<u32x2*>(my + addr_of<L>(lane, 24)) = (u32x2){v1[2], v1[3]};
            } else {
                *reinterpret_cast<u32x4*>(my + addr_of<L>(lane, 0)) = v0;
                *reinterpret_cast<u32x4*>(my + addr_of<L>(lane, 16)) = v1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (MODE != 1) {
            u32x2 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (q >> 1) * 32 + 8 * g + (q & 1) * 4 + (i >> 2);
                const unsigned a = base + addr_of<L>(row, (i & 3) * 8);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[q]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += r[q][0] ^ r[q][1];
        }
        __builtin_amdgcn_wave_barrier();
        v0[0] += acc; v1[3] ^= acc;
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <int L, int MODE> float run(float* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<L, MODE>), dim3(1024), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<L, MODE>), dim3(1024), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}
#define ROW(L) printf("layout %d: write+read %8.1f us   write %8.1f   read %8.1f\n", L, run<L,0>(d, it), run<L,1>(d, it), run<L,2>(d, it));
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    const int it = 1000;
    ROW(0) ROW(2) ROW(7) ROW(8) ROW(9)
    return 0;
}
