// probe: semantics of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out, int pattern) {
  __shared__ unsigned short tile[16 * 16];
  for (int e = threadIdx.x; e < 256; e += 64) tile[e] = (unsigned short)e;   // tile[r][c] = r*16 + c
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, g = l >> 4;
  int row, cg;
  if (pattern == 0) { row = 4 * g + (t >> 2); cg = t & 3; }      // chunk t = (row t/4, colgroup t%4)
  else              { row = 4 * g + (t & 3);  cg = t >> 2; }     // chunk t = (row t%4, colgroup t/4)
  unsigned addr = (unsigned)(uintptr_t)&tile[row * 16 + cg * 4];
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; l += (l < 20 ? 1 : 13)) printf("  lane %2d: (r%d,c%d) (r%d,c%d) (r%d,c%d) (r%d,c%d)\n", l, h[l*4]/16, h[l*4]%16, h[l*4+1]/16, h[l*4+1]%16, h[l*4+2]/16, h[l*4+2]%16, h[l*4+3]/16, h[l*4+3]%16);
  }
  return 0;
}
