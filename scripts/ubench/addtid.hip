// Semantics check of ds_write_addtid_b32 on gfx950: which LDS dword does lane L of wave W write?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned int* out, int nop) {
  extern __shared__ unsigned int lds[];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 0xdeadu;
  __syncthreads();
  const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)lds + 1024u * (threadIdx.x >> 6));
  const unsigned int val = 1000u * (threadIdx.x >> 6) + (threadIdx.x & 63);
  if (nop)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:256" ::"v"(val), "s"(base) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:256" ::"v"(val), "s"(base) : "memory", "m0");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
  unsigned int* d;
  hipMalloc(&d, 2048 * 4);
  static unsigned int h[2048];
  for (int nop = 0; nop < 2; ++nop) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, nop);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("nop=%d:", nop);
    int shown = 0;
    for (int i = 0; i < 2048 && shown < 24; ++i)
      if (h[i] != 0xdeadu && (i % 64 < 2 || i % 64 == 63)) { printf(" [%d]=%u", i, h[i]); ++shown; }
    int cnt = 0;
    for (int i = 0; i < 2048; ++i) cnt += h[i] != 0xdeadu;
    printf("  (written %d)\n", cnt);
  }
  return 0;
}
