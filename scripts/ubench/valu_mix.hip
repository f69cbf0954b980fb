// Micro-benchmark: does the double-rate issue of plain f32 fma/mul survive in the lane kernel's instruction mix, and at
// which occupancy?  One "sample" = 13 v_fmac + 8 v_mul + 2 v_cvt(sdwa) + v_cmp + 2 v_cndmask + v_lshl_add_u64 + v_lshl_add_u32.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MIX>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11, w0 = 0.5f, w1 = 0.25f, y0 = 1.f, y1 = 2.f, c0 = 1.f, c1 = -1.f, cp = 0.f;
  unsigned long long q = (unsigned long long)threadIdx.x * 0x9E3779B97F4A7C15ull, dq = 0x123456789ull;
  unsigned int word = threadIdx.x * 2654435761u, addr = 0;
  for (int i = 0; i < iters; ++i) {
    if (MIX == 0) {  // the lane kernel's mix, interleaved roughly as the compiler schedules it
      REP8(asm volatile(
          "v_cvt_f32_i32_sdwa %12, sext(%19) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n"
          "v_cvt_f32_i32_sdwa %13, sext(%19) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
          "v_lshl_add_u64 %20, %20, 0, %21\n"
          "v_lshl_add_u32 %22, %19, 3, 0\n"
          "v_mul_f32 %14, %12, %16\n v_mul_f32 %15, %13, %16\n v_fmac_f32 %14, %13, %17\n v_fmac_f32 %15, %12, %17\n"
          "v_cmp_lt_i32 vcc, -1, %19\n v_cndmask_b32 %18, %23, %24, vcc\n v_cndmask_b32 %18, %24, %23, vcc\n"
          "v_fmac_f32 %0, %23, %14\n v_fmac_f32 %1, %23, %15\n v_fmac_f32 %2, %18, %14\n v_fmac_f32 %3, %18, %15\n v_fmac_f32 %4, %24, %14\n v_fmac_f32 %5, %24, %15\n"
          "v_fmac_f32 %6, %23, %14\n v_fmac_f32 %7, %23, %15\n v_fmac_f32 %8, %18, %14\n v_fmac_f32 %9, %18, %15\n v_fmac_f32 %10, %24, %14\n v_fmac_f32 %11, %24, %15\n"
          "v_mul_f32 %12, %16, %17\n v_mul_f32 %13, %17, %17\n v_fmac_f32 %12, %16, %16\n v_fmac_f32 %13, %16, %17\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
            "+v"(y0), "+v"(y1), "+v"(w0), "+v"(w1), "+v"(c0), "+v"(c1), "+v"(cp), "+v"(word), "+v"(q), "+v"(dq), "+v"(addr)
          : "v"(c0), "v"(c1) : "vcc");)
    }
    if (MIX == 1) {  // the 21 full-rate instructions alone
      REP8(asm volatile(
          "v_mul_f32 %14, %12, %16\n v_mul_f32 %15, %13, %16\n v_fmac_f32 %14, %13, %17\n v_fmac_f32 %15, %12, %17\n"
          "v_fmac_f32 %0, %23, %14\n v_fmac_f32 %1, %23, %15\n v_fmac_f32 %2, %18, %14\n v_fmac_f32 %3, %18, %15\n v_fmac_f32 %4, %24, %14\n v_fmac_f32 %5, %24, %15\n"
          "v_fmac_f32 %6, %23, %14\n v_fmac_f32 %7, %23, %15\n v_fmac_f32 %8, %18, %14\n v_fmac_f32 %9, %18, %15\n v_fmac_f32 %10, %24, %14\n v_fmac_f32 %11, %24, %15\n"
          "v_mul_f32 %12, %16, %17\n v_mul_f32 %13, %17, %17\n v_fmac_f32 %12, %16, %16\n v_fmac_f32 %13, %16, %17\n v_mul_f32 %18, %16, %17\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
            "+v"(y0), "+v"(y1), "+v"(w0), "+v"(w1), "+v"(c0), "+v"(c1), "+v"(cp), "+v"(word), "+v"(q), "+v"(dq), "+v"(addr)
          : "v"(c0), "v"(c1) : "vcc");)
    }
    if (MIX == 2) {  // the 7 other instructions alone
      REP8(asm volatile(
          "v_cvt_f32_i32_sdwa %12, sext(%19) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n"
          "v_cvt_f32_i32_sdwa %13, sext(%19) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
          "v_lshl_add_u64 %20, %20, 0, %21\n"
          "v_lshl_add_u32 %22, %19, 3, 0\n"
          "v_cmp_lt_i32 vcc, -1, %19\n v_cndmask_b32 %18, %23, %24, vcc\n v_cndmask_b32 %18, %24, %23, vcc\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
            "+v"(y0), "+v"(y1), "+v"(w0), "+v"(w1), "+v"(c0), "+v"(c1), "+v"(cp), "+v"(word), "+v"(q), "+v"(dq), "+v"(addr)
          : "v"(c0), "v"(c1) : "vcc");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + y0 + y1 + w0 + w1 + cp + (float)q + (float)addr;
}
template <int MIX> void run(const char* name, float* d, int blocks_per_cu, int threads) {
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks_per_cu * (threads / 64) / 4.0;
  const double samples_per_simd = waves_per_simd * iters * 8.0;
  printf("%-34s %2.0f waves/SIMD: %8.3f ms -> %.1f cycles per sample per SIMD (2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / samples_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int b : {1, 2, 4, 8}) {
    run<0>("lane mix (21 full-rate + 7 other)", d, b, 256);
    run<1>("21 full-rate only", d, b, 256);
    run<2>("7 other only", d, b, 256);
  }
  return 0;
}
