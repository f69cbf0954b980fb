// Micro-benchmark: per-instruction VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f, c = seed * 0.25f;
  unsigned long long g = (unsigned long long)threadIdx.x * 0x9E3779B97F4A7C15ull;
  double d0 = seed, d1 = seed + 1;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(d0), "v"(d1));) }
    if (OP == 2) { REP16(asm volatile("v_cmp_gt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_u64 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc\n v_cmp_gt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_u64 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc" : "+v"(a0) : "v"(g), "v"(d0), "v"(b), "v"(c) : "vcc");) }
    if (OP == 3) { REP16(asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_f32 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc\n v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_f32 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc" : "+v"(a0) : "v"(a1), "v"(a2), "v"(b), "v"(c) : "vcc");) }
    if (OP == 4) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));) }
    if (OP == 5) { REP16(asm volatile("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n v_cvt_f32_i32_sdwa %4, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %5, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %6, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %7, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));) }
    if (OP == 6) { REP16(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 7) { REP16(asm volatile("v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1" : "+v"(d0), "+v"(d1) : "v"(*(double*)&a0), "v"(*(double*)&a2));) }
    if (OP == 8) { REP16(asm volatile("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3\n v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(d0));) }
    if (OP == 9) { REP16(asm volatile("v_sub_f32 %0, %8, %9 clamp\n v_sub_f32 %1, %8, %9 clamp\n v_sub_f32 %2, %8, %9 clamp\n v_sub_f32 %3, %8, %9 clamp\n v_sub_f32 %4, %8, %9 clamp\n v_sub_f32 %5, %8, %9 clamp\n v_sub_f32 %6, %8, %9 clamp\n v_sub_f32 %7, %8, %9 clamp" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));) }
    if (OP == 10) { REP16(asm volatile("v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_u32 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc\n v_cmp_gt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cmp_gt_u32 vcc, %2, %1\n v_cndmask_b32 %0, %4, %3, vcc" : "+v"(a0) : "v"(a1), "v"(a2), "v"(b), "v"(c) : "vcc");) }
    if (OP == 11) { REP16(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 12) { REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 13) { unsigned int* u = (unsigned int*)&a0; (void)u; REP16(asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n v_addc_co_u32_e32 %1, vcc, %5, %1, vcc\n v_add_co_u32_e32 %2, vcc, %4, %2\n v_addc_co_u32_e32 %3, vcc, %5, %3, vcc\n v_add_co_u32_e32 %0, vcc, %4, %0\n v_addc_co_u32_e32 %1, vcc, %5, %1, vcc\n v_add_co_u32_e32 %2, vcc, %4, %2\n v_addc_co_u32_e32 %3, vcc, %5, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");) }
    if (OP == 14) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2" : "+v"(d0), "+v"(d1) : "v"(g));) }
    if (OP == 15) { REP16(asm volatile("v_min3_u32 %0, %0, %8, %9\n v_max3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_max3_u32 %3, %3, %8, %9\n v_min3_u32 %4, %4, %8, %9\n v_max3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_max3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 16) { REP16(asm volatile("v_lshl_add_u32 %0, %8, 2, %9\n v_lshl_add_u32 %1, %8, 2, %9\n v_lshl_add_u32 %2, %8, 2, %9\n v_lshl_add_u32 %3, %8, 2, %9\n v_lshl_add_u32 %4, %8, 2, %9\n v_lshl_add_u32 %5, %8, 2, %9\n v_lshl_add_u32 %6, %8, 2, %9\n v_lshl_add_u32 %7, %8, 2, %9" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(b), "v"(c));) }
    if (OP == 17) { REP16(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 18) { REP16(asm volatile("v_min_u32 %0, %8, %0\n v_max_u32 %1, %8, %1\n v_min_u32 %2, %8, %2\n v_max_u32 %3, %8, %3\n v_min_u32 %4, %8, %4\n v_max_u32 %5, %8, %5\n v_min_u32 %6, %8, %6\n v_max_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 19) { REP16(asm volatile("v_alignbit_b32 %0, %8, %0, 7\n v_alignbit_b32 %1, %8, %1, 7\n v_alignbit_b32 %2, %8, %2, 7\n v_alignbit_b32 %3, %8, %3, 7\n v_bfe_u32 %4, %8, 3, 9\n v_bfe_u32 %5, %8, 3, 9\n v_and_or_b32 %6, %8, %9, %6\n v_and_or_b32 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)d0 + (float)d1 + (float)g;
}
template <int OP> double run(const char* name, float* d, int ninstr_per_iter) {
  const int iters = 2000, blocks = 256 * 8;  // 8 blocks/CU -> 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves_per_simd = blocks * 4.0 / (256 * 4);
  double instr_per_simd = waves_per_simd * (double)iters * ninstr_per_iter;
  double cyc = ms * 1e-3 * 2.4e9;  // at max clock
  printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (assuming 2.4 GHz)\n", name, ms, cyc / instr_per_simd);
  return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", d, 128);
  run<1>("v_pk_fma_f32", d, 128);
  run<11>("v_mul_f32", d, 128);
  run<6>("v_add_f32", d, 128);
  run<8>("v_pk_add_f32", d, 128);
  run<2>("v_cmp_gt_u64+v_cndmask", d, 128);
  run<3>("v_cmp_gt_f32+v_cndmask", d, 128);
  run<10>("v_cmp_gt_u32+v_cndmask", d, 128);
  run<4>("v_cvt_f32_ubyteN", d, 128);
  run<5>("v_cvt_f32_i32_sdwa sext", d, 128);
  run<9>("v_sub_f32 clamp", d, 128);
  run<7>("v_fma_f64", d, 128);
  run<12>("v_fma_mix_f32", d, 128);
  run<13>("v_add_co+v_addc_co", d, 128);
  run<14>("v_lshl_add_u64", d, 128);
  run<15>("v_min3/max3_u32", d, 128);
  run<16>("v_lshl_add_u32", d, 128);
  run<17>("v_add_u32", d, 128);
  run<18>("v_min/max_u32", d, 128);
  run<19>("alignbit/bfe/and_or", d, 128);
  return 0;
}
