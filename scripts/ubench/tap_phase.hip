// Micro-benchmark: one tap phase of corr_cboc.hip - sixteen samples of {v_and_or_b32 (bit 31 of the ramp word -> +-1.0), v_add_u32 (ramp
// step), two v_fma_f32 (the sign-modulated running sums)} - alone on the VALU, and with the phase's LDS traffic next to it:
//   A  the 64 VALU instructions only
//   B  + 32 ds_write_addtid_b32 (m0 set-up + s_nop per eight, as the kernel parks its running sums)
//   C  + 8 ds_write_b128 instead (the same bytes in a quarter of the instructions)
//   D  B + six ds_read2st64_b32 and their s_waitcnt at the end of the phase (the transition rows)
//   F / G  A + 16 writes, in two bursts of eight / two after every second sample;  H / I  32 writes, two after every sample (m0 set
//      each time / once): does spreading the writes let the VALU keep its rate?
//   J  the 32 dwords as 16 ds_write2st64_b32 (one address register, the same rows);  K  as 16 ds_write_b64 (rows of [lane][re, im])
//   E  A with the and_or / add of each sample replaced by fma (all 64 at the double rate: the issue ceiling)
// at 1, 2, 3, 4 waves per SIMD.  Answers whether the kernel's 4.4 cycles per VALU instruction at four waves per SIMD is the LDS
// instructions' doing.   Build: hipcc --offload-arch=gfx950 -O3 tap_phase.hip -o tap_phase ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define S4(n0, n1, n2, n3)                                                                                                           \
  "v_and_or_b32 %4, %2, %3, 1.0\n v_add_u32 %2, %2, %5\n v_fma_f32 %0, %4, %" #n0 ", %0\n v_fma_f32 %1, %4, %" #n1 ", %1\n"           \
  "v_and_or_b32 %4, %2, %3, 1.0\n v_add_u32 %2, %2, %5\n v_fma_f32 %0, %4, %" #n2 ", %0\n v_fma_f32 %1, %4, %" #n3 ", %1\n"
#define F4(n0, n1, n2, n3)                                                                                                           \
  "v_fma_f32 %4, %2, %3, 1.0\n v_fma_f32 %2, %2, %5, %2\n v_fma_f32 %0, %4, %" #n0 ", %0\n v_fma_f32 %1, %4, %" #n1 ", %1\n"          \
  "v_fma_f32 %4, %2, %3, 1.0\n v_fma_f32 %2, %2, %5, %2\n v_fma_f32 %0, %4, %" #n2 ", %0\n v_fma_f32 %1, %4, %" #n3 ", %1\n"
#define W8 "s_mov_b32 m0, %10\n s_nop 0\n ds_write_addtid_b32 %0 offset:0\n ds_write_addtid_b32 %1 offset:0x100\n ds_write_addtid_b32 %0 offset:0x200\n ds_write_addtid_b32 %1 offset:0x300\n ds_write_addtid_b32 %0 offset:0x400\n ds_write_addtid_b32 %1 offset:0x500\n ds_write_addtid_b32 %0 offset:0x600\n ds_write_addtid_b32 %1 offset:0x700\n"
#define W2 "s_mov_b32 m0, %10\n s_nop 0\n ds_write_addtid_b32 %0 offset:0\n ds_write_addtid_b32 %1 offset:0x100\n"
#define W2N "ds_write_addtid_b32 %0 offset:0x200\n ds_write_addtid_b32 %1 offset:0x300\n"
#define S2(n0, n1) "v_and_or_b32 %4, %2, %3, 1.0\n v_add_u32 %2, %2, %5\n v_fma_f32 %0, %4, %" #n0 ", %0\n v_fma_f32 %1, %4, %" #n1 ", %1\n"
#define X2(o0, o1) "ds_write2st64_b32 %11, %0, %1 offset0:" #o0 " offset1:" #o1 "\n"
#define X64(o) "ds_write_b64 %11, %13 offset:" #o "\n"
#define W128 "ds_write_b128 %11, %12 offset:0\n ds_write_b128 %11, %12 offset:0x400\n"
#define R6 "ds_read2st64_b32 %13, %11 offset1:1\n ds_read2st64_b32 %14, %11 offset0:2 offset1:3\n ds_read2st64_b32 %15, %11 offset0:4 offset1:5\n ds_read2st64_b32 %13, %11 offset0:6 offset1:7\n ds_read2st64_b32 %14, %11 offset0:8 offset1:9\n ds_read2st64_b32 %15, %11 offset0:10 offset1:11\n s_waitcnt lgkmcnt(0)\n"
#define OPS : "+v"(sr), "+v"(si), "+v"(w), "+s"(mask), "+v"(sg), "+v"(step), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+s"(m0v), "+v"(addr), "+v"(q), "+v"(r0), "+v"(r1), "+v"(r2)::"memory"

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  extern __shared__ float lds[];
  float sr = seed, si = seed + 1, sg = 1.f, y0 = seed + threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
  unsigned int w = threadIdx.x * 2654435761u, step = 0x4e5e0a73u, mask = 0x80000000u;
  unsigned int m0v = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 8192u), addr = (threadIdx.x & 63) * (V == 9 ? 4u : V == 10 ? 8u : 16u) + (threadIdx.x >> 6) * 8192u;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f4 q = {y0, y1, y2, y3};
  f2 r0 = {0.f, 0.f}, r1 = r0, r2 = r0;
  lds[threadIdx.x] = seed;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(S4(6, 7, 8, 9) S4(7, 8, 9, 6) S4(8, 9, 6, 7) S4(9, 6, 7, 8) S4(6, 7, 8, 9) S4(7, 8, 9, 6) S4(8, 9, 6, 7) S4(9, 6, 7, 8) OPS);
    if (V == 1) asm volatile(S4(6, 7, 8, 9) S4(7, 8, 9, 6) W8 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 S4(6, 7, 8, 9) S4(7, 8, 9, 6) W8 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 OPS);
    if (V == 2) asm volatile(S4(6, 7, 8, 9) S4(7, 8, 9, 6) W128 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W128 S4(6, 7, 8, 9) S4(7, 8, 9, 6) W128 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W128 OPS);
    if (V == 3) asm volatile(S4(6, 7, 8, 9) S4(7, 8, 9, 6) W8 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 S4(6, 7, 8, 9) S4(7, 8, 9, 6) W8 S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 R6 OPS);
    if (V == 5) asm volatile(S4(6, 7, 8, 9) S4(7, 8, 9, 6) S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 S4(6, 7, 8, 9) S4(7, 8, 9, 6) S4(8, 9, 6, 7) S4(9, 6, 7, 8) W8 OPS);
    if (V == 6) asm volatile(S4(6, 7, 8, 9) W2 S4(7, 8, 9, 6) W2 S4(8, 9, 6, 7) W2 S4(9, 6, 7, 8) W2 S4(6, 7, 8, 9) W2 S4(7, 8, 9, 6) W2 S4(8, 9, 6, 7) W2 S4(9, 6, 7, 8) W2 OPS);
    if (V == 7) asm volatile(S2(6, 7) W2 S2(8, 9) W2 S2(7, 8) W2 S2(9, 6) W2 S2(8, 9) W2 S2(6, 7) W2 S2(9, 6) W2 S2(7, 8) W2 S2(6, 7) W2 S2(8, 9) W2 S2(7, 8) W2 S2(9, 6) W2 S2(8, 9) W2 S2(6, 7) W2 S2(9, 6) W2 S2(7, 8) W2 OPS);
    if (V == 8) asm volatile("s_mov_b32 m0, %10\n s_nop 0\n" S2(6, 7) W2N S2(8, 9) W2N S2(7, 8) W2N S2(9, 6) W2N S2(8, 9) W2N S2(6, 7) W2N S2(9, 6) W2N S2(7, 8) W2N S2(6, 7) W2N S2(8, 9) W2N S2(7, 8) W2N S2(9, 6) W2N S2(8, 9) W2N S2(6, 7) W2N S2(9, 6) W2N S2(7, 8) W2N OPS);
    if (V == 9) asm volatile(S2(6, 7) X2(0, 1) S2(8, 9) X2(2, 3) S2(7, 8) X2(4, 5) S2(9, 6) X2(6, 7) S2(8, 9) X2(8, 9) S2(6, 7) X2(10, 11) S2(9, 6) X2(12, 13) S2(7, 8) X2(14, 15) S2(6, 7) X2(16, 17) S2(8, 9) X2(18, 19) S2(7, 8) X2(20, 21) S2(9, 6) X2(22, 23) S2(8, 9) X2(24, 25) S2(6, 7) X2(26, 27) S2(9, 6) X2(28, 29) S2(7, 8) X2(30, 31) OPS);
    if (V == 10) asm volatile(S2(6, 7) X64(0) S2(8, 9) X64(512) S2(7, 8) X64(1024) S2(9, 6) X64(1536) S2(8, 9) X64(2048) S2(6, 7) X64(2560) S2(9, 6) X64(3072) S2(7, 8) X64(3584) S2(6, 7) X64(4096) S2(8, 9) X64(4608) S2(7, 8) X64(5120) S2(9, 6) X64(5632) S2(8, 9) X64(6144) S2(6, 7) X64(6656) S2(9, 6) X64(7168) S2(7, 8) X64(7680) OPS);
    if (V == 4) asm volatile(F4(6, 7, 8, 9) F4(7, 8, 9, 6) F4(8, 9, 6, 7) F4(9, 6, 7, 8) F4(6, 7, 8, 9) F4(7, 8, 9, 6) F4(8, 9, 6, 7) F4(9, 6, 7, 8) OPS);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sr + si + sg + (float)w + r0.x + r1.x + r2.x + lds[(threadIdx.x * 7) & 255];
}

template <int V>
void run(const char* name, float* d, int blocks_per_cu) {
  const int iters = 20000, blocks = 256 * blocks_per_cu;
  const size_t lds = 4 * 8192 + 4096;  // four waves x 8 KB of rows
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double phases_per_simd = (double)blocks_per_cu * iters;
  printf("%-44s %d waves/SIMD: %8.3f ms -> %6.1f cycles per phase per SIMD = %.2f per VALU instruction (2.4 GHz)\n", name, blocks_per_cu, ms,
         ms * 1e-3 * 2.4e9 / phases_per_simd, ms * 1e-3 * 2.4e9 / phases_per_simd / 64.0);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int b : {1, 2, 3, 4}) {
    run<0>("A 64 VALU (16 x and_or, add, fma, fma)", d, b);
    run<1>("B A + 32 ds_write_addtid_b32", d, b);
    run<2>("C A + 8 ds_write_b128", d, b);
    run<3>("D B + 6 ds_read2st64_b32 + waitcnt", d, b);
    run<4>("E 64 v_fma_f32", d, b);
    run<5>("F A + 16 ds_write_addtid_b32 (2 x 8)", d, b);
    run<6>("G A + 16 ds_write_addtid_b32 (8 x 2)", d, b);
    run<7>("H A + 32 ds_write_addtid_b32 (16 x 2)", d, b);
    run<9>("J A + 16 ds_write2st64_b32 (same rows)", d, b);
    run<10>("K A + 16 ds_write_b64 ([lane][re, im] rows)", d, b);
    run<8>("I A + 32 ds_write_addtid_b32 (16 x 2, one m0)", d, b);
  }
  return 0;
}
