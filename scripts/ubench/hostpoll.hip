// Two transport questions for a host-fed persistent kernel on this box:
//  (1) can the CPU write fine-grained DEVICE memory directly (large BAR)?  A kernel that is already running polls it.
//  (2) does a wave that polls HOST memory get its own host-memory stores out while it keeps running?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void poll_kernel(volatile unsigned int* flag, volatile unsigned int* out, int rounds) {
  for (int r = 1; r <= rounds; ++r) {
    unsigned int spins = 0;
    while (__builtin_nontemporal_load((const unsigned int*)flag) != (unsigned int)r && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2);
    out[0] = 1000u * r + (spins >= (1u << 24) ? 999u : 1u);  // answer into HOST memory
    __threadfence_system();
  }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned int *dflag = nullptr, *hflag = nullptr, *hout = nullptr;
  hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped);
  hipHostMalloc((void**)&hout, 64, hipHostMallocMapped);
  hipError_t e = hipExtMallocWithFlags((void**)&dflag, 64, hipDeviceMallocFinegrained);
  printf("fine-grained device alloc: %s\n", hipGetErrorString(e));
  for (int mode = 0; mode < 2; ++mode) {
    unsigned int* flag = mode == 0 ? hflag : dflag;
    if (!flag) continue;
    if (mode == 0) *hflag = 0; else hipMemset(dflag, 0, 64);
    *hout = 0;
    hipDeviceSynchronize();
    const int rounds = 2000;
    hipLaunchKernelGGL(poll_kernel, dim3(1), dim3(64), 0, 0, flag, hout, rounds);
    double t0 = now_us(), worst = 0;
    int ok = 0;
    for (int r = 1; r <= rounds; ++r) {
      double t1 = now_us();
      *(volatile unsigned int*)flag = (unsigned int)r;   // CPU store: into host memory (mode 0) or straight into VRAM (mode 1)
      while (*(volatile unsigned int*)hout / 1000u != (unsigned int)r && now_us() - t1 < 2e5) {}
      if (*(volatile unsigned int*)hout == 1000u * r + 1u) ++ok;
      worst = std::max(worst, now_us() - t1);
      if (now_us() - t1 >= 2e5) { printf("mode %d: round %d timed out (out = %u)\n", mode, r, *hout); break; }
    }
    printf("mode %d (%s flag): %d / %d rounds answered, %.2f us per round trip, worst %.1f us\n", mode, mode == 0 ? "host-memory" : "device-memory", ok, rounds, (now_us() - t0) / rounds, worst);
    for (int r = 1; r <= rounds; ++r) *(volatile unsigned int*)flag = (unsigned int)r;  // let the kernel run out
    *(volatile unsigned int*)flag = (unsigned int)rounds;
    hipDeviceSynchronize();
  }
  return 0;
}
