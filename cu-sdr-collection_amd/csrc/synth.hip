// synth.hip — libgnsssynth.so: synthetic IF record generator on the GPU (test / bench utility, NOT part
// of the product ABI).  Fills a device buffer with int8 interleaved I/Q samples of the same signal
// model as cu-sdr-collection_amd/synth.py (sum of BPSK-spread carriers + complex AWGN, rounded and
// clipped), so that the 60-s / 2.16-GB config-2 record of BASELINE.json can be produced in HBM in a
// fraction of a second instead of minutes of NumPy time.  Noise comes from a counter-based hash
// (splitmix64 -> Box-Muller), so a record is a pure function of (seed, sample index).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

extern "C" {
typedef struct gs_sat {
  int32_t prn;
  int32_t code_index;         // row of `codes`
  double doppler;             // Hz
  double code_phase_samples;  // sample index (may be fractional) at which a code period starts
  double carrier_phase;       // rad at n = 0
  double amplitude;           // LSB
  double code_rate;           // chips/s including code Doppler
} gs_sat;

typedef struct gs_sat2 {
  int32_t prn;
  int32_t code_len;           // chips (table entries) per code period
  int32_t code_offset;        // offset of this satellite's chips in the flat `codes` array
  int32_t pilot_offset;       // offset of the pilot component's chips (same length), or -1
  int32_t bit_periods;        // code periods per data bit
  int32_t reserved;
  double doppler, code_phase_samples, carrier_phase, amplitude, code_rate;
  double pilot_phase;         // carrier phase offset of the pilot (rad)
  double intermediate_freq;   // Hz
} gs_sat2;
}

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

constexpr int kMaxSats = 64;

struct Params {
  gs_sat sat[kMaxSats];
  int nsat;
  int code_len;
  int bit_periods;
  double fs, intermediate_freq, sigma;
  uint64_t seed, nsamples;
};

__global__ __launch_bounds__(256) void synth_kernel(int8_t* __restrict__ out, const int8_t* __restrict__ codes,
                                                     const Params* __restrict__ pp) {
  const Params& p = *pp;
  const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t n0 = chunk * 8;
  if (n0 >= p.nsamples) return;
  float accI[8], accQ[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) accI[j] = accQ[j] = 0.f;
  for (int s = 0; s < p.nsat; ++s) {
    const gs_sat& st = p.sat[s];
    const double ratio = st.code_rate / p.fs;
    const double cp0 = ((double)n0 - st.code_phase_samples) * ratio;  // chips since the reference code start
    const double cpf = floor(cp0);
    const long long chip0 = (long long)cpf;
    float frac = (float)(cp0 - cpf);
    const float fr = (float)ratio;
    const double turns = (p.intermediate_freq + st.doppler) / p.fs;
    const double ph = st.carrier_phase * 0.15915494309189535 + (double)n0 * turns;
    float sn, cs;
    sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
    float dsn, dcs;
    sincospif(2.0f * (float)(turns - floor(turns)), &dsn, &dcs);
    const int8_t* code = codes + (size_t)st.code_index * p.code_len;
    const float amp = (float)st.amplitude;
    long long chip = chip0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // floor-div by code_len for possibly negative chip indices
      long long period = chip / p.code_len;
      long long idx = chip - period * p.code_len;
      if (idx < 0) {
        idx += p.code_len;
        period -= 1;
      }
      long long bit = period / p.bit_periods;
      if (period < 0 && bit * p.bit_periods != period) bit -= 1;
      const uint64_t h = splitmix64(p.seed ^ ((uint64_t)st.prn << 48) ^ (uint64_t)bit * 0x9E3779B97F4A7C15ull);
      const float data = (h & 1) ? 1.f : -1.f;
      const float v = amp * data * (float)code[idx];
      accI[j] += v * cs;
      accQ[j] += v * sn;
      // advance one sample
      const float ncs = cs * dcs - sn * dsn;
      const float nsn = cs * dsn + sn * dcs;
      cs = ncs;
      sn = nsn;
      frac += fr;
      while (frac >= 1.f) {
        frac -= 1.f;
        ++chip;
      }
    }
  }
  unsigned int w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t h = splitmix64(p.seed * 0xD1342543DE82EF95ull + (n0 + j));
    const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u2 = (float)(uint32_t)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1)) * (float)p.sigma;
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    float vi = rintf(accI[j] + r * cs), vq = rintf(accQ[j] + r * sn);
    vi = fminf(fmaxf(vi, -127.f), 127.f);
    vq = fminf(fmaxf(vq, -127.f), 127.f);
    const unsigned int bi = (unsigned int)(int)vi & 0xffu, bq = (unsigned int)(int)vq & 0xffu;
    w[j >> 1] |= (bi | (bq << 8)) << ((j & 1) * 16);
  }
  if (n0 + 8 <= p.nsamples) {
    *reinterpret_cast<uint4*>(out + 2 * n0) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (uint64_t j = 0; n0 + j < p.nsamples; ++j) {
      out[2 * (n0 + j)] = (int8_t)((w[j >> 1] >> ((j & 1) * 16)) & 0xff);
      out[2 * (n0 + j) + 1] = (int8_t)((w[j >> 1] >> ((j & 1) * 16 + 8)) & 0xff);
    }
  }
}


// ---- version 2: heterogeneous records (BASELINE config 5: several signal families in one band) ------------------
// Every satellite names its own code (offset / length into one flat chip array), optional pilot component at a carrier
// phase offset, data-bit length and intermediate frequency; the record is written as int8 or int16 I/Q.
struct Params2 {
  gs_sat2 sat[kMaxSats];
  int nsat;
  int out_i16;   // bit 0: int16 output; bit 1: Q,I sample order (GLONASS front ends, GLO_GL1/include/tracking.m:227)
  double fs, sigma;
  uint64_t seed, nsamples;
};

__global__ __launch_bounds__(256) void synth2_kernel(void* __restrict__ out, const int8_t* __restrict__ codes,
                                                      const Params2* __restrict__ pp) {
  const Params2& p = *pp;
  const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t n0 = chunk * 8;
  if (n0 >= p.nsamples) return;
  float accI[8], accQ[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) accI[j] = accQ[j] = 0.f;
  for (int s = 0; s < p.nsat; ++s) {
    const gs_sat2& st = p.sat[s];
    const double ratio = st.code_rate / p.fs;
    const double cp0 = ((double)n0 - st.code_phase_samples) * ratio;
    const double cpf = floor(cp0);
    long long chip = (long long)cpf;
    float frac = (float)(cp0 - cpf);
    const float fr = (float)ratio;
    const double turns = (st.intermediate_freq + st.doppler) / p.fs;
    const double ph = st.carrier_phase * 0.15915494309189535 + (double)n0 * turns;
    float sn, cs;
    sincospif(2.0f * (float)(ph - floor(ph)), &sn, &cs);
    float dsn, dcs;
    sincospif(2.0f * (float)(turns - floor(turns)), &dsn, &dcs);
    const int8_t* code = codes + st.code_offset;
    const int8_t* pilot = st.pilot_offset >= 0 ? codes + st.pilot_offset : nullptr;
    float psn = 0.f, pcs = 0.f;
    if (pilot) sincosf((float)st.pilot_phase, &psn, &pcs);
    const float amp = (float)st.amplitude;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long long period = chip / st.code_len;
      long long idx = chip - period * st.code_len;
      if (idx < 0) {
        idx += st.code_len;
        period -= 1;
      }
      long long bit = period / st.bit_periods;
      if (period < 0 && bit * st.bit_periods != period) bit -= 1;
      const uint64_t h = splitmix64(p.seed ^ ((uint64_t)(unsigned int)st.prn << 48) ^ ((uint64_t)s << 40) ^ (uint64_t)bit * 0x9E3779B97F4A7C15ull);
      const float data = (h & 1) ? 1.f : -1.f;
      float re = data * (float)code[idx], im = 0.f;
      if (pilot) {
        const float pv = (float)pilot[idx];
        re += pv * pcs;
        im = pv * psn;
      }
      accI[j] += amp * (re * cs - im * sn);
      accQ[j] += amp * (re * sn + im * cs);
      const float ncs = cs * dcs - sn * dsn;
      const float nsn = cs * dsn + sn * dcs;
      cs = ncs;
      sn = nsn;
      frac += fr;
      while (frac >= 1.f) {
        frac -= 1.f;
        ++chip;
      }
    }
  }
  const bool i16 = (p.out_i16 & 1) != 0, qi = (p.out_i16 & 2) != 0;
  const float lim = i16 ? 32767.f : 127.f;
  short v16[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t h = splitmix64(p.seed * 0xD1342543DE82EF95ull + (n0 + j));
    const float u1 = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = (float)(uint32_t)(h & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1)) * (float)p.sigma;
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    float vi = rintf(accI[j] + r * cs), vq = rintf(accQ[j] + r * sn);
    v16[2 * j + (qi ? 1 : 0)] = (short)fminf(fmaxf(vi, -lim), lim);
    v16[2 * j + (qi ? 0 : 1)] = (short)fminf(fmaxf(vq, -lim), lim);
  }
  const uint64_t nvalid = (n0 + 8 <= p.nsamples) ? 8 : p.nsamples - n0;
  if (i16) {
    short* o = reinterpret_cast<short*>(out) + 2 * n0;
    for (uint64_t j = 0; j < 2 * nvalid; ++j) o[j] = v16[j];
  } else {
    int8_t* o = reinterpret_cast<int8_t*>(out) + 2 * n0;
    for (uint64_t j = 0; j < 2 * nvalid; ++j) o[j] = (int8_t)v16[j];
  }
}

}  // namespace

extern "C" int gs_version(void) { return 2; }

// Fills d_out[0 .. 2*nsamples) (device pointer, 16-B aligned) on `device`.  `codes`: host array of
// ncodes x code_len int8 (+-1).  Returns 0 or a negative hipError.
extern "C" int gs_generate(void* d_out, uint64_t nsamples, int device, double fs, double intermediate_freq,
                           const int8_t* codes, int ncodes, int code_len, int bit_periods, const gs_sat* sats,
                           int nsat, double sigma, uint64_t seed) {
  if (!d_out || !codes || !sats || nsat < 0 || nsat > kMaxSats || ((uintptr_t)d_out & 15)) return -1;
  if (hipSetDevice(device) != hipSuccess) return -2;
  Params hp;
  for (int i = 0; i < nsat; ++i) hp.sat[i] = sats[i];
  hp.nsat = nsat;
  hp.code_len = code_len;
  hp.bit_periods = bit_periods;
  hp.fs = fs;
  hp.intermediate_freq = intermediate_freq;
  hp.sigma = sigma;
  hp.seed = seed;
  hp.nsamples = nsamples;
  Params* dp = nullptr;
  int8_t* dcodes = nullptr;
  if (hipMalloc((void**)&dp, sizeof(Params)) != hipSuccess) return -3;
  if (hipMalloc((void**)&dcodes, (size_t)ncodes * code_len) != hipSuccess) return -3;
  int rc = 0;
  if (hipMemcpy(dp, &hp, sizeof(Params), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dcodes, codes, (size_t)ncodes * code_len, hipMemcpyHostToDevice) != hipSuccess)
    rc = -4;
  if (!rc) {
    const uint64_t chunks = (nsamples + 7) / 8;
    const unsigned int grid = (unsigned int)((chunks + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(256), 0, 0, (int8_t*)d_out, dcodes, dp);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = -5;
  }
  (void)hipFree(dp);
  (void)hipFree(dcodes);
  return rc;
}

// Version 2: per-satellite codes / pilots / intermediate frequencies, int8 (out_i16 = 0) or int16 I/Q output.
extern "C" int gs_generate2(void* d_out, uint64_t nsamples, int device, double fs, const int8_t* codes, uint64_t codes_bytes,
                            const gs_sat2* sats, int nsat, double sigma, uint64_t seed, int out_i16) {
  if (!d_out || !codes || !sats || nsat < 0 || nsat > kMaxSats || ((uintptr_t)d_out & 15)) return -1;
  if (hipSetDevice(device) != hipSuccess) return -2;
  static Params2 hp;
  for (int i = 0; i < nsat; ++i) hp.sat[i] = sats[i];
  hp.nsat = nsat;
  hp.out_i16 = out_i16;
  hp.fs = fs;
  hp.sigma = sigma;
  hp.seed = seed;
  hp.nsamples = nsamples;
  Params2* dp = nullptr;
  int8_t* dcodes = nullptr;
  if (hipMalloc((void**)&dp, sizeof(Params2)) != hipSuccess) return -3;
  if (hipMalloc((void**)&dcodes, (size_t)codes_bytes) != hipSuccess) return -3;
  int rc = 0;
  if (hipMemcpy(dp, &hp, sizeof(Params2), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dcodes, codes, (size_t)codes_bytes, hipMemcpyHostToDevice) != hipSuccess)
    rc = -4;
  if (!rc) {
    const uint64_t chunks = (nsamples + 7) / 8;
    const unsigned int grid = (unsigned int)((chunks + 255) / 256);
    hipLaunchKernelGGL(synth2_kernel, dim3(grid), dim3(256), 0, 0, d_out, dcodes, dp);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = -5;
  }
  (void)hipFree(dp);
  (void)hipFree(dcodes);
  return rc;
}
