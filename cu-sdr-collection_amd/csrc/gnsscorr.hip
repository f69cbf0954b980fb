// gnsscorr.hip — context, IF buffer, code tables, correlate / replay entry points of the
// C-ABI declared in include/gnsscorr.h.  gfx950 only; there is no CPU fallback in this
// library: every compute entry point launches HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <utility>

#include "corr_common.h"

static thread_local std::string g_last_error;

void gc_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int gc_bytes_per_sample(int dtype, int layout) {
  const int comp = (layout == GC_REAL) ? 1 : 2;
  return comp * (dtype == GC_I16 ? 2 : 1);
}

extern "C" {

const char* gc_last_error(void) { return g_last_error.c_str(); }
int gc_api_version(void) { return GC_API_VERSION; }

int gc_create(gc_context** out, int device_id) {
  if (!out) {
    gc_set_error("gc_create: null output pointer");
    return GC_E_INVALID;
  }
  *out = nullptr;
  int ndev = 0;
  GC_HIP(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) {
    gc_set_error("gc_create: device %d not available (%d devices visible)", device_id, ndev);
    return GC_E_HIP;
  }
  GC_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  GC_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    gc_set_error("gc_create: device %d is %s; this library is built for gfx950 (MI355X) only",
                 device_id, prop.gcnArchName);
    return GC_E_UNSUPPORTED;
  }
  gc_context* ctx = new (std::nothrow) gc_context();
  if (!ctx) return GC_E_NOMEM;
  ctx->device = device_id;
  ctx->compute_units = prop.multiProcessorCount;
  // prop.name is empty on some ROCm 7.2 boxes: fall back to the architecture string
  std::snprintf(ctx->device_name, sizeof ctx->device_name, "%s", prop.name[0] ? prop.name : prop.gcnArchName);
  GC_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  GC_HIP(hipEventCreate(&ctx->ev_start));
  GC_HIP(hipEventCreate(&ctx->ev_stop));
  GC_HIP(hipMalloc(&ctx->d_channels, sizeof(DevChannel) * GC_MAX_CHANNELS));
  GC_HIP(hipMemset(ctx->d_channels, 0, sizeof(DevChannel) * GC_MAX_CHANNELS));
  if (const char* e = GC_TUNE_ENV("GC_FORCE_GENERIC")) ctx->force_generic = std::atoi(e) != 0;  // tuning: lane kernel everywhere
  *out = ctx;
  return GC_OK;
}

static void free_if(gc_context* ctx) {
  if (ctx->d_if && ctx->if_owned) (void)hipFree(ctx->d_if);
  ctx->d_if = nullptr;
  ctx->if_owned = false;
  ctx->if_nsamples = 0;
  ctx->if_capacity_bytes = 0;
  ctx->acq_cond_n = 0;  // a conditioned signal (gc_acq_condition) belongs to the record it was made from
}

int gc_destroy(gc_context* ctx) {
  if (!ctx) return GC_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_if(ctx);
  for (auto& c : ctx->ch) {
    for (auto& t : c.d_tab)
      if (t) (void)hipFree(t);
    for (auto& t : c.d_tab2)
      if (t) (void)hipFree(t);
    for (auto& t : c.d_tab2b)
      if (t) (void)hipFree(t);
    if (c.d_tabh) (void)hipFree(c.d_tabh);
    if (c.d_tabf) (void)hipFree(c.d_tabf);
  }
  if (ctx->d_channels) (void)hipFree(ctx->d_channels);
  if (ctx->d_blocks) (void)hipFree(ctx->d_blocks);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  if (ctx->h_blocks_pinned) (void)hipHostFree(ctx->h_blocks_pinned);
  if (ctx->h_out_pinned) (void)hipHostFree(ctx->h_out_pinned);
  if (ctx->h_tagged_pinned) (void)hipHostFree(ctx->h_tagged_pinned);
  if (ctx->d_replay_blocks) (void)hipFree(ctx->d_replay_blocks);
  if (ctx->d_replay_out) (void)hipFree(ctx->d_replay_out);
  for (GcBuf& b : ctx->trk) gc_buf_free(b);
  for (GcBuf& b : ctx->nav) gc_buf_free(b);
  for (GcBuf& b : ctx->acqbuf) gc_buf_free(b);
  gc_acq_free(ctx);
  (void)hipEventDestroy(ctx->ev_start);
  (void)hipEventDestroy(ctx->ev_stop);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GC_OK;
}

int gc_device_info(gc_context* ctx, char* name, int name_len, int* compute_units) {
  if (!ctx) return GC_E_INVALID;
  if (name && name_len > 0) std::snprintf(name, (size_t)name_len, "%s", ctx->device_name);
  if (compute_units) *compute_units = ctx->compute_units;
  return GC_OK;
}

int gc_device_count(int* count) {
  if (!count) return GC_E_INVALID;
  *count = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return GC_OK;  // no driver / no device: zero devices, not an error of the call
  }
  *count = n;
  return GC_OK;
}

int gc_synchronize(gc_context* ctx) {
  if (!ctx) return GC_E_INVALID;
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

// Test hook: 1 forces the generic (per-sample table lookup) kernel even where the fast kernel applies.
int gc_force_generic_kernel(gc_context* ctx, int on) {
  if (!ctx) return GC_E_INVALID;
  ctx->force_generic = on != 0;
  return GC_OK;
}

int gc_set_sampling_freq(gc_context* ctx, double fs) {
  if (!ctx || !(fs > 0)) {
    gc_set_error("gc_set_sampling_freq: fs must be positive");
    return GC_E_INVALID;
  }
  ctx->fs = fs;
  return GC_OK;
}

// ---- IF buffer ---------------------------------------------------------------------------

static int check_fmt(int dtype, int layout) {
  if ((dtype != GC_I8 && dtype != GC_I16) || (layout != GC_REAL && layout != GC_IQ && layout != GC_QI)) {
    gc_set_error("unknown dtype/layout (%d/%d)", dtype, layout);
    return GC_E_INVALID;
  }
  return GC_OK;
}

int gc_alloc_if(gc_context* ctx, uint64_t nsamples, int dtype, int layout) {
  if (!ctx || nsamples == 0) {
    gc_set_error("gc_alloc_if: bad arguments");
    return GC_E_INVALID;
  }
  int rc = check_fmt(dtype, layout);
  if (rc) return rc;
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  free_if(ctx);
  const uint64_t bytes = nsamples * (uint64_t)gc_bytes_per_sample(dtype, layout);
  const uint64_t cap = ((bytes + 63) / 64) * 64 + 64;  // kernels read whole 32-B chunks
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess) {
    gc_set_error("gc_alloc_if: hipMalloc(%llu) failed: %s", (unsigned long long)cap, hipGetErrorString(e));
    return GC_E_NOMEM;
  }
  GC_HIP(hipMemsetAsync((uint8_t*)p + (cap - 128), 0, 128, ctx->stream));
  ctx->d_if = (uint8_t*)p;
  ctx->if_owned = true;
  ctx->if_nsamples = nsamples;
  ctx->if_capacity_bytes = cap;
  ctx->if_dtype = dtype;
  ctx->if_layout = layout;
  return GC_OK;
}

int gc_load_if(gc_context* ctx, const void* samples, uint64_t nsamples, int dtype, int layout) {
  if (!ctx || !samples) {
    gc_set_error("gc_load_if: null argument");
    return GC_E_INVALID;
  }
  int rc = gc_alloc_if(ctx, nsamples, dtype, layout);
  if (rc) return rc;
  const uint64_t bytes = nsamples * (uint64_t)gc_bytes_per_sample(dtype, layout);
  GC_HIP(hipMemcpyAsync(ctx->d_if, samples, bytes, hipMemcpyHostToDevice, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

// 2-bit sign-magnitude complex samples, two per byte (GPS_L5C/include/unpack_cplx.m:32-49 expands them on the CPU
// into a four-times larger schar file): bits {0,2} = sign, magnitude of I1, {1,3} of Q1, {4,6} of I2, {5,7} of Q2,
// value = (1 + 2*mag) * (1 - 2*sign).  Here the packed bytes cross PCIe and are expanded into the int8 I/Q record
// in HBM: 4 input bytes -> one 16-byte store per thread.
namespace {
__global__ void unpack2bit_kernel(const uint32_t* __restrict__ in, uint4* __restrict__ out, uint64_t nwords) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  const uint32_t w = in[i];
  uint32_t o[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint32_t v = (w >> (8 * b)) & 0xffu;
    auto dec = [&](int sbit, int mbit) -> uint32_t {
      const int val = (1 + 2 * (int)((v >> mbit) & 1u)) * (1 - 2 * (int)((v >> sbit) & 1u));
      return (uint32_t)(uint8_t)(int8_t)val;
    };
    o[b] = dec(0, 2) | (dec(1, 3) << 8) | (dec(4, 6) << 16) | (dec(5, 7) << 24);
  }
  out[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
}  // namespace

int gc_load_if_packed2(gc_context* ctx, const void* packed, uint64_t nbytes) {
  if (!ctx || !packed || nbytes == 0) {
    gc_set_error("gc_load_if_packed2: bad arguments");
    return GC_E_INVALID;
  }
  int rc = gc_alloc_if(ctx, nbytes * 2, GC_I8, GC_IQ);  // two complex samples per packed byte
  if (rc) return rc;
  const uint64_t nwords = (nbytes + 3) / 4;
  uint32_t* d_in = nullptr;
  GC_HIP(hipMalloc((void**)&d_in, nwords * 4));
  GC_HIP(hipMemsetAsync(d_in + (nwords - 1), 0, 4, ctx->stream));
  hipError_t e = hipMemcpyAsync(d_in, packed, nbytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    // the record's allocation is padded past its payload, so a last partial word may store its full 16 bytes
    hipLaunchKernelGGL(unpack2bit_kernel, dim3((unsigned int)((nwords + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_in,
                       reinterpret_cast<uint4*>(ctx->d_if), nwords);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_in);
  if (e != hipSuccess) {
    gc_set_error("gc_load_if_packed2: %s", hipGetErrorString(e));
    return GC_E_HIP;
  }
  return GC_OK;
}

int gc_open_if_file(gc_context* ctx, const char* path, uint64_t skip_bytes, uint64_t nsamples,
                    int dtype, int layout) {
  if (!ctx || !path) return GC_E_INVALID;
  int rc = check_fmt(dtype, layout);
  if (rc) return rc;
  FILE* f = std::fopen(path, "rb");
  if (!f) {
    gc_set_error("Unable to read file %s", path);  // postProcessing.m:157
    return GC_E_INVALID;
  }
  const uint64_t bps = (uint64_t)gc_bytes_per_sample(dtype, layout);
  std::fseek(f, 0, SEEK_END);
  const uint64_t fsize = (uint64_t)std::ftell(f);
  if (skip_bytes >= fsize) {
    std::fclose(f);
    gc_set_error("gc_open_if_file: skip (%llu) beyond end of file", (unsigned long long)skip_bytes);
    return GC_E_RANGE;
  }
  uint64_t avail = (fsize - skip_bytes) / bps;
  if (nsamples == 0 || nsamples > avail) nsamples = avail;
  rc = gc_alloc_if(ctx, nsamples, dtype, layout);
  if (rc) {
    std::fclose(f);
    return rc;
  }
  std::fseek(f, (long)skip_bytes, SEEK_SET);
  // stream through a pinned staging buffer (two halves, copy overlaps the next read)
  const size_t chunk = 64u << 20;
  uint8_t* stage = nullptr;
  if (hipHostMalloc((void**)&stage, 2 * chunk, hipHostMallocDefault) != hipSuccess) {
    std::fclose(f);
    gc_set_error("gc_open_if_file: pinned staging allocation failed");
    return GC_E_NOMEM;
  }
  hipEvent_t ev[2];
  (void)hipEventCreate(&ev[0]);
  (void)hipEventCreate(&ev[1]);
  uint64_t done = 0;
  const uint64_t total = nsamples * bps;
  int half = 0;
  bool used[2] = {false, false};
  int status = GC_OK;
  while (done < total) {
    const size_t n = (size_t)std::min<uint64_t>(chunk, total - done);
    if (used[half]) (void)hipEventSynchronize(ev[half]);
    if (std::fread(stage + half * chunk, 1, n, f) != n) {
      gc_set_error("gc_open_if_file: short read");
      status = GC_E_RANGE;
      break;
    }
    if (hipMemcpyAsync(ctx->d_if + done, stage + half * chunk, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      gc_set_error("gc_open_if_file: H2D copy failed");
      status = GC_E_HIP;
      break;
    }
    (void)hipEventRecord(ev[half], ctx->stream);
    used[half] = true;
    done += n;
    half ^= 1;
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  (void)hipHostFree(stage);
  std::fclose(f);
  return status;
}

int gc_attach_if(gc_context* ctx, void* device_ptr, uint64_t nsamples, int dtype, int layout) {
  if (!ctx || !device_ptr || nsamples == 0) return GC_E_INVALID;
  int rc = check_fmt(dtype, layout);
  if (rc) return rc;
  if (((uintptr_t)device_ptr & 31) != 0) {
    gc_set_error("gc_attach_if: device pointer must be 32-byte aligned");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  free_if(ctx);
  const uint64_t bytes = nsamples * (uint64_t)gc_bytes_per_sample(dtype, layout);
  ctx->d_if = (uint8_t*)device_ptr;
  ctx->if_owned = false;
  ctx->if_nsamples = nsamples;
  ctx->if_capacity_bytes = ((bytes + 31) / 32) * 32;  // the allocation must be readable up to here
  ctx->if_dtype = dtype;
  ctx->if_layout = layout;
  return GC_OK;
}

int gc_if_buffer(gc_context* ctx, void** device_ptr, uint64_t* nsamples) {
  if (!ctx) return GC_E_INVALID;
  if (device_ptr) *device_ptr = ctx->d_if;
  if (nsamples) *nsamples = ctx->if_nsamples;
  return GC_OK;
}

int gc_if_format(gc_context* ctx, int* dtype, int* layout) {
  if (!ctx) return GC_E_INVALID;
  if (dtype) *dtype = ctx->if_dtype;
  if (layout) *layout = ctx->if_layout;
  return GC_OK;
}

int gc_read_if(gc_context* ctx, uint64_t first, uint64_t n, void* dst) {
  if (!ctx || !dst) return GC_E_INVALID;
  if (!ctx->d_if) {
    gc_set_error("gc_read_if: no IF buffer loaded");
    return GC_E_STATE;
  }
  if (first + n > ctx->if_nsamples) {
    gc_set_error("gc_read_if: range beyond IF buffer");
    return GC_E_RANGE;
  }
  const uint64_t bps = (uint64_t)gc_bytes_per_sample(ctx->if_dtype, ctx->if_layout);
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipMemcpyAsync(dst, ctx->d_if + first * bps, n * bps, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

// ---- code tables -------------------------------------------------------------------------

int gc_set_channel(gc_context* ctx, int channel, int arms, double index_scale) {
  if (!ctx || channel < 0 || channel >= GC_MAX_CHANNELS || arms < 1 || arms > GC_MAX_ARMS ||
      !(index_scale >= 1.0)) {
    gc_set_error("gc_set_channel: bad arguments (channel %d, arms %d, R %g)", channel, arms, index_scale);
    return GC_E_INVALID;
  }
  HostChannel& c = ctx->ch[channel];
  c.configured = true;
  c.arms = arms;
  c.index_scale = index_scale;
  for (int a = 0; a < GC_MAX_ARMS; ++a) {  // a re-configured channel starts from defaults: no window, multiplier 1
    c.window[a] = 0;
    c.mult[a] = 1.0;
  }
  c.derived_state = -1;
  ctx->channels_dirty = true;
  return GC_OK;
}

int gc_set_code(gc_context* ctx, int channel, int arm, const int8_t* table, int n_entries,
                double arm_mult) {
  if (!ctx || channel < 0 || channel >= GC_MAX_CHANNELS || !table || n_entries < 3 ||
      !(arm_mult >= 1.0)) {
    gc_set_error("gc_set_code: bad arguments");
    return GC_E_INVALID;
  }
  HostChannel& c = ctx->ch[channel];
  if (!c.configured || arm < 0 || arm >= c.arms) {
    gc_set_error("gc_set_code: channel %d not configured for arm %d", channel, arm);
    return GC_E_STATE;
  }
  for (int i = 0; i < n_entries; ++i)
    if (table[i] < -1 || table[i] > 1) {
      gc_set_error("gc_set_code: table values must be in {-1,0,+1}");
      return GC_E_INVALID;
    }
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  if (c.d_tab[arm]) (void)hipFree(c.d_tab[arm]);
  c.d_tab[arm] = nullptr;
  GC_HIP(hipMalloc((void**)&c.d_tab[arm], (size_t)n_entries + 64));
  GC_HIP(hipMemset(c.d_tab[arm], 0, (size_t)n_entries + 64));
  GC_HIP(hipMemcpy(c.d_tab[arm], table, (size_t)n_entries, hipMemcpyHostToDevice));
  // pre-differenced float2 form for the fast kernel: entry m <-> k = m - 1, c[-1] := c[0], c[>=n] := 0
  {
    std::vector<float2> t2((size_t)n_entries + 3);
    for (int m = 0; m < n_entries + 3; ++m) {
      const int k = m - 1;
      const float c0 = (k < 0) ? (float)table[0] : (k < n_entries) ? (float)table[k] : 0.0f;
      const float c1 = (k + 1 < n_entries) ? (float)table[k + 1] : 0.0f;
      t2[m] = make_float2(c0, c1 - c0);
    }
    if (c.d_tab2[arm]) (void)hipFree(c.d_tab2[arm]);
    c.d_tab2[arm] = nullptr;
    GC_HIP(hipMalloc((void**)&c.d_tab2[arm], t2.size() * sizeof(float2)));
    GC_HIP(hipMemcpy(c.d_tab2[arm], t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice));
    std::vector<unsigned short> t2b(t2.size());
    for (size_t m = 0; m < t2.size(); ++m)
      t2b[m] = (unsigned short)(((unsigned int)(unsigned char)(signed char)t2[m].x) |
                                ((unsigned int)(unsigned char)(signed char)t2[m].y << 8));
    if (c.d_tab2b[arm]) (void)hipFree(c.d_tab2b[arm]);
    c.d_tab2b[arm] = nullptr;
    GC_HIP(hipMalloc((void**)&c.d_tab2b[arm], t2b.size() * sizeof(unsigned short)));
    GC_HIP(hipMemcpy(c.d_tab2b[arm], t2b.data(), t2b.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  }
  c.h_tab[arm].assign(table, table + n_entries);
  c.nent[arm] = n_entries;
  c.mult[arm] = arm_mult;
  c.derived_state = -1;
  ctx->channels_dirty = true;
  return GC_OK;
}

// Optional: restrict LDS staging of arm `arm` to `window_entries` entries starting at the
// block's table_offset (GPS L2C CL code, 1 534 502-entry table, GPS_L2C tracking.m:261).
int gc_set_code_window(gc_context* ctx, int channel, int arm, int window_entries) {
  if (!ctx || channel < 0 || channel >= GC_MAX_CHANNELS || arm < 0 || arm >= GC_MAX_ARMS ||
      window_entries < 0)
    return GC_E_INVALID;
  ctx->ch[channel].window[arm] = window_entries;
  ctx->ch[channel].derived_state = -1;
  ctx->channels_dirty = true;
  return GC_OK;
}

}  // extern "C"

// Three arms {a, b, b'} where b' is b with a sign pattern at six times the ramp rate — BOC(6,1) next to BOC(1,1) (BDS B1C
// wide-band pilot, Galileo E1-C CBOC): entry k6 of b' (padded like every table) is entry p = (k6 + 5) / 6 of b times
// (-1)^(p + k6).  Then the lane kernel needs no third table (csrc/corr_lane.hip, DER).
static bool tables_derivable(const int8_t* t1, int nent1, const int8_t* t6, int nent6) {
  const int n1 = nent1 - 2, n6 = nent6 - 2;
  if (n1 < 1 || n6 != 6 * n1) return false;
  for (int k6 = 0; k6 < nent6; ++k6) {
    const int pidx = (k6 + 5) / 6;
    if (t6[k6] != t1[pidx] * (((pidx + k6) & 1) ? -1 : 1)) return false;
  }
  return true;
}
extern "C" int gc_debug_tables_derivable(const int8_t* t1, int n1, const int8_t* t6, int n6) {
  return (t1 && t6 && tables_derivable(t1, n1, t6, n6)) ? 1 : 0;
}

static bool channel_is_derived_uncached(const HostChannel& c) {
  if (GC_TUNE_ENV("GC_NO_DERIVED_ARM")) return false;
  if (c.arms != 3 || c.mult[0] != c.mult[1] || c.mult[2] != 6.0 * c.mult[1]) return false;
  for (int a = 0; a < 3; ++a)
    if (c.window[a] != 0 || (int)c.h_tab[a].size() != c.nent[a]) return false;
  // the two interleaved arms must fit the lane kernel's LDS budget as f16 at least (f32 up to 96 KiB)
  if (((size_t)std::max(c.nent[0], c.nent[1]) + 2 * gcorr::kGuard) * 2 * 2 + 2048 > 160 * 1024) return false;
  return tables_derivable(c.h_tab[1].data(), c.nent[1], c.h_tab[2].data(), c.nent[2]);
}
bool gc_channel_is_derived(const HostChannel& c) {
  if (c.derived_state < 0) c.derived_state = channel_is_derived_uncached(c) ? 1 : 0;
  return c.derived_state == 1;
}

int gc_sync_channels(gc_context* ctx) {
  if (!ctx->channels_dirty) return GC_OK;
  std::vector<DevChannel> dev(GC_MAX_CHANNELS);
  std::memset(dev.data(), 0, sizeof(DevChannel) * GC_MAX_CHANNELS);
  for (int i = 0; i < GC_MAX_CHANNELS; ++i) {
    const HostChannel& c = ctx->ch[i];
    if (!c.configured) continue;
    DevChannel& d = dev[i];
    d.arms = c.arms;
    d.index_scale = c.index_scale;
    int off = 0;
    for (int a = 0; a < c.arms; ++a) {
      if (!c.d_tab[a]) continue;  // checked per launch
      d.tab[a] = c.d_tab[a];
      d.tab2[a] = c.d_tab2[a];
      d.tab2b[a] = c.d_tab2b[a];
      d.nent[a] = c.nent[a];
      d.mult[a] = c.mult[a];
      d.stage_len[a] = (c.window[a] > 0) ? std::min(c.window[a], c.nent[a]) : c.nent[a];
      d.lds_off[a] = off;
      off += ((d.stage_len[a] + 8 + 15) / 16) * 16;
    }
    bool mixed = false;
    for (int a = 1; a < c.arms; ++a) mixed |= c.mult[a] != c.mult[0];
    const bool derived = mixed && gc_channel_is_derived(c);
    d.derived = derived ? 1 : 0;
    const int larms = derived ? 2 : c.arms;  // arms with a table in LDS
    if (derived) mixed = false;
    d.lds_bytes = off;
    // generic kernel: interleaved f16 copy of the whole tables (when no arm is windowed)
    HostChannel& hc = ctx->ch[i];
    if (hc.d_tabh) (void)hipFree(hc.d_tabh);
    if (hc.d_tabf) (void)hipFree(hc.d_tabf);
    hc.d_tabh = nullptr;
    hc.d_tabf = nullptr;
    bool whole = !mixed;
    int maxn = 0;
    for (int a = 0; a < larms; ++a) {
      whole &= c.d_tab[a] != nullptr && c.window[a] == 0 && (int)c.h_tab[a].size() == c.nent[a];
      maxn = std::max(maxn, d.stage_len[a]);
    }
    if (whole && maxn <= 65536) {
      const int ap = gcorr::gc_arm_pitch(larms);
      const int apf = (derived && GC_LANE_PN != 0) ? 4 : ap;  // derived: {arm 0, arm 1, arm 1 * (-1)^entry, 0}
      const size_t entries = (size_t)maxn + 2 * gcorr::kGuard;
      const size_t bytes = (entries * ap * 2 + 15) / 16 * 16;
      const size_t fbytes = (entries * apf * 4 + 15) / 16 * 16;
      std::vector<uint16_t> t(bytes / 2, 0);
      std::vector<float> tf(fbytes / 4, 0.0f);
      for (int a = 0; a < larms; ++a)
        for (int e = 0; e < c.nent[a]; ++e) {
          const int8_t v = c.h_tab[a][e];
          t[((size_t)e + gcorr::kGuard) * ap + a] = v > 0 ? 0x3C00 : v < 0 ? 0xBC00 : 0;  // f16 +1 / -1 / 0
          tf[((size_t)e + gcorr::kGuard) * apf + a] = (float)v;
          if (derived && GC_LANE_PN != 0 && a == 1) tf[((size_t)e + gcorr::kGuard) * apf + 2] = (e & 1) ? -(float)v : (float)v;
        }
      GC_HIP(hipMalloc((void**)&hc.d_tabh, bytes));
      GC_HIP(hipMemcpy(hc.d_tabh, t.data(), bytes, hipMemcpyHostToDevice));
      GC_HIP(hipMalloc((void**)&hc.d_tabf, fbytes));
      GC_HIP(hipMemcpy(hc.d_tabf, tf.data(), fbytes, hipMemcpyHostToDevice));
      d.tabf = hc.d_tabf;
      d.tabh = hc.d_tabh;
      d.tabh_ap = ap;
      d.tabh_bytes = (int32_t)bytes;
      d.tabf_ap = apf;
      d.tabf_bytes = (int32_t)fbytes;
    }
  }
  GC_HIP(hipMemcpyAsync(ctx->d_channels, dev.data(), sizeof(DevChannel) * GC_MAX_CHANNELS,
                        hipMemcpyHostToDevice, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  ctx->channels_dirty = false;
  return GC_OK;
}

// ---- host-side near-tie analysis --------------------------------------------------------------------
// The kernels evaluate the replica ramps in exact real arithmetic; the reference evaluates
// fl(a + fl(i*d)).  The two can disagree on ceil() only if some sample of the block lies within ~1e-13
// chip of a table edge.  Whether a block contains such a sample is an exact number-theoretic question:
// is there 0 <= i < N with (A + i*D) mod 2^64 inside a small window around 0, where A, D are the 64-bit
// fractions of the ramp start and step?  It is answered in O(log) by a Euclid-style descent, so that
// almost all blocks can be marked tie-free and skip the per-chunk filter in the kernels.
namespace {
typedef unsigned __int128 u128;
const uint64_t kNone = ~0ull;

// smallest x >= 0, x < limit, with l <= (a*x mod m) <= r;  0 <= a < m <= 2^63 here (the 2^64 level is peeled
// off by the caller), 0 <= l <= r < m.
uint64_t first_in_range_u64(uint64_t a, uint64_t m, uint64_t l, uint64_t r, uint64_t limit) {
  if (l == 0) return limit > 0 ? 0 : kNone;
  if (a == 0 || limit == 0) return kNone;
  const uint64_t c = l / a + (l % a != 0);
  if ((u128)a * c <= r) return c < limit ? c : kNone;
  const uint64_t mm = m % a;
  if (mm == 0) return kNone;
  const uint64_t lp = l % a, rp = r % a;  // same quotient: no multiple of a in [l, r]
  const u128 ylim = ((u128)limit * a) / m + 2;
  const uint64_t y = first_in_range_u64(mm, a, a - rp, a - lp, ylim > kNone ? kNone : (uint64_t)ylim);
  if (y == kNone) return kNone;
  const u128 x = ((u128)m * y + l + a - 1) / a;
  return x < limit ? (uint64_t)x : kNone;
}

// same with modulus 2^64 (A, D, l, r given as uint64)
uint64_t first_in_range_2p64(uint64_t a, uint64_t l, uint64_t r, uint64_t limit) {
  if (l == 0) return limit > 0 ? 0 : kNone;
  if (a == 0 || limit == 0) return kNone;
  const u128 m = (u128)1 << 64;
  const uint64_t c = l / a + (l % a != 0);
  if ((u128)a * c <= r) return c < limit ? c : kNone;
  const uint64_t mm = (uint64_t)(m % a);
  if (mm == 0) return kNone;
  const uint64_t lp = l % a, rp = r % a;
  const u128 ylim = ((u128)limit * a) / m + 2;
  const uint64_t y = first_in_range_u64(mm, a, a - rp, a - lp, (uint64_t)ylim);
  if (y == kNone) return kNone;
  const u128 x = (m * y + l + a - 1) / a;
  return x < limit ? (uint64_t)x : kNone;
}

uint64_t frac64(double v) {
  const double f = v - std::floor(v);  // [0,1)
  const double hi = std::floor(f * 4294967296.0);
  const double lo = std::floor((f * 4294967296.0 - hi) * 4294967296.0);
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}
}  // namespace

// First sample index i in [0, n) whose ramp value a + i*step lies within eps (chips, < 0.25) of an integer,
// or -1.  Exact in 2^-64 chip arithmetic.
int64_t gc_first_sample_near_edge(double a, double step, int64_t n, double eps) {
  if (n <= 0) return -1;
  const uint64_t e = (uint64_t)(eps * 18446744073709551616.0);
  const uint64_t A = frac64(a), D = frac64(step);
  const uint64_t W = 2 * e;
  const uint64_t B = A + e;  // mod 2^64: the window [-e, +e] around an integer becomes [0, W] after this shift
  if (B <= W) return 0;      // sample 0 itself sits in the window
  // (B + i*D) mod 2^64 in [0, W]  <=>  (i*D) mod 2^64 in [L, L + W] with L = 2^64 - B  (no wrap: B > W)
  const uint64_t L = 0 - B;
  const uint64_t r = L + W;
  const uint64_t x = first_in_range_2p64(D, L, r, (uint64_t)n);
  return x == kNone ? -1 : (int64_t)x;
}

// Marks tie-free blocks for a launch whose kernel needs the band `eps_chips`: bit 0 of `reserved` = no sample of any ramp within the
// band of a table edge (the kernels skip their near-tie tests); bit 1 (channels with a derived six-fold arm) = no sample within the
// six-fold ramps' own narrow band of a sub-entry edge, whatever the base ramps do (corr_cboc.hip skips its per-sample integer test).
void gc_mark_tie_free(const gc_context* ctx, gc_block* b, int64_t n, double eps_unit_steps) {
  const bool off = GC_TUNE_ENV("GC_NO_TIE_MARK") != nullptr;  // debugging aid: every block takes the in-kernel tests
  for (int64_t i = 0; i < n; ++i) {
    gc_block& k = b[i];
    k.reserved &= ~3;
    if (off) continue;
    const HostChannel& c = ctx->ch[k.channel];
    if (c.mult[0] != 1.0) continue;
    // a derived third arm (BOC(6,1) from BOC(1,1), gc_channel_is_derived) runs its own ramp at mult[2] times the rate: its edges
    // are searched as well, with the window the kernel uses for both ramp sets (the larger multiplier's)
    const double m6 = (c.arms == 3 && gc_channel_is_derived(c)) ? c.mult[2] : 0.0;
    bool plain_mults = true;
    for (int a = 1; a < c.arms; ++a) plain_mults = plain_mults && (c.mult[a] == 1.0 || (a == 2 && m6 != 0.0));
    if (!plain_mults) continue;
    const double R = c.index_scale, sp = k.code_phase_step * R;
    const double maxv = (std::fabs(k.rem_code_phase - k.el_spacing) * R + std::fabs(k.rem_code_phase + k.el_spacing) * R +
                         (double)k.blksize * std::fabs(sp) + 1.0) * std::max(1.0, m6);
    // band: the lane kernel's own window (corr_common.h, splits = 1 is the widest) plus 3 units, or the float
    // step quotient's resolution (eps_unit_steps samples of ramp) for the fast kernel
    const double eps = std::max((gcorr::gc_tie_window_units(maxv, k.blksize / 64 + 2) + 3.0) / 4294967296.0, eps_unit_steps * sp);
    const double starts[3] = {(k.rem_code_phase - k.el_spacing) * R, k.rem_code_phase * R,
                              (k.rem_code_phase + k.el_spacing) * R};
    bool clean = true;
    for (int t = 0; t < 3 && clean; ++t) clean = gc_first_sample_near_edge(starts[t], sp, k.blksize, eps) < 0;
    // the derived arm's sub-entry comes out of the base ramp's fraction times mult[2]: so does that ramp's rounding (gc_tie_window_units6)
    const double eps6 = (gcorr::gc_tie_window_units6(maxv, k.blksize / 64 + 2, m6) + 3.0 * m6) / 4294967296.0;
    bool clean6 = m6 != 0.0;
    for (int t = 0; t < 3 && clean6; ++t) clean6 = gc_first_sample_near_edge(starts[t] * m6, sp * m6, k.blksize, eps6) < 0;
    if (clean6) k.reserved |= 2;
    for (int t = 0; t < 3 && clean && m6 != 0.0; ++t) clean = gc_first_sample_near_edge(starts[t] * m6, sp * m6, k.blksize, std::max(eps, eps6)) < 0;
    if (clean) k.reserved |= 1;
  }
}

extern "C" int gc_build_flags(void) { return GC_TUNING ? GC_BUILD_TUNING : 0; }
extern "C" int gc_debug_last_kernel(const gc_context* ctx) { return ctx ? ctx->last_kernel : -2; }
extern "C" int gc_debug_last_track_mode(const gc_context* ctx) { return ctx ? ctx->last_track_mode : -1; }

namespace {
template <int K>
__global__ __launch_bounds__(64) void wts_debug_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int lane = (int)threadIdx.x;
  float v[K];
#pragma unroll
  for (int c = 0; c < K; ++c) v[c] = in[c * 64 + lane];
  const float r = gcorr::wave_transpose_sum<K>(v, lane);
  const int slot = gcorr::wave_transpose_slot(lane);
  if (slot < K) out[slot] = r;
}
template <int K>
bool wts_debug_launch(int k, hipStream_t st, const float* in, float* out) {
  if (k != K) return false;
  hipLaunchKernelGGL(wts_debug_kernel<K>, dim3(1), dim3(64), 0, st, in, out);
  return true;
}
template <int... Ks>
bool wts_debug_any(std::integer_sequence<int, Ks...>, int k, hipStream_t st, const float* in, float* out) {
  return (wts_debug_launch<Ks + 1>(k, st, in, out) || ...);
}
}  // namespace

extern "C" int gc_debug_wave_transpose_sum(gc_context* ctx, int k, const float* in, float* out) {
  if (!ctx || !in || !out || k < 1 || k > 32) {
    gc_set_error("gc_debug_wave_transpose_sum: bad arguments");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  GcBuf& b = ctx->acqbuf[gc_context::ACQ_FINE_OUT];
  if (gc_buf_reserve(b, (size_t)(k * 64 + 32) * sizeof(float), false) != hipSuccess) {
    gc_set_error("gc_debug_wave_transpose_sum: device allocation failed");
    return GC_E_NOMEM;
  }
  float* d = (float*)b.p;
  GC_HIP(hipMemcpyAsync(d, in, (size_t)k * 64 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  wts_debug_any(std::make_integer_sequence<int, 32>{}, k, ctx->stream, d, d + k * 64);
  GC_HIP(hipGetLastError());
  GC_HIP(hipMemcpyAsync(out, d + k * 64, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

extern "C" long long gc_debug_first_sample_near_edge(double a, double step, long long n, double eps) {
  return gc_first_sample_near_edge(a, step, n, eps);
}

int gc_block_lowrate_level(const gc_context* ctx, const gc_block& b) {
  const HostChannel& c = ctx->ch[b.channel];
  // at most one table transition per lane-chunk (8 or 16 samples), with a safety margin
  const double s = b.code_phase_step * c.index_scale * c.mult[0];
  return (15.0 * s < 0.995) ? 2 : (7.0 * s < 0.995) ? 1 : 0;
}

int gc_block_multi_kt(const gc_context* ctx, const gc_block& b) {
  const HostChannel& c = ctx->ch[b.channel];
  // (16 - 1) samples advance the table index by 15*s entries: at most KT integers are crossed when that stays below KT
  const double s = 15.0 * b.code_phase_step * c.index_scale * c.mult[0];
  return s < 0.995 ? 1 : s < 1.995 ? 2 : s < 3.995 ? 4 : 0;
}

bool gc_block_shares_el_lane(const gc_context* ctx, const gc_block& b) {
  const HostChannel& c = ctx->ch[b.channel];
  const double v = 2.0 * b.el_spacing * c.index_scale * c.mult[0];
  return v == 1.0;  // exactly half a table entry between prompt and early / late: the lane kernel's one-ramp (HALF) variant
}

bool gc_block_shares_el(const gc_context* ctx, const gc_block& b) {
  const HostChannel& c = ctx->ch[b.channel];
  return c.arms == 1 && b.el_spacing * c.index_scale * c.mult[0] == 0.5;
}

// LDS needs of the launch being prepared ("scope" = the channels its descriptors reference): the kernels
// size their staging areas for the largest table among THOSE channels, not among everything configured.
void gc_scope_reset(gc_context* ctx) {
  ctx->max_lds_bytes = 0;
  ctx->max_stage_len = 0;
  ctx->max_arms_configured = 0;
}

void gc_scope_add(gc_context* ctx, int channel) {
  const HostChannel& c = ctx->ch[channel];
  int off = 0, maxn = 0;
  bool mixed = false;
  for (int a = 0; a < c.arms; ++a) {
    const int stage = (c.window[a] > 0) ? std::min(c.window[a], c.nent[a]) : c.nent[a];
    off += ((stage + 8 + 15) / 16) * 16;  // as DevChannel::lds_off in gc_sync_channels
    maxn = std::max(maxn, stage);
    mixed |= c.mult[a] != c.mult[0];
  }
  ctx->max_arms_configured = std::max(ctx->max_arms_configured, c.arms);
  if (mixed && gc_channel_is_derived(c)) {  // third arm derived from the second: only two tables go to LDS
    ctx->max_stage_len = std::max(ctx->max_stage_len, std::max(c.nent[0], c.nent[1]));
    return;
  }
  if (mixed) return;  // mixed-multiplier channels use the LDS-free exact kernel
  ctx->max_lds_bytes = std::max(ctx->max_lds_bytes, off);
  ctx->max_stage_len = std::max(ctx->max_stage_len, maxn);
}

int gc_fast_table_mode(const gc_context* ctx) {
  if (8 * ctx->max_lds_bytes + 512 <= 64 * 1024) return 0;   // float2 tables, one wave per workgroup
  if (2 * ctx->max_lds_bytes + 512 <= 40 * 1024 && ctx->max_arms_configured <= 2) return 1;  // int8 pairs, 4 waves share them
  return -1;
}

bool gc_fast_lds_ok(const gc_context* ctx) {
  const int m = gc_fast_table_mode(ctx);
  if (m == 0) return true;
  // WIDE is instantiated for int8 I/Q (Q/I) records and 8-sample chunks only
  return m == 1 && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL;
}

// Validates descriptors on the host; returns the largest arm count among the referenced
// channels, or a negative status.  *all_lowrate is cleared if any block needs the generic kernel.
static int validate_blocks(gc_context* ctx, int64_t n, const gc_block* b, int* all_lowrate, bool* all_share = nullptr) {
  *all_lowrate = 2;
  if (all_share) *all_share = true;
  bool any_derived = false, any_plain_mixed = false, any_three_plain = false;
  ctx->launch_derived = false;
  if (!ctx->d_if) {
    gc_set_error("no IF buffer loaded");
    return GC_E_STATE;
  }
  if (!(ctx->fs > 0)) {
    gc_set_error("sampling frequency not set (gc_set_sampling_freq)");
    return GC_E_STATE;
  }
  int max_arms = 1;
  bool seen[GC_MAX_CHANNELS] = {false};
  gc_scope_reset(ctx);
  ctx->scope_share_lane = true;
  int kt = 1, kt6 = 1;
  for (int64_t i = 0; i < n; ++i) {
    const gc_block& k = b[i];
    if (k.channel < 0 || k.channel >= GC_MAX_CHANNELS || !ctx->ch[k.channel].configured) {
      gc_set_error("block %lld: channel %d not configured", (long long)i, k.channel);
      return GC_E_STATE;
    }
    const HostChannel& c = ctx->ch[k.channel];
    if (!seen[k.channel]) {
      seen[k.channel] = true;
      gc_scope_add(ctx, k.channel);
    }
    for (int a = 0; a < c.arms; ++a) {
      if (!c.d_tab[a]) {
        gc_set_error("block %lld: channel %d arm %d has no code table", (long long)i, k.channel, a);
        return GC_E_STATE;
      }
      if (c.mult[a] != c.mult[0]) {  // mixed multipliers: exact per-sample kernel, unless the odd arm can be derived
        if (gc_channel_is_derived(c)) any_derived = true;
        else any_plain_mixed = true;
      }
      if (k.table_offset[a] < 0 || k.table_offset[a] + 3 > c.nent[a]) {
        gc_set_error("block %lld: table offset out of range", (long long)i);
        return GC_E_INVALID;
      }
    }
    double max_mult = c.mult[0];
    for (int a = 1; a < c.arms; ++a) max_mult = std::max(max_mult, c.mult[a]);
    if (k.blksize <= 0 || k.first_sample < 0 || !(k.code_phase_step > 0) ||
        !(k.el_spacing * c.index_scale * max_mult < 1.0) || !(k.el_spacing >= 0) ||
        !(k.rem_code_phase > -1.0) || !std::isfinite(k.carr_freq) || !std::isfinite(k.rem_carr_phase)) {
      gc_set_error("block %lld: invalid descriptor", (long long)i);
      return GC_E_INVALID;
    }
    if ((uint64_t)k.first_sample + (uint64_t)k.blksize > ctx->if_nsamples) {
      gc_set_error("block %lld: samples [%lld, %lld) exceed the IF buffer (%llu samples)", (long long)i,
                   (long long)k.first_sample, (long long)(k.first_sample + k.blksize),
                   (unsigned long long)ctx->if_nsamples);
      return GC_E_RANGE;  // tracking.m:241-245
    }
    // highest table index the ramps can reach must stay inside the staged window
    for (int a = 0; a < c.arms; ++a) {
      const double tmax = ((k.blksize - 1) * k.code_phase_step + k.rem_code_phase + k.el_spacing) *
                          c.index_scale * c.mult[a];
      const int stage = (c.window[a] > 0) ? std::min(c.window[a], c.nent[a]) : c.nent[a];
      const int avail = std::min(stage, c.nent[a] - k.table_offset[a]);
      if (std::ceil(tmax) > avail - 1) {
        gc_set_error("block %lld: code ramp reaches index %g beyond table (%d entries)", (long long)i,
                     std::ceil(tmax), avail);
        return GC_E_INVALID;
      }
    }
    max_arms = std::max(max_arms, c.arms);
    if (kt > 0) {  // corr_multi.hip: whole int8 tables of one ramp multiplier, one or two arms
      bool plain = c.arms <= 2;
      for (int a = 0; a < c.arms; ++a) plain = plain && c.mult[a] == c.mult[0] && c.window[a] == 0 && k.table_offset[a] == 0;
      const int need = plain ? gc_block_multi_kt(ctx, k) : 0;
      kt = need == 0 ? 0 : std::max(kt, need);
    }
    if (kt6 > 0) {  // corr_cboc.hip: whole tables, derived third arm, base ramp with at most two transitions per chunk
      bool der = c.arms == 3 && gc_channel_is_derived(c);
      for (int a = 0; a < c.arms; ++a) der = der && k.table_offset[a] == 0;
      const int need = der ? gc_block_multi_kt(ctx, k) : 0;
      kt6 = (need == 0 || need > 2) ? 0 : std::max(kt6, need);
    }
    if (*all_lowrate >= 0) *all_lowrate = std::min(*all_lowrate, gc_block_lowrate_level(ctx, k));
    if (all_share && !gc_block_shares_el(ctx, k)) *all_share = false;
    if (ctx->scope_share_lane && !gc_block_shares_el_lane(ctx, k)) ctx->scope_share_lane = false;
    if (c.arms == 3 && !gc_channel_is_derived(c)) any_three_plain = true;
  }
  // mixed ramp multipliers: the exact per-sample kernel (-1), unless every such channel's odd arm can be derived from its
  // neighbour (BOC(6,1) from BOC(1,1)) and the record is int8 I/Q: then the lane kernel's derived-arm instantiation (0)
  ctx->scope_kt = (kt >= 2 && ctx->if_layout != GC_REAL && !any_derived && !any_plain_mixed) ? kt : 0;
  ctx->scope_kt6 = (any_derived && !any_plain_mixed && !any_three_plain && ctx->if_dtype == GC_I8 && ctx->if_layout != GC_REAL) ? kt6 : 0;
  if (any_plain_mixed || (any_derived && (any_three_plain || ctx->if_dtype != GC_I8 || ctx->if_layout == GC_REAL))) {
    *all_lowrate = -1;
  } else if (any_derived) {
    *all_lowrate = 0;
    ctx->launch_derived = true;
    ctx->scope_share_lane = false;
  }
  return max_arms;
}

static int ensure(void** p, int64_t* cap, int64_t need, size_t elem) {
  if (*cap >= need) return GC_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  const int64_t n = std::max<int64_t>(need, 1024);
  if (hipMalloc(p, (size_t)n * elem) != hipSuccess) {
    gc_set_error("device allocation of %lld bytes failed", (long long)(n * (int64_t)elem));
    return GC_E_NOMEM;
  }
  *cap = n;
  return GC_OK;
}

// Number of workgroups per block for small launches: aim at >= 2 workgroups per CU.
static int choose_splits(gc_context* ctx, int64_t nblocks, const gc_block* b, int wg_threads, int spl) {
  if (nblocks * (wg_threads / 64) >= 8 * (int64_t)ctx->compute_units) return 1;
  int min_chunks = 1 << 30;
  for (int64_t i = 0; i < nblocks; ++i) min_chunks = std::min(min_chunks, b[i].blksize / spl + 1);
  // aim at ~8 wavefronts per CU, but keep at least two chunks per thread in every split
  int s = (int)((8 * (int64_t)ctx->compute_units * 64 / wg_threads + nblocks - 1) / nblocks);
  s = std::min(s, std::max(1, min_chunks / (2 * wg_threads)));
  return std::max(1, std::min(s, 64));
}

// Lane kernel (corr_lane.hip): one wavefront per (block, split) item, 16 items per workgroup sharing a block
// -> splits is a multiple of 16; aim at 16 wavefronts per CU, keep >= 8 samples per lane in every split.
int gc_lane_splits(const gc_context* ctx, int64_t nblocks, int min_blksize, int cap) {
  if (nblocks >= 2 * (int64_t)ctx->compute_units) return 1;  // one block per 16-wave workgroup, combined in LDS
  int64_t s = (16 * (int64_t)ctx->compute_units + nblocks - 1) / nblocks;
  s = std::min<int64_t>(s, std::max(1, min_blksize / 512));
  s = (s + 15) / 16 * 16;
  return (int)std::max<int64_t>(16, std::min<int64_t>(s, cap / 16 * 16));
}

extern "C" {

int gc_correlate(gc_context* ctx, int nblocks, const gc_block* blocks, double* out) {
  if (!ctx || nblocks < 0 || (nblocks > 0 && (!blocks || !out))) {
    gc_set_error("gc_correlate: bad arguments");
    return GC_E_INVALID;
  }
  if (nblocks == 0) return GC_OK;
  GC_HIP(hipSetDevice(ctx->device));
  int lowrate;
  bool share;
  const int max_arms = validate_blocks(ctx, nblocks, blocks, &lowrate, &share);
  if (max_arms < 0) return max_arms;
  int rc = gc_sync_channels(ctx);
  if (rc) return rc;
  const int fast = lowrate < 0 ? -1 : (gc_fast_lds_ok(ctx) && !ctx->force_generic) ? lowrate : 0;
  int splits = choose_splits(ctx, nblocks, blocks, fast > 0 ? 64 : 256, fast == 2 ? 16 : 8);
  if (fast > 0 && gc_fast_table_mode(ctx) == 1 && splits > 1) splits = std::max(4, (splits / 4) * 4);  // WIDE kernel
  if (fast == 0) {
    int min_blk = 1 << 30;
    for (int i = 0; i < nblocks; ++i) min_blk = std::min(min_blk, blocks[i].blksize);
    splits = gc_lane_splits(ctx, nblocks, min_blk, 256);
  }
  if ((rc = ensure((void**)&ctx->d_blocks, &ctx->d_blocks_cap, nblocks, sizeof(gc_block)))) return rc;
  if ((rc = ensure((void**)&ctx->d_out, &ctx->d_out_cap, (int64_t)nblocks * GC_OUT_STRIDE, sizeof(double)))) return rc;
  if (splits > 1 &&
      (rc = ensure((void**)&ctx->d_partial, &ctx->d_partial_cap, (int64_t)nblocks * splits * GC_OUT_STRIDE, sizeof(double))))
    return rc;
  std::vector<gc_block> marked(blocks, blocks + nblocks);
  gc_mark_tie_free(ctx, marked.data(), nblocks, fast > 0 ? 8e-6 : 0.0);
  GC_HIP(hipMemcpyAsync(ctx->d_blocks, marked.data(), sizeof(gc_block) * (size_t)nblocks, hipMemcpyHostToDevice, ctx->stream));
  rc = gc_launch_correlator(ctx, ctx->d_blocks, nblocks, splits, ctx->d_out, ctx->d_partial, max_arms, fast, 0, 0u, share);
  if (rc) return rc;
  GC_HIP(hipMemcpyAsync(out, ctx->d_out, sizeof(double) * (size_t)nblocks * GC_OUT_STRIDE, hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

int gc_replay_prepare(gc_context* ctx, int64_t nblocks, const gc_block* blocks) {
  if (!ctx || nblocks <= 0 || !blocks) {
    gc_set_error("gc_replay_prepare: bad arguments");
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  int lowrate;
  bool share;
  const int max_arms = validate_blocks(ctx, nblocks, blocks, &lowrate, &share);
  if (max_arms < 0) return max_arms;
  int rc = gc_sync_channels(ctx);
  if (rc) return rc;
  ctx->replay_share_el = share;
  ctx->replay_scope[0] = ctx->max_lds_bytes;
  ctx->replay_scope[1] = ctx->max_stage_len;
  ctx->replay_scope[2] = ctx->max_arms_configured;
  ctx->replay_share_lane = ctx->scope_share_lane;
  ctx->replay_kt = ctx->scope_kt;
  ctx->replay_kt6 = ctx->scope_kt6;
  ctx->replay_derived = ctx->launch_derived;
  ctx->replay_fast = lowrate < 0 ? -1 : (gc_fast_lds_ok(ctx) && !ctx->force_generic) ? lowrate : 0;
  GC_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->d_replay_blocks) (void)hipFree(ctx->d_replay_blocks);
  if (ctx->d_replay_out) (void)hipFree(ctx->d_replay_out);
  ctx->d_replay_blocks = nullptr;
  ctx->d_replay_out = nullptr;
  ctx->replay_nblocks = 0;
  if (hipMalloc((void**)&ctx->d_replay_blocks, sizeof(gc_block) * (size_t)nblocks) != hipSuccess ||
      hipMalloc((void**)&ctx->d_replay_out, sizeof(double) * (size_t)nblocks * GC_OUT_STRIDE) != hipSuccess) {
    gc_set_error("gc_replay_prepare: device allocation failed");
    return GC_E_NOMEM;
  }
  ctx->replay_nblocks = nblocks;
  ctx->replay_max_arms = max_arms;
  ctx->replay_min_blksize = 1 << 30;
  for (int64_t i = 0; i < nblocks; ++i) ctx->replay_min_blksize = std::min(ctx->replay_min_blksize, blocks[i].blksize);
  // channel pattern period: blocks[i].channel == blocks[i % P].channel (epoch-major replay lists)
  int period = 0;
  for (int64_t i = 1; i < nblocks && i <= GC_MAX_CHANNELS; ++i)
    if (blocks[i].channel == blocks[0].channel) {
      period = (int)i;
      break;
    }
  if (period > 0)
    for (int64_t i = 0; i < nblocks; ++i)
      if ((i >= period && blocks[i].channel != blocks[i - period].channel) || blocks[i].table_offset[0] != 0 ||
          blocks[i].table_offset[1] != 0 || blocks[i].table_offset[2] != 0) {
        period = 0;
        break;
      }
  {
    std::vector<gc_block> marked(blocks, blocks + nblocks);
    // the band the kernel that may take the list tests in: 8e-6 samples of ramp = twice the 4e-6 of corr_fast.hip and corr_multi.hip
    // (a list of derived-arm channels needs the wide band only where corr_cboc.hip takes it - gc_cboc_takes, the launcher's own
    // predicate: with it on every such list the lane kernel lost its tie-free marks on ~40 % of config 3's blocks, 0.26 -> 0.22)
    const bool cboc_list = ctx->replay_kt6 > 0 && !GC_TUNE_ENV("GC_NO_CBOC") && gc_cboc_takes(ctx, nblocks, period);
    gc_mark_tie_free(ctx, marked.data(), nblocks, (ctx->replay_kt > 0 || cboc_list || ctx->replay_fast > 0) ? 8e-6 : 0.0);
    GC_HIP(hipMemcpyAsync(ctx->d_replay_blocks, marked.data(), sizeof(gc_block) * (size_t)nblocks, hipMemcpyHostToDevice, ctx->stream));
    GC_HIP(hipStreamSynchronize(ctx->stream));
  }
  ctx->replay_period = period;
  return GC_OK;
}

int gc_replay_launch(gc_context* ctx) {
  if (!ctx || ctx->replay_nblocks <= 0) {
    gc_set_error("gc_replay_launch: nothing prepared");
    return GC_E_STATE;
  }
  GC_HIP(hipSetDevice(ctx->device));
  ctx->max_lds_bytes = ctx->replay_scope[0];
  ctx->max_stage_len = ctx->replay_scope[1];
  ctx->max_arms_configured = ctx->replay_scope[2];
  ctx->scope_share_lane = ctx->replay_share_lane;
  ctx->scope_kt = ctx->replay_kt;
  ctx->scope_kt6 = ctx->replay_kt6;
  ctx->launch_derived = ctx->replay_derived;
  int splits = 1;
  if (ctx->replay_fast == 0) {
    // lane kernel: periodic lists with enough blocks run one block per wavefront (bpw path of the launcher),
    // everything else is split 16-fold or more
    const bool periodic = ctx->replay_period > 0 && ctx->replay_nblocks >= 8 * (int64_t)ctx->compute_units;
    if (!periodic) splits = gc_lane_splits(ctx, ctx->replay_nblocks, ctx->replay_min_blksize, 256);
    if (splits > 1) {
      int rc = ensure((void**)&ctx->d_partial, &ctx->d_partial_cap, ctx->replay_nblocks * splits * GC_OUT_STRIDE, sizeof(double));
      if (rc) return rc;
    }
  } else if (ctx->replay_nblocks * (ctx->replay_fast > 0 ? 1 : 4) < 8 * (int64_t)ctx->compute_units) {
    // small replay sets: split blocks over several workgroups, scratch from d_partial
    const int wg_waves = ctx->replay_fast > 0 ? 1 : 4;
    splits = (int)std::min<int64_t>(8, (8 * (int64_t)ctx->compute_units / wg_waves + ctx->replay_nblocks - 1) / ctx->replay_nblocks);
    if (ctx->replay_fast > 0 && gc_fast_table_mode(ctx) == 1 && splits > 1) splits = std::max(4, (splits / 4) * 4);
    int rc = ensure((void**)&ctx->d_partial, &ctx->d_partial_cap, ctx->replay_nblocks * splits * GC_OUT_STRIDE, sizeof(double));
    if (rc) return rc;
  }
  return gc_launch_correlator(ctx, ctx->d_replay_blocks, ctx->replay_nblocks, splits, ctx->d_replay_out,
                              ctx->d_partial, ctx->replay_max_arms, ctx->replay_fast, ctx->replay_period, 0u,
                              ctx->replay_share_el);
}

int gc_replay_fetch(gc_context* ctx, double* out) {
  if (!ctx || !out || ctx->replay_nblocks <= 0) return GC_E_STATE;
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipMemcpyAsync(out, ctx->d_replay_out, sizeof(double) * (size_t)ctx->replay_nblocks * GC_OUT_STRIDE,
                        hipMemcpyDeviceToHost, ctx->stream));
  GC_HIP(hipStreamSynchronize(ctx->stream));
  return GC_OK;
}

int gc_timer_start(gc_context* ctx) {
  if (!ctx) return GC_E_INVALID;
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
  return GC_OK;
}

int gc_timer_stop(gc_context* ctx, double* elapsed_ms) {
  if (!ctx || !elapsed_ms) return GC_E_INVALID;
  GC_HIP(hipSetDevice(ctx->device));
  GC_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
  GC_HIP(hipEventSynchronize(ctx->ev_stop));
  float ms = 0.f;
  GC_HIP(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
  *elapsed_ms = (double)ms;
  return GC_OK;
}

}  // extern "C"
