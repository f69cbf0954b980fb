// Internal declarations shared by the translation units of libgnsscorr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gnsscorr.h"

#define GC_MAX_CHANNELS 256

// Tuning knobs.  Every A/B switch of the kernels' development (docs/KNOBS.md) is read through GC_TUNE_ENV: an environment variable in
// the tuning build (libgnsscorr_tuning.so, -DGC_TUNING=1: what the knob tests, scripts/variants.sh and the profiling scripts load
// through GC_LIB_PATH), a null pointer - and therefore no code, no string, no dependence on the process environment - in the
// library that ships.  What libgnsscorr.so itself still reads at run time: GC_TRACK_POLL_TIMEOUT_MS (how long the host waits for a
// record of a persistent kernel before it calls the run lost).
#if defined(GC_TUNING) && GC_TUNING
#define GC_TUNE_ENV(name) std::getenv(name)
#else
#define GC_TUNING 0
#define GC_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

void gc_set_error(const char* fmt, ...);

#define GC_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      gc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GC_E_HIP;                                                                     \
    }                                                                                      \
  } while (0)

// Device-side view of one tracking channel (gc_set_channel / gc_set_code).
struct DevChannel {
  const int8_t* tab[GC_MAX_ARMS];  // padded tables in HBM
  const unsigned short* tab2b[GC_MAX_ARMS];  // the same as int8 pairs (low byte c, high byte dc), WIDE fast kernel
  const float2* tab2[GC_MAX_ARMS];  // {c[k], c[k+1]-c[k]} for k = -1 .. nent+1 (nent+3 entries), fast kernel
  int32_t nent[GC_MAX_ARMS];       // entries per table
  double mult[GC_MAX_ARMS];        // per-arm ramp multiplier (1 or 6)
  double index_scale;              // R
  int32_t arms;
  int32_t stage_len[GC_MAX_ARMS];  // entries staged into LDS per arm (window or whole table)
  int32_t lds_off[GC_MAX_ARMS];    // byte offset of each staged table in LDS
  int32_t lds_bytes;               // total staged bytes (16-B aligned)
  // generic kernel: all arms interleaved as f16 with kGuard zero entries on both sides, ready to be copied
  // into LDS 16 bytes at a time (whole tables, no window); nullptr = stage from tab[] entry by entry
  const uint16_t* tabh;
  const float* tabf;   // the same as f32 (2 * tabh_bytes bytes)
  int32_t tabh_ap;     // f16 values per entry (1, 2 or 4)
  int32_t tabh_bytes;  // multiple of 16
  // arm 2 is arm 1 with a sign pattern at six times the ramp rate (BOC(6,1) next to BOC(1,1): entry k6 of its padded table
  // equals entry p = (k6 + 5) / 6 of arm 1's times (-1)^(p + k6)): the lane kernel derives it instead of reading a table
  int32_t derived;
  // derived channels stage their f32 image with four values per entry: {arm 0, arm 1, arm 1 * (-1)^entry, 0} - the third column
  // is what the derived arm multiplies by the sign of its sixfold sub-entry (corr_lane.hip)
  int32_t tabf_ap;     // f32 values per entry of tabf
  int32_t tabf_bytes;  // multiple of 16
  int32_t pad_;
};

struct HostChannel {
  bool configured = false;
  int arms = 0;
  double index_scale = 1.0;
  int8_t* d_tab[GC_MAX_ARMS] = {nullptr, nullptr, nullptr};
  float2* d_tab2[GC_MAX_ARMS] = {nullptr, nullptr, nullptr};
  unsigned short* d_tab2b[GC_MAX_ARMS] = {nullptr, nullptr, nullptr};
  int nent[GC_MAX_ARMS] = {0, 0, 0};
  double mult[GC_MAX_ARMS] = {1.0, 1.0, 1.0};
  int window[GC_MAX_ARMS] = {0, 0, 0};  // 0 = stage the whole table
  std::vector<int8_t> h_tab[GC_MAX_ARMS];  // host copies (interleaved f16 form is built at sync time)
  uint16_t* d_tabh = nullptr;
  float* d_tabf = nullptr;
  mutable int derived_state = -1;  // -1 unknown, else DevChannel::derived (gc_channel_is_derived; reset whenever the channel changes)
};

// Grow-only scratch kept by the context.  hipFree / hipHostFree wait for EVERY stream of the device, so a tracking call
// that frees its message and record buffers on the way out stalls behind the other contexts' persistent kernels
// (gc_track_multi measured fully serialised that way); these buffers are released by gc_destroy only.
struct GcBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool host = false;
};
inline hipError_t gc_buf_reserve(GcBuf& b, size_t bytes, bool host_mapped) {
  if (b.p && b.cap >= bytes && b.host == host_mapped) return hipSuccess;
  if (b.p) (void)(b.host ? hipHostFree(b.p) : hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  b.host = host_mapped;
  const size_t cap = (bytes + 4095) / 4096 * 4096;
  const hipError_t e = host_mapped ? hipHostMalloc(&b.p, cap, hipHostMallocMapped) : hipMalloc(&b.p, cap);
  if (e == hipSuccess) b.cap = cap;
  else b.p = nullptr;
  return e;
}
inline void gc_buf_free(GcBuf& b) {
  if (b.p) (void)(b.host ? hipHostFree(b.p) : hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
}

struct gc_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int compute_units = 0;
  char device_name[128] = {0};

  // IF buffer
  uint8_t* d_if = nullptr;
  bool if_owned = false;
  uint64_t if_nsamples = 0;
  uint64_t if_capacity_bytes = 0;  // bytes the kernels may read (>= payload)
  int if_dtype = GC_I8;
  int if_layout = GC_IQ;
  double fs = 0.0;

  // channels
  HostChannel ch[GC_MAX_CHANNELS];
  DevChannel* d_channels = nullptr;  // GC_MAX_CHANNELS entries
  bool channels_dirty = true;
  // scope of the launch being prepared (gc_scope_reset / gc_scope_add): LDS needs of the channels it references
  int max_lds_bytes = 0;  // int8 bytes of the largest channel (fast kernels scale it by 8 or 2)
  int max_stage_len = 0;  // longest staged table (entries), generic kernel
  int max_arms_configured = 0;
  int replay_scope[3] = {0, 0, 0};
  bool scope_share_lane = false;  // every block of the scope has 2*el_spacing*R*M an exact positive integer
  bool replay_share_lane = false;
  // multi-transition kernel (corr_multi.hip): 0 = some block or channel of the scope does not qualify, else the largest number of
  // table transitions a 16-sample chunk of any block can see (2 or 4) - int8 tables of one ramp multiplier, no windows
  int scope_kt = 0;
  int replay_kt = 0;
  // hybrid kernel (corr_cboc.hip): every channel of the scope has a derived six-fold arm and every block's base ramp sees at most
  // this many (1 or 2) table transitions per 16-sample chunk; 0 = does not qualify
  int scope_kt6 = 0;
  int replay_kt6 = 0;
  int replay_min_blksize = 0;

  // scratch for gc_correlate / gc_track
  gc_block* d_blocks = nullptr;
  int64_t d_blocks_cap = 0;
  double* d_out = nullptr;
  int64_t d_out_cap = 0;  // doubles
  double* d_partial = nullptr;
  int64_t d_partial_cap = 0;
  gc_block* h_blocks_pinned = nullptr;  // host-mapped, for the closed loop
  double* h_out_pinned = nullptr;
  void* h_tagged_pinned = nullptr;  // host-mapped tagged result slots (closed loop, fast kernel)
  int64_t tagged_cap = 0;            // slots
  int pinned_cap_blocks = 0;

  // replay
  gc_block* d_replay_blocks = nullptr;
  double* d_replay_out = nullptr;
  int64_t replay_nblocks = 0;
  int replay_max_arms = 1;
  int replay_fast = 0;
  bool replay_derived = false;
  int replay_period = 0;
  bool replay_share_el = false;  // channel pattern period of the replay list (0 = not periodic)
  bool force_generic = false;
  bool launch_derived = false;  // the launch being prepared runs the lane kernel's derived-arm instantiation
  int last_kernel = -2;  // gc_debug_last_kernel
  int last_track_mode = -1;  // gc_debug_last_track_mode: 0 launch per epoch, 1 persistent host-fed kernel, 2 device loop

  // acquisition scratch (acq_coarse.hip: AcqScratch)
  void* acq_scratch = nullptr;
  enum { ACQ_FINE_CODE = 0, ACQ_FINE_DET, ACQ_FINE_OUT, ACQ_COND_SIG, ACQ_COND_A, ACQ_COND_B, ACQ_COND_TAPS, ACQ_NBUF };
  GcBuf acqbuf[ACQ_NBUF];  // fine-frequency stage: codes, detections, per-code sums; conditioned signal of gc_acq_condition + its scratch
  long long acq_cond_n = 0;  // complex float samples in acqbuf[ACQ_COND_SIG] (gc_acq_params.source = 1 searches them)

  // gc_track_multi: this context's tracking call runs next to other contexts' on the same device.  Its persistent kernels
  // are then launched with a plain launch instead of a cooperative one (gc_launch_persistent below).
  enum { TRK_CHAN = 0, TRK_DESC, TRK_PART, TRK_ARGS, TRK_RECORDS, TRK_HDESC, TRK_CNO, TRK_NBUF };
  GcBuf trk[TRK_NBUF];  // gc_track / gc_track_device: channel state, descriptor and partial-sum messages, arguments, records
  GcBuf nav[3];  // gc_sync_xcorr: prompt stream, pattern, result
  double* cno_out = nullptr;  // gc_set_cno_output: caller-owned C/N0 buffer of the next tracking calls
  long long cno_cap = 0;
  int persist_member_cap = 0;  // gc_track_device: team size limit of the retry after a grid that did not fit the device (0: none)
  bool concurrent_jobs = false;
  int concurrent_channels = 0;  // channels of all jobs on this device (sizes the persistent kernels' teams)
};

// Launch of a persistent (host-fed or device-loop) kernel whose workgroups wait for each other's messages and therefore must
// all be resident.  Alone on the device: a cooperative launch, the runtime guarantees residency.  Next to other contexts'
// persistent kernels (gc_track_multi): cooperative launches of different streams do not overlap, and a host-fed kernel that
// waits behind another one never gets its descriptors consumed - so those are plain launches, admitted by a per-device
// ledger of the persistent kernels in flight (multi.hip): a grid goes ahead only when, together with the grids already
// resident, every one of its workgroups finds room whatever order the dispatcher places them in; otherwise the call returns
// hipErrorCooperativeLaunchTooLarge - what a cooperative launch answers to a grid that does not fit - and the caller falls
// back (gc_track: a launch per epoch; gc_track_device: GC_E_UNSUPPORTED, i.e. gc_track).  GC_PERSIST_COOP=0/1 forces either
// launch kind (experiments).  gc_persistent_done takes the context's entry out of the ledger when its kernel has ended.
hipError_t gc_launch_persistent(gc_context* ctx, const void* fn, dim3 grid, dim3 block, void** args, unsigned int smem);
void gc_persistent_done(gc_context* ctx);
// internal status of the persistent launchers, never returned through the C-ABI: the grid did not fit - the ONE refusal a caller may
// answer with smaller teams (a structural GC_E_UNSUPPORTED - tables or arms without an instantiation - no team size can fix)
#define GC_E_NOFIT (-100)
#define GC_PERSIST(call)                                                                                     \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ == hipErrorCooperativeLaunchTooLarge) {                                                           \
      (void)hipGetLastError();                                                                               \
      gc_set_error("the persistent kernel's grid does not fit the device next to the kernels in flight");    \
      return GC_E_NOFIT;                                                                                     \
    }                                                                                                        \
    if (e_ != hipSuccess) {                                                                                  \
      gc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);               \
      return GC_E_HIP;                                                                                       \
    }                                                                                                        \
  } while (0)

// gc_track over one window of a record (gc_track_resume / gc_track_file, track.hip + stream.hip)
struct GcTrackResume {
  gc_channel_state* state = nullptr;  // [nch] in (resume) / out
  bool resume = false;                // state holds the end state of the previous window
  bool pause_at_end = false;          // more of the record follows this window
  int64_t origin = 0;                 // record index of the IF buffer's first sample
  bool paused = false;                // out: stopped because a block did not fit this window
};
int gc_track_window(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init, double* out,
                    int32_t* epochs_done, GcTrackResume* r);

// trackResults.CNo.VSMValue of finished records on the host (the host-closed loops; the device loop has its own copy in devloop.h)
void gc_fill_cno_host(gc_context* ctx, const gc_track_params* p, int nch, const double* out, const int32_t* epochs_done);

int gc_bytes_per_sample(int dtype, int layout);
void gc_acq_free(gc_context* ctx);  // acq_coarse.hip
int gc_sync_channels(gc_context* ctx);
void gc_scope_reset(gc_context* ctx);
void gc_scope_add(gc_context* ctx, int channel);
// Launches the correlator for `nblocks` descriptors already on the device.
int gc_launch_correlator(gc_context* ctx, const gc_block* d_blocks, int64_t nblocks, int splits,
                         double* d_out, double* d_partial, int max_arms, int fast, int period = 0,
                         unsigned int notify_tag = 0, bool share_el = false);
// el_spacing * R * M == 1/2 exactly: early and late ramps differ by one whole table entry
bool gc_block_shares_el(const gc_context* ctx, const gc_block& b);
// 2 * el_spacing * R * M == 1 exactly: early, prompt and late taps read table entries k and k + 1 of ONE ramp (lane kernel, HALF)
bool gc_block_shares_el_lane(const gc_context* ctx, const gc_block& b);
// Kernel class a block qualifies for: 0 = generic only, 1 = fast kernel with 8-sample lane-chunks,
// 2 = fast kernel with 16-sample lane-chunks (at most one table transition per chunk and tap).
int gc_block_lowrate_level(const gc_context* ctx, const gc_block& b);
bool gc_fast_lds_ok(const gc_context* ctx);
int gc_lane_splits(const gc_context* ctx, int64_t nblocks, int min_blksize, int cap);
int64_t gc_first_sample_near_edge(double a, double step, int64_t n, double eps);
void gc_mark_tie_free(const gc_context* ctx, gc_block* b, int64_t n, double eps_unit_steps);
// 0 = float2 tables / single-wave workgroups, 1 = WIDE (int8 pairs, four waves), -1 = tables too large for the fast kernel
int gc_fast_table_mode(const gc_context* ctx);
// corr_multi.hip
int gc_multi_table_bytes(int max_entries, int arms);
int gc_block_multi_kt(const gc_context* ctx, const gc_block& b);  // 1, 2, 4 transitions per 16-sample chunk at most; 0 = more
bool gc_channel_is_derived(const HostChannel& c);  // cached in HostChannel::derived_state
