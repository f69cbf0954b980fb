// multi.hip — gc_track_multi: several of the reference's tracking() calls at the same time.
//
// The reference tracks the channels of ONE package per call (GPS/GPS_L1CA/include/tracking.m:133 loops over
// settings.numberOfChannels with one `settings`); a receiver for several signals runs the packages one after the other
// (each has its own init.m / postProcessing.m).  BASELINE config 5 puts channels of different packages on one GPU: GPS L1 C/A,
// Galileo E1 and BDS B1C channels read the SAME L1-band record, L5 / E5a / B2a channels another one.  Channels are independent
// (SURVEY.md §8e), so the packages' loops can run side by side: one context per (record, package) - its own stream, code
// tables, descriptor and result buffers -, one host thread per context for the loop closure (tracking.m:302-335), records
// shared between contexts without a copy (gc_share_if).  Contexts may also live on different devices: a single-process host
// (MATLAB's interpreter thread behind the MEX gateway) drives all GPUs of a node through one call.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gc_internal.h"

extern "C" int gc_share_if(gc_context* dst, gc_context* src) {
  if (!dst || !src || dst == src) {
    gc_set_error("gc_share_if: two different contexts are needed");
    return GC_E_INVALID;
  }
  if (!src->d_if) {
    gc_set_error("gc_share_if: the source context holds no IF record");
    return GC_E_STATE;
  }
  if (dst->device != src->device) {
    gc_set_error("gc_share_if: contexts on devices %d and %d - a record is shared within one GPU (across GPUs: broadcast it, INTEGRATION.md)",
                 dst->device, src->device);
    return GC_E_INVALID;
  }
  int rc = gc_attach_if(dst, src->d_if, src->if_nsamples, src->if_dtype, src->if_layout);
  if (rc) return rc;
  dst->if_capacity_bytes = src->if_capacity_bytes;  // the owner's padding is readable through the alias too
  if (src->fs > 0) dst->fs = src->fs;
  return GC_OK;
}

// ---- admission of persistent kernels that share a device (gc_launch_persistent, gc_internal.h) --------------------------------
// Workgroups of a persistent kernel spin on their team mates' messages, so a grid that is only partly resident never ends, and
// two partly resident grids block each other for good.  Workgroup b of a launch goes to XCD b mod 8 and there to any CU with room
// left.  A workgroup of kernel i takes at most 1 / occ_i of every CU resource (occ_i = workgroups of that kernel per CU when it is
// alone: hipOccupancyMaxActiveBlocksPerMultiprocessor with its LDS size), so per XCD the set of kernels in flight loads the CUs
// with sum_i w_i / occ_i CU-equivalents, w_i = ceil(grid_i / 8).  A workgroup that needs 1 / occ of a CU is shut out of a CU
// only while that CU carries more than 1 - 1 / occ; with r = the largest 1 / occ_i of the set, fewer than load / (1 - r) CUs can
// be that full - and when some kernel needs whole CUs (r = 1), a CU is shut for it as soon as it holds ANY workgroup, so the
// workgroups of the set must not outnumber the XCD's CUs.  Under either condition every workgroup of every admitted grid finds a
// CU whatever the dispatch order.  The ledger knows this process' kernels; ranks of other processes on the same device are
// outside it (bench.py serialises their closed loops when GC_BENCH_DEVICE puts several ranks on one GPU).
namespace {
struct ResidentGrid {
  const gc_context* ctx;
  int wg_per_xcd;
  int occ;
};
std::mutex g_ledger_mu;
std::vector<ResidentGrid> g_ledger[64];

// XCDs of the device: 8 on MI300X / MI355X (256-304 CUs); smaller parts are treated as one XCD per 32 CUs (the model only needs an upper
// bound of the workgroups that can meet on one XCD)
int xcds_of(int cus) { return cus >= 128 ? 8 : std::max(1, cus / 32); }

bool ledger_fits(const std::vector<ResidentGrid>& set, const ResidentGrid& add, int cus) {
  const double cu_per_xcd = std::max(1, cus / xcds_of(cus));
  double load = (double)add.wg_per_xcd / add.occ, rmax = 1.0 / add.occ;
  long long wgs = add.wg_per_xcd;
  for (const ResidentGrid& g : set) {
    load += (double)g.wg_per_xcd / g.occ;
    rmax = std::max(rmax, 1.0 / g.occ);
    wgs += g.wg_per_xcd;
  }
  if (rmax >= 1.0) return (double)wgs <= cu_per_xcd;
  return load <= cu_per_xcd * (1.0 - rmax);
}
}  // namespace

hipError_t gc_launch_persistent(gc_context* ctx, const void* fn, dim3 grid, dim3 block, void** args, unsigned int smem) {
  bool coop = !ctx->concurrent_jobs;
  if (const char* e = GC_TUNE_ENV("GC_PERSIST_COOP")) coop = std::atoi(e) != 0;
  if (coop) {
    if (GC_TUNE_ENV("GC_TRACK_DEBUG")) {
      int occ = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)block.x, smem);
      std::fprintf(stderr, "gc_launch_persistent: cooperative, grid %u x %u threads, %u B of LDS, occupancy %d workgroups per CU x %d CUs\n", grid.x, block.x, smem, occ,
                   ctx->compute_units);
    }
    return hipLaunchCooperativeKernel(fn, grid, block, args, smem, ctx->stream);
  }
  if (ctx->device < 0 || ctx->device >= 64) return hipErrorInvalidDevice;
  int occ = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)block.x, smem);
  if (e != hipSuccess) return e;
  if (const char* ev = GC_TUNE_ENV("GC_PERSIST_OCC")) occ = std::min(occ, std::atoi(ev));  // tests: pretend the kernel needs more of a CU
  if (occ < 1) return hipErrorCooperativeLaunchTooLarge;
  const int nx = xcds_of(ctx->compute_units);
  const ResidentGrid mine{ctx, (int)((grid.x + nx - 1) / nx), occ};
  std::lock_guard<std::mutex> lock(g_ledger_mu);
  std::vector<ResidentGrid>& set = g_ledger[ctx->device];
  set.erase(std::remove_if(set.begin(), set.end(), [&](const ResidentGrid& g) { return g.ctx == ctx; }), set.end());
  if (!ledger_fits(set, mine, ctx->compute_units)) return hipErrorCooperativeLaunchTooLarge;
  e = hipLaunchKernel(fn, grid, block, args, smem, ctx->stream);
  if (e == hipSuccess) set.push_back(mine);
  return e;
}

void gc_persistent_done(gc_context* ctx) {
  if (!ctx || ctx->device < 0 || ctx->device >= 64) return;
  std::lock_guard<std::mutex> lock(g_ledger_mu);
  std::vector<ResidentGrid>& set = g_ledger[ctx->device];
  set.erase(std::remove_if(set.begin(), set.end(), [&](const ResidentGrid& g) { return g.ctx == ctx; }), set.end());
}

namespace {
// Streams for the concurrent jobs of one device.  Two persistent kernels overlap only when their streams sit on different
// hardware queues, and the runtime maps streams to its few queues (GPU_MAX_HW_QUEUES, 4) by its own bookkeeping: whether the
// contexts' own streams collide depended on how many streams the process had created before (bench.py: config 4's two jobs
// took 1.32 s when they shared a queue and 0.66 s when they did not; GPU_MAX_HW_QUEUES = 2 / 8 moved the collision to other
// configs).  Streams of different PRIORITY come from different queue pools, so three streams - normal, high, low - are on
// three different queues whatever else the process did.  Created once per device, never destroyed.
struct MultiStreams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  int n = 0;
};
MultiStreams* multi_streams(int device) {
  static std::mutex mu;
  static MultiStreams pool[64];
  static bool made[64] = {false};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  MultiStreams& m = pool[device];
  if (!made[device]) {
    made[device] = true;
    if (GC_TUNE_ENV("GC_MULTI_OWN_STREAMS") == nullptr && hipSetDevice(device) == hipSuccess) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      const int prio[3] = {0, greatest, least};
      for (int k = 0; k < 3; ++k) {
        if (k > 0 && (prio[k] == 0 || (k == 2 && prio[2] == prio[1]))) break;  // fewer distinct priorities than three
        if (hipStreamCreateWithPriority(&m.s[m.n], hipStreamNonBlocking, prio[k]) != hipSuccess) break;
        ++m.n;
      }
      (void)hipGetLastError();
    }
  }
  return &m;
}
}  // namespace

extern "C" int gc_track_multi(int njobs, gc_track_job* jobs) {
  if (njobs <= 0 || !jobs) {
    gc_set_error("gc_track_multi: bad arguments");
    return GC_E_INVALID;
  }
  for (int i = 0; i < njobs; ++i) {
    gc_track_job& j = jobs[i];
    j.status = GC_OK;
    j.error[0] = 0;
    if (!j.ctx || !j.params || !j.init || !j.out || !j.epochs_done || j.nch <= 0) {
      gc_set_error("gc_track_multi: job %d: bad arguments", i);
      return GC_E_INVALID;
    }
    for (int k = 0; k < i; ++k)
      if (jobs[k].ctx == j.ctx) {
        gc_set_error("gc_track_multi: jobs %d and %d share a context; a context serves one tracking call at a time", k, i);
        return GC_E_INVALID;
      }
  }
  // Persistent kernels of jobs on the same device must be resident together: each context's launch goes through the ledger of
  // gc_launch_persistent, which admits a grid only when it fits next to those in flight; a job whose kernel is refused runs its
  // loop with a launch per epoch (gc_track) - slower, never stuck.  Teams are sized for all channels of the device.
  for (int i = 0; i < njobs; ++i) {
    int same = 0, channels = 0;
    for (int k = 0; k < njobs; ++k)
      if (jobs[k].ctx->device == jobs[i].ctx->device) {
        ++same;
        channels += jobs[k].nch;
      }
    jobs[i].ctx->concurrent_jobs = same > 1;
    jobs[i].ctx->concurrent_channels = channels;
  }
  // Whatever way this call ends, every context leaves it with its own stream and out of multi-job state (a later gc_track on a
  // context left in that state would launch its persistent kernel plainly, teams spread, on a stream shared with other contexts).
  struct Restore {
    int njobs;
    gc_track_job* jobs;
    std::vector<hipStream_t> own;
    ~Restore() {
      for (int i = 0; i < njobs; ++i) {
        gc_context* c = jobs[i].ctx;
        if (own[(size_t)i]) {
          (void)hipSetDevice(c->device);
          (void)hipStreamSynchronize(c->stream);
          c->stream = own[(size_t)i];
        }
        c->concurrent_jobs = false;
        c->concurrent_channels = 0;
        gc_persistent_done(c);
      }
    }
  } restore{njobs, jobs, std::vector<hipStream_t>((size_t)njobs, nullptr)};
  // Code-table uploads (and the frees of the tables they replace: hipFree waits for every stream of the device) happen here,
  // one context after the other, before any persistent kernel is running.
  for (int i = 0; i < njobs; ++i) {
    GC_HIP(hipSetDevice(jobs[i].ctx->device));
    const int rc = gc_sync_channels(jobs[i].ctx);
    if (rc) return rc;
  }
  if (njobs == 1) {  // nothing to overlap: run on the caller's thread
    gc_track_job& j = jobs[0];
    j.status = j.device_loop ? gc_track_device(j.ctx, j.params, j.nch, j.init, j.out, j.epochs_done) : GC_E_UNSUPPORTED;
    if (!j.device_loop || j.status == GC_E_UNSUPPORTED) j.status = gc_track(j.ctx, j.params, j.nch, j.init, j.out, j.epochs_done);
    if (j.status != GC_OK) std::snprintf(j.error, sizeof j.error, "%s", gc_last_error());
    return j.status;
  }
  // jobs of one device onto streams of different hardware queues (multi_streams above); a device's fourth and later job keeps
  // its context's own stream
  for (int i = 0; i < njobs; ++i) {
    int idx = 0;
    for (int k = 0; k < i; ++k) idx += jobs[k].ctx->device == jobs[i].ctx->device;
    MultiStreams* ms = multi_streams(jobs[i].ctx->device);
    if (ms && idx < ms->n && jobs[i].ctx->concurrent_jobs) {
      GC_HIP(hipSetDevice(jobs[i].ctx->device));
      GC_HIP(hipStreamSynchronize(jobs[i].ctx->stream));
      restore.own[(size_t)i] = jobs[i].ctx->stream;
      jobs[i].ctx->stream = ms->s[idx];
    }
  }
  std::vector<std::thread> workers;
  workers.reserve((size_t)njobs);
  const bool timing = GC_TUNE_ENV("GC_TRACK_TIMING") != nullptr;
  const auto t_call = std::chrono::steady_clock::now();
  for (int i = 0; i < njobs; ++i)
    workers.emplace_back([&jobs, i, timing, t_call]() {
      gc_track_job& j = jobs[i];
      const auto t0 = std::chrono::steady_clock::now();
      int st = GC_E_UNSUPPORTED;
      if (j.device_loop) st = gc_track_device(j.ctx, j.params, j.nch, j.init, j.out, j.epochs_done);
      if (st == GC_E_UNSUPPORTED) st = gc_track(j.ctx, j.params, j.nch, j.init, j.out, j.epochs_done);
      j.status = st;
      if (st != GC_OK) std::snprintf(j.error, sizeof j.error, "%s", gc_last_error());  // the error text is per thread
      if (timing) {
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "gc_track_multi: job %d (%d channels, %d epochs) ran from %.3f to %.3f ms after the call\n", i, j.nch, j.params->n_epochs,
                     std::chrono::duration<double, std::milli>(t0 - t_call).count(), std::chrono::duration<double, std::milli>(t1 - t_call).count());
      }
    });
  for (auto& w : workers) w.join();
  int first = GC_OK;
  for (int i = 0; i < njobs; ++i) {
    // a short read of one package (GC_E_RANGE, partial records returned) must not hide a failure of another
    if (jobs[i].status != GC_OK && (first == GC_OK || first == GC_E_RANGE)) {
      first = jobs[i].status;
      gc_set_error("gc_track_multi: job %d: %s", i, jobs[i].error);
    }
  }
  return first;
}
