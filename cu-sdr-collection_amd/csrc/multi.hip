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
    if (std::getenv("GC_MULTI_OWN_STREAMS") == nullptr && hipSetDevice(device) == hipSuccess) {
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
  // Persistent kernels of jobs on the same device must be resident together (gc_launch_persistent): admit them only when
  // the device has room to spare - at most 32 workgroups per channel (track.hip) of at most 8 wavefronts each, against
  // 8+ wavefront slots per SIMD; above that the jobs still run, launching their correlators per epoch.
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
    j.ctx->concurrent_jobs = false;
    return j.status;
  }
  // jobs of one device onto streams of different hardware queues (multi_streams above); a device's fourth and later job keeps
  // its context's own stream
  std::vector<hipStream_t> own((size_t)njobs, nullptr);
  for (int i = 0; i < njobs; ++i) {
    int idx = 0;
    for (int k = 0; k < i; ++k) idx += jobs[k].ctx->device == jobs[i].ctx->device;
    MultiStreams* ms = multi_streams(jobs[i].ctx->device);
    if (ms && idx < ms->n && jobs[i].ctx->concurrent_jobs) {
      GC_HIP(hipSetDevice(jobs[i].ctx->device));
      GC_HIP(hipStreamSynchronize(jobs[i].ctx->stream));
      own[i] = jobs[i].ctx->stream;
      jobs[i].ctx->stream = ms->s[idx];
    }
  }
  std::vector<std::thread> workers;
  workers.reserve((size_t)njobs);
  const bool timing = std::getenv("GC_TRACK_TIMING") != nullptr;
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
    if (own[i]) {
      (void)hipSetDevice(jobs[i].ctx->device);
      (void)hipStreamSynchronize(jobs[i].ctx->stream);
      jobs[i].ctx->stream = own[i];
    }
    jobs[i].ctx->concurrent_jobs = false;
    // a short read of one package (GC_E_RANGE, partial records returned) must not hide a failure of another
    if (jobs[i].status != GC_OK && (first == GC_OK || first == GC_E_RANGE)) {
      first = jobs[i].status;
      gc_set_error("gc_track_multi: job %d: %s", i, jobs[i].error);
    }
  }
  return first;
}
