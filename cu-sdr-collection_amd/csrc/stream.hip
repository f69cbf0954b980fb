// stream.hip — tracking a record that does not fit the device: window by window (streaming ingest).
//
// The reference never holds its record: tracking.m freads one block per epoch and channel (tracking.m:226-245), so a file of
// any length works (postProcessing.m:61-96).  gc_track wants the record resident in HBM.  Here the file is cut into windows
// of `window_samples`; two device buffers alternate: while the tracking loop runs on window k, a reader thread preads
// window k + 1 through a pinned staging buffer and uploads it on a stream of its own.  Consecutive windows overlap by a
// margin of three nominal blocks, so that whatever block a channel was about to read when window k ran out lies inside
// window k + 1 (the channels stop in lock step, within one block of each other).  Window origins are multiples of 256
// samples: a block's address alignment - hence the order of the kernels' float additions - is the same as with the whole
// record resident, and the results are bit-identical to gc_track's (tests/test_gpu_stream.py).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <climits>

#include "gc_internal.h"

extern "C" int gc_track_resume(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
                               gc_channel_state* state, int flags, int64_t origin, double* out, int32_t* epochs_done,
                               int32_t* paused) {
  if (!state || origin < 0) {
    gc_set_error("gc_track_resume: bad arguments");
    return GC_E_INVALID;
  }
  GcTrackResume r;
  r.state = state;
  r.resume = (flags & GC_TRACK_RESUME) != 0;
  r.pause_at_end = (flags & GC_TRACK_PAUSE_AT_END) != 0;
  r.origin = origin;
  const int rc = gc_track_window(ctx, p, nch, init, out, epochs_done, &r);
  if (paused) *paused = r.paused ? 1 : 0;
  return rc;
}

namespace {

struct WindowLoader {
  int fd = -1;
  int device = 0;
  uint64_t file_offset0 = 0;  // byte offset of record sample 0
  uint64_t bps = 0;
  hipStream_t stream = nullptr;
  uint8_t* stage = nullptr;  // two pinned halves
  size_t chunk = 32u << 20;
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::string error;

  // record samples [first, first + n) -> dst (device).  Runs on the calling thread.
  bool load(uint8_t* dst, uint64_t first, uint64_t n) {
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    const uint64_t total = n * bps;
    uint64_t done = 0;
    int half = 0;
    bool used[2] = {false, false};
    while (done < total) {
      const size_t want = (size_t)std::min<uint64_t>(chunk, total - done);
      if (used[half] && hipEventSynchronize(ev[half]) != hipSuccess) return fail("hipEventSynchronize");
      size_t got = 0;
      while (got < want) {
        const ssize_t k = pread(fd, stage + half * chunk + got, want - got, (off_t)(file_offset0 + first * bps + done + got));
        if (k <= 0) return fail("short read");
        got += (size_t)k;
      }
      if (hipMemcpyAsync(dst + done, stage + half * chunk, want, hipMemcpyHostToDevice, stream) != hipSuccess) return fail("H2D copy");
      if (hipEventRecord(ev[half], stream) != hipSuccess) return fail("hipEventRecord");
      used[half] = true;
      done += want;
      half ^= 1;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("hipStreamSynchronize");
    return true;
  }
  bool fail(const char* what) {
    error = what;
    return false;
  }
};

}  // namespace

extern "C" int gc_track_file(gc_context* ctx, const char* path, uint64_t skip_bytes, int dtype, int layout, uint64_t window_samples,
                             const gc_track_params* p, int nch, const gc_channel_init* init, double* out, int32_t* epochs_done) {
  if (!ctx || !path || !p || nch <= 0 || nch > GC_MAX_CHANNELS || !init || !out || !epochs_done || p->n_epochs <= 0 ||
      !(p->sampling_freq > 0) || !(p->code_length > 0) || !(p->code_freq_basis > 0)) {
    gc_set_error("gc_track_file: bad arguments");
    return GC_E_INVALID;
  }
  const int bpsi = gc_bytes_per_sample(dtype, layout);
  if (bpsi <= 0) {
    gc_set_error("gc_track_file: unknown sample format");
    return GC_E_INVALID;
  }
  const uint64_t bps = (uint64_t)bpsi;
  const int fd = open(path, O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0) {
    if (fd >= 0) close(fd);
    gc_set_error("Unable to read file %s", path);  // postProcessing.m:157
    return GC_E_INVALID;
  }
  if ((uint64_t)sb.st_size <= skip_bytes) {
    close(fd);
    gc_set_error("gc_track_file: skip (%llu) beyond end of file", (unsigned long long)skip_bytes);
    return GC_E_RANGE;
  }
  const uint64_t total = ((uint64_t)sb.st_size - skip_bytes) / bps;  // record samples
  if (window_samples == 0 || window_samples >= total) {  // fits: the resident path
    close(fd);
    int rc = gc_open_if_file(ctx, path, skip_bytes, 0, dtype, layout);
    if (rc) return rc;
    return gc_track(ctx, p, nch, init, out, epochs_done);
  }
  // nominal block length (one code period) with head room for the code NCO; the overlap between windows
  const double block = p->code_length / p->code_freq_basis * p->sampling_freq;
  const uint64_t margin = ((uint64_t)(3.0 * block * 1.01) + 255 + 256) / 256 * 256;
  const uint64_t W = window_samples / 256 * 256;
  if (W < 4 * margin) {
    close(fd);
    gc_set_error("gc_track_file: a window of %llu samples is too short for blocks of %.0f samples (needs at least %llu)",
                 (unsigned long long)window_samples, block, (unsigned long long)(4 * margin));
    return GC_E_INVALID;
  }
  GC_HIP(hipSetDevice(ctx->device));
  const uint64_t stride = W - margin;
  uint8_t* dbuf[2] = {nullptr, nullptr};
  WindowLoader ld;
  ld.fd = fd;
  ld.device = ctx->device;
  ld.file_offset0 = skip_bytes;
  ld.bps = bps;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_if == dbuf[0] || ctx->d_if == dbuf[1]) {  // the context must not keep a pointer into a freed window
      ctx->d_if = nullptr;
      ctx->if_nsamples = 0;
      ctx->if_capacity_bytes = 0;
    }
    for (uint8_t* b : dbuf)
      if (b) (void)hipFree(b);
    if (ld.stage) (void)hipHostFree(ld.stage);
    for (hipEvent_t e : ld.ev)
      if (e) (void)hipEventDestroy(e);
    if (ld.stream) (void)hipStreamDestroy(ld.stream);
    close(fd);
  };
  if (hipMalloc((void**)&dbuf[0], W * bps + 4096) != hipSuccess || hipMalloc((void**)&dbuf[1], W * bps + 4096) != hipSuccess ||
      hipHostMalloc((void**)&ld.stage, 2 * ld.chunk, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&ld.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ld.ev[0]) != hipSuccess ||
      hipEventCreate(&ld.ev[1]) != hipSuccess) {
    cleanup();
    gc_set_error("gc_track_file: could not allocate two windows of %llu samples", (unsigned long long)W);
    return GC_E_NOMEM;
  }
  const int n_total = p->n_epochs;
  std::fill(out, out + (size_t)nch * GC_TRK_NFIELDS * n_total, 0.0);
  std::vector<gc_channel_state> state((size_t)nch);
  std::memset(state.data(), 0, sizeof(gc_channel_state) * (size_t)nch);
  std::vector<int32_t> done_total((size_t)nch, 0), done_w((size_t)nch, 0);
  // epochs one call can possibly run inside a window
  const int epw = (int)std::min<uint64_t>((uint64_t)n_total, (uint64_t)((double)W / (block * 0.98)) + 2);
  std::vector<double> out_w((size_t)nch * GC_TRK_NFIELDS * epw);

  auto window_len = [&](uint64_t k) { return std::min<uint64_t>(W, total - k * stride); };
  auto is_last = [&](uint64_t k) { return k * stride + W >= total; };
  // The first window is the one the earliest channel starts in (settings.skipNumberOfBytes may put that gigabytes into the
  // file: reading and uploading every window of the skipped part, each pausing with zero epochs, would be the price of
  // starting at window 0).  Window origins are multiples of the 256-sample-aligned stride, so block alignment is as before.
  int64_t first_start = INT64_MAX;
  for (int c = 0; c < nch; ++c) first_start = std::min<int64_t>(first_start, p->skip_samples + init[c].code_phase - 1);
  uint64_t k0 = first_start > 0 ? (uint64_t)first_start / stride : 0;
  while (k0 > 0 && k0 * stride >= total) --k0;  // a start beyond the record: the last window reports the short read
  if (!ld.load(dbuf[k0 & 1], k0 * stride, window_len(k0))) {
    cleanup();
    gc_set_error("gc_track_file: %s while reading window %llu", ld.error.c_str(), (unsigned long long)k0);
    return GC_E_RANGE;
  }
  int rc = GC_OK, last_rc = GC_OK;
  bool first_call = true, finished = false;
  for (uint64_t k = k0; !finished; ++k) {
    const bool last = is_last(k);
    std::thread reader;
    std::atomic<bool> reader_ok{true};
    if (!last) reader = std::thread([&, k]() { reader_ok = ld.load(dbuf[(k + 1) & 1], (k + 1) * stride, window_len(k + 1)); });
    rc = gc_attach_if(ctx, dbuf[k & 1], window_len(k), dtype, layout);
    while (rc == GC_OK) {
      int remaining = n_total;
      for (int c = 0; c < nch; ++c) remaining = std::min(remaining, n_total - done_total[c]);  // lock step: the same for all running channels
      bool any_running = false;
      for (int c = 0; c < nch; ++c) any_running |= first_call || (state[c].status == 0 && done_total[c] < n_total);
      if (!any_running || remaining <= 0) {
        finished = true;
        break;
      }
      gc_track_params pw = *p;
      pw.n_epochs = std::min(remaining, epw);
      GcTrackResume r;
      r.state = state.data();
      r.resume = !first_call;
      r.pause_at_end = !last;
      r.origin = (int64_t)(k * stride);
      last_rc = gc_track_window(ctx, &pw, nch, init, out_w.data(), done_w.data(), &r);
      first_call = false;
      if (last_rc != GC_OK && last_rc != GC_E_RANGE && last_rc != GC_E_INVALID) {  // RANGE / INVALID: a channel ended, records are valid
        rc = last_rc;
        break;
      }
      for (int c = 0; c < nch; ++c) {
        for (int f = 0; f < GC_TRK_NFIELDS; ++f)
          std::copy(out_w.data() + ((size_t)c * GC_TRK_NFIELDS + f) * pw.n_epochs,
                    out_w.data() + ((size_t)c * GC_TRK_NFIELDS + f) * pw.n_epochs + done_w[c],
                    out + ((size_t)c * GC_TRK_NFIELDS + f) * n_total + done_total[c]);
        done_total[c] += done_w[c];
      }
      if (last_rc != GC_OK) {  // end of the record (or a diverged NCO): gc_track's own early-return semantics apply
        finished = true;
        break;
      }
      if (r.paused) {
        // every channel's next block must start inside the next window
        for (int c = 0; c < nch; ++c)
          if (state[c].status == 0 && (uint64_t)state[c].next_sample < (k + 1) * stride) {
            gc_set_error("gc_track_file: channel slot %d fell %llu samples behind the next window (margin %llu)", c,
                         (unsigned long long)((k + 1) * stride - (uint64_t)state[c].next_sample), (unsigned long long)margin);
            rc = GC_E_INVALID;
          }
        break;  // next window
      }
      // not paused and no error: pw.n_epochs epochs done inside this window; go on in the same window
    }
    if (reader.joinable()) reader.join();
    if (rc == GC_OK && !last && !reader_ok) {
      gc_set_error("gc_track_file: %s while reading window %llu", ld.error.c_str(), (unsigned long long)(k + 1));
      rc = GC_E_RANGE;
    }
    if (rc != GC_OK) break;
    if (last && !finished) finished = true;
  }
  cleanup();
  if (rc != GC_OK) return rc;
  // the reference returns from the whole function at the first short read: channels after that one are never run
  int first_aborted = nch;
  for (int c = 0; c < nch; ++c)
    if (state[c].status == 2) {
      first_aborted = c;
      break;
    }
  for (int c = 0; c < nch; ++c) {
    if (c > first_aborted) {
      double* o = out + (size_t)c * GC_TRK_NFIELDS * n_total;
      std::fill(o, o + (size_t)GC_TRK_NFIELDS * n_total, 0.0);
      epochs_done[c] = 0;
    } else {
      epochs_done[c] = done_total[c];
    }
  }
  gc_fill_cno_host(ctx, p, nch, out, epochs_done);
  return last_rc;
}
