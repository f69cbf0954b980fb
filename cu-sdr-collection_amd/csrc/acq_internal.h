// acq_internal.h - what the translation units of the acquisition share (csrc/acq_fft.hip: plans, pass kernels, launch_pass;
// acq_coarse.hip: the carrier-per-bin searches, peak reductions, scratch; acq_shift.hip: the circshift family; acq_fine.hip: fine
// frequency stages; acq_cond.hip: input conditioning; acq_guard.hip: float64 re-evaluation of single cells).
// Reference: GPS/GPS_L1CA/include/acquisition.m:116-260 and its per-package variants.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "acq_guard.h"
#include "gc_internal.h"

namespace gcacq {

constexpr int kGuardListCap = 4096;  // cells within gc_acq_tie_eps of a PRN's winner that the guard's slow path re-evaluates at most
constexpr int kMaxRadices = 12;
constexpr int kMaxPassLen = 2048;  // longest vector of a pass: one tile of 2048 complex values (choose_cols), i.e. transforms of up to 2048 x 2048 points
constexpr int kFftThreads = 256;
constexpr int kFftSlots = 8;  // tile elements per thread at most: L*C <= kFftSlots * kFftThreads

struct SubPlan {
  int len;
  int nrad;
  int rad[kMaxRadices];
};

struct Plan {
  int n, n1, n2;  // n = n1 * n2; n1 = column length (stride n2), n2 = row length (contiguous)
  SubPlan p1, p2;
};

// Division of a small wave-uniform number by a run-time constant of the launch (hops per bin, bins per spectrum, hop groups ...) as one
// multiply-high: q = (x * mul) >> 32 with mul = floor(2^32 / d) + 1 is floor(x / d) whenever x * d < 2^32 (launch_pass checks the
// launch's largest batch number against that).  The pass kernels did these as 64-bit divisions - the compiler's float-reciprocal
// sequences, ~10 of them per fetch: a fifth of the vector instructions of a rows pass that is VALU-bound (BDS B1C, DESIGN.md 4.4 xxv).
struct FDiv {
  unsigned mul, d;
};
inline FDiv make_fdiv(long long d) {
  FDiv f;
  f.d = d > 0 ? (unsigned)d : 0u;
  f.mul = d > 1 ? (unsigned)((1ull << 32) / (unsigned long long)d) + 1u : 0u;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FDiv f) { return f.d <= 1u ? x : __umulhi(x, f.mul); }
__device__ __forceinline__ unsigned fmodu(unsigned x, FDiv f) { return x - fdiv(x, f) * f.d; }

enum PreOp { PRE_NONE = 0, PRE_IF_CARRIER, PRE_CODE, PRE_MUL_CONJ };
enum PostOp { POST_STORE = 0, POST_TWIDDLE, POST_ABS_ACC };

struct PassArgs {
  // geometry of this pass
  int len;           // vector length L
  int nvec;          // vectors per transform
  int estride;       // element stride (in complex elements)
  int vstride;       // vector stride
  int cols;          // vectors per workgroup tile
  int nrad;
  int rad[kMaxRadices];
  int n;             // full transform size (twiddle table period)
  int inverse;       // 0: exp(-i..), 1: exp(+i..)
  int pre, post;
  long long in_batch_stride;   // elements between transforms of the batch
  long long out_batch_stride;
  const float2* in;
  float2* out;
  const float2* tw;   // exp(-2*pi*i*k/n), k = 0..n-1
  // PRE_IF_CARRIER
  const int8_t* if_base;
  const float2* if_f32;  // the conditioned signal of gc_acq_condition instead of the int8 record (nullptr: the record)
  long long first_sample;
  int spc, nhops;
  double f0, fstep, fs;  // bin frequency f_b = f0 - fstep*b (Hz)
  // PRE_CODE
  const int8_t* codes;  // [batch][spc]
  // PRE_MUL_CONJ
  const float2* other;  // code spectrum (same layout)
  // POST_ABS_ACC: batch index = bin; loops over nhops transforms in*, accumulates |.|/n
  float* acc_out;
  int acc_add;  // POST_ABS_ACC: add to what acc_out already holds (second code arm of the same PRN)
  float acc_scale;  // POST_ABS_ACC: weight of this arm (B1C: sqrt(11/40), sqrt(29/40)); 0 means 1
  // POST_ABS_ACC with few bins: the hops of a bin are split over hop_groups workgroups (otherwise tiles x bins
  // workgroups, ~2 per CU, each walking all the hops); group g's raw sums go to acc_part[g][bin][n] and
  // abs_combine_kernel adds them in group order
  int hop_groups, acc_bins;
  // Hand-over between the inverse transform's two passes in the CONSUMER's tile order: element (row r, column k) of the [OTHER rows][L]
  // intermediate at (k / B) * (rows * B) + r * B + k % B, B = the columns pass's tile width (a power of two) - its tile is then
  // ONE contiguous run instead of `rows` segments of B values (half a 128-byte line each at B = 8).  out_blocked: log2(B) + 1 on
  // the producing rows pass, in_blocked != 0 on the consuming columns pass; 0: natural order.
  int out_blocked, in_blocked;
  // shifted rows pass of the inverse transform: one workgroup walks row_reps consecutive batches (hops of ONE bin: the shift, the
  // source rows, the code-spectrum values and the twiddle tables are the same for all of them); 0 or 1: one batch per workgroup
  int row_reps;
  int bins_per_wg;   // fused columns pass without hop groups: consecutive batches (bins) one workgroup takes, the next one's inputs fetched ahead (0 / 1: one)
  int nbatch_total;  // ... and how many batches the launch has in all
  int no_xcd_pairs;  // 0: every XCD a contiguous run of the strided passes' tiles; 1 (GC_ACQ_NO_XCD_PAIRS): blockIdx -> tile as it comes; 2 (GC_ACQ_XCD_MAP=pairs): neighbours paired (A/B)
  float* acc_part;
  // PRE_MUL_CONJ with circular spectrum shifts (the circshift search family): batch tb reads input transform
  // tb / shift_bins shifted by tb % shift_bins natural-frequency bins; n1, n2 give the [k1][k2] storage order
  int shift_bins, n1, n2;
  // the same with batch tb = bin * nhops + hop (coarse search whose bin spacing is a whole number of FFT bins): reads input
  // transform `hop` shifted by bin * shift_q.  shift_den > 1: the spacing is shift_q / shift_den bins (Galileo E5b: 60 Hz x 2 ms = 3 / 25,
  // Galileo E1: 150 Hz x 8 ms = 6 / 5) - bin b reads transform (b % shift_den) * nhops + hop, one of shift_den x nhops, shifted by
  // (b / shift_den) * shift_q whole bins
  int shift_q;
  int shift_den;
  // rows pass of a data + pilot search with both arms in ONE launch (gc_acquire_coarse_offsets): the launch's first arm_batches batches
  // are arm 0's, the next ones arm 1's ...; arm k multiplies with other + k * n and writes transform (bin * narms + k) * nhops + hop of
  // the intermediate, so that the columns pass sees narms * nhops hops per bin and adds the arms' magnitudes like hops.  0: one arm
  int arm_batches, narms_merged;
  // the columns pass of such a search when the arms have different weights (BDS B1C: sqrt(11/40), sqrt(29/40)): hop r of a bin belongs to
  // arm r / arm_hops and its magnitude counts arm_w[arm] times.  0: every hop counts once
  int arm_hops;
  // the launch's run-time divisors as multiply-high constants (filled by launch_pass)
  FDiv fd_nhops, fd_shift_bins, fd_sden, fd_hg, fd_arm_batches, fd_arm_hops;
  float arm_w[4];
  int shift0;  // whole bins added to every batch's shift (a search around another centre frequency: gc_acquire_coarse_offsets), in [0, n)
  // PRE_IF_CARRIER on a transform longer than the reference's 2*spc (sizes the radix-{2..8} plan cannot take are padded to
  // the next one it can, launch in gc_acquire_coarse_multi): positions wrap_len .. wrap_len + spc - 1 repeat the first spc
  // (mixed) samples, everything behind is zero
  int wrap_len;
  // POST_ABS_ACC without hop groups, last code arm of a PRN: the workgroup's own peak candidate (PeakTrack::publish_slot) over the
  // first peak_valid columns goes to peak_slots[2 * blockIdx.x] - the finished results are not read back by a peak kernel
  // (94 bins x 144 000 columns = 54 MB per PRN in the Galileo E1 search: 45 us of the 190 us a PRN took)
  unsigned long long* peak_slots;
  int peak_valid;
  // ... and, next to each slot, the workgroup's runner-up value (float bits; PeakTrack::m2) for the float64 guard; nullptr: not wanted
  unsigned int* peak_second;
  // first batch of the launch (fft_pass_ct): gc_acq_shift_row recomputes ONE row of a search whose results were not written
  int batch0;
  // POST_ABS_ACC (fft_pass_ct): batch q's sums land at acc_out + (q - acc_row0) * N - the batch search writes PRN k's winning row
  // (batch irow) to slot k of its row buffer with acc_out = slot k, acc_row0 = irow (no pointer formed outside the allocation)
  int acc_row0;
};

// The running maximum of a thread / workgroup with MATLAB's first-occurrence rule (acquisition.m:196-198), see the peak kernels below
struct PeakTrack {
  unsigned int m = 0, bin = 0xffffffffu, col = 0xffffffffu;
  // the largest value among all OTHER cells seen (== m when another cell holds the same value): how far the runner-up is from the
  // winner decides whether the float32 ordering can be trusted or the cells go to the float64 guard (acq_guard.h)
  unsigned int m2 = 0;
  __device__ __forceinline__ void see(float v, unsigned int b, unsigned int c) {
    const unsigned int u = __float_as_uint(v);
    if (u > m) {
      m2 = m;
      m = u;
      bin = b;
      col = c;
    } else {
      m2 = max(m2, u);
      if (u == m) {
        bin = min(bin, b);
        col = min(col, c);
      }
    }
  }
  // one pair of atomics per workgroup at most, and none when the workgroup's maximum is below what is already there
  // (every wave of a 4 000-workgroup launch hitting the same two addresses cost 0.37 ms per PRN)
  // the workgroup's candidate in thread 0: {maximum's bits, smallest bin, smallest column among the lanes that hold it}
  __device__ __forceinline__ bool reduce(unsigned long long& ka, unsigned long long& kb, unsigned int* second = nullptr) const {
    __shared__ unsigned int sm[16], sb[16], sc[16], s2[16];  // one entry per wavefront (workgroups of up to 1024 threads)
    unsigned int wm = m;
    for (int off = 32; off > 0; off >>= 1) wm = max(wm, (unsigned int)__shfl_xor((int)wm, off, 64));
    unsigned int b = m == wm ? bin : 0xffffffffu, c = m == wm ? col : 0xffffffffu;
    // the wave's runner-up: every lane's second, every lane's maximum except ONE holder of the wave's (two holders: a tie)
    const unsigned long long holders = __ballot(m == wm);
    unsigned int w2 = (m == wm && __popcll(holders) == 1) ? m2 : (m == wm ? m : max(m, m2));
    for (int off = 32; off > 0; off >>= 1) {
      b = min(b, (unsigned int)__shfl_xor((int)b, off, 64));
      c = min(c, (unsigned int)__shfl_xor((int)c, off, 64));
      w2 = max(w2, (unsigned int)__shfl_xor((int)w2, off, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
      sm[wave] = wm;
      sb[wave] = b;
      sc[wave] = c;
      s2[wave] = w2;
    }
    __syncthreads();
    ka = kb = 0;
    if (threadIdx.x != 0) return false;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) {
      if (sm[w] > wm) {
        w2 = max(max(w2, wm), s2[w]);  // the old maximum is now a runner-up
        wm = sm[w];
        b = sb[w];
        c = sc[w];
      } else {
        w2 = max(max(w2, sm[w]), s2[w]);  // (sm[w] == wm: a second holder, w2 becomes wm)
        if (sm[w] == wm) {
          b = min(b, sb[w]);
          c = min(c, sc[w]);
        }
      }
    }
    if (second) *second = w2;
    if (b == 0xffffffffu) return true;  // nothing seen: keys stay 0
    ka = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - b);
    kb = ((unsigned long long)wm << 32) | (unsigned long long)(0xffffffffu - c);
    return true;
  }
  // one pair of atomics per workgroup at most, and none when the workgroup's maximum is below what is already there
  // (every wave of a 4 000-workgroup launch hitting the same two addresses cost 0.37 ms per PRN)
  // second != nullptr: the PRN's runner-up value (float bits) by the same scheme - whichever of {this workgroup's maximum, the key it
  // displaces} loses goes to *second together with the workgroup's own second
  __device__ __forceinline__ void publish(unsigned long long* keys, unsigned int* second = nullptr) const {
    unsigned long long ka, kb;
    unsigned int w2 = 0;
    if (reduce(ka, kb, &w2) && ka) {
      unsigned int loser = (unsigned int)(ka >> 32);
      if (ka > __hip_atomic_load(&keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        const unsigned long long old = atomicMax(&keys[0], ka);
        if (ka > old) loser = (unsigned int)(old >> 32);
      }
      if (kb > __hip_atomic_load(&keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[1], kb);
      if (second) {
        w2 = max(w2, loser);
        if (w2 > __hip_atomic_load(second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(second, w2);
      }
    }
  }
  // the same candidate as a plain store into this workgroup's own slot (keys_reduce_kernel picks them up).  Thread 0 of EVERY
  // workgroup stores unconditionally (zeros when it saw nothing): the slot buffer is cleared only when it is allocated
  // second_slot != nullptr: the workgroup's runner-up value (float bits) next to its candidate (keys_reduce_kernel)
  __device__ __forceinline__ void publish_slot(unsigned long long* slot, unsigned int* second_slot = nullptr) const {
    unsigned long long ka, kb;
    unsigned int w2 = 0;
    if (reduce(ka, kb, &w2)) {
      slot[0] = ka;
      slot[1] = kb;
      if (second_slot) *second_slot = w2;
    }
  }
};

struct AcqScratch {
  int n = 0;
  Plan plan;
  float2* tw = nullptr;       // n
  float2* sig = nullptr;      // nbh * n   signal spectra, layout [k1][k2]
  float2* tmp = nullptr;      // nbh * n   scratch between passes
  float2* codespec = nullptr; // nprn * n
  float* results = nullptr;   // nbins * n: sums of a PRN's earlier code arms (what its last arm adds to).  NOT the finished results of a search:
                              // the last arm of a PRN feeds nothing but its peak keys and is not written back (abs_combine_kernel, fft_pass_ct)
  float* partial = nullptr;   // hop-group sums of the last inverse pass (launch_pass)
  size_t partial_cap = 0;
  // second lane of the PRN loop (gc_acquire_coarse_multi): odd PRNs run on a stream of their own with their own intermediates, so one
  // PRN's columns pass fills the device while the next PRN's rows pass drains (and the other way round)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
  hipStream_t lane_stream[2] = {nullptr, nullptr};  // where the lanes' launches go in the call under way
  float2* tmp2 = nullptr;
  float* results2 = nullptr;
  float* partial2 = nullptr;
  size_t partial2_cap = 0;
  int lane = 0;               // the lane the launches under way belong to (launch_abs_pass picks its partial buffer by it)
  int nlanes = 1;             // lanes of the PRN loop under way: their launches run together, which counts when hop groups are chosen
  // circshift search with the row maxima taken inside the last pass (gc_acq_shift_search): `results` holds nothing then and
  // gc_acq_shift_row transforms the row it is asked for again, with the arms and weights of the search
  bool shift_rows_fused = false;
  int shift_narms = 0;
  double shift_weight[4] = {1.0, 1.0, 1.0, 1.0};
  int8_t* codes = nullptr;    // nprn * spc
  size_t codes_cap = 0;
  long long* sums = nullptr;  // 3 + scratch for argmax
  long long nbh = 0;
  int nprn = 0, nbins = 0;
  // circshift search family (gc_acq_shift_*)
  gc_acq_shift_params shift;  // what `sig` currently holds (n == 0: nothing)
  bool shift_padded = false;  // the block length is no size for the plan: every row has its own carrier, transforms of s->n >= 2*shift.n points
  float* rowmax = nullptr;
  int* rowarg = nullptr;
  float* rowsecond = nullptr;  // per row the largest value of any OTHER cell of the row (== rowmax: a second cell holds the maximum); nullptr: not tracked
  // pinned staging for the circshift family's read-backs (row maxima per search, the winning row's n sums per PRN): a copy into the
  // caller's pageable array goes through the runtime's own staging in pieces - 0.15 - 0.3 ms for the 1.4 MB row of a B1C search
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  int shift_rows = 0;
  // gc_acq_shift_search_batch: every PRN's codes, code spectra, row maxima, the winning rows and the picks of one search
  GcBuf b_codes, b_chips, b_codespec, b_rowmax, b_rowarg, b_rows, b_pick, b_rowsec;
  int shift_slot_lanes = 1;   // lanes of the batch call under way: launch_abs_pass gives each its own region of row-candidate slots
  unsigned long long* peaks = nullptr;  // per-PRN peak keys of gc_acquire_coarse_multi
  int peaks_cap = 0;
  unsigned long long* slots = nullptr;  // per-workgroup peak candidates of abs_combine_kernel, one region per PRN
  size_t slots_cap = 0;
  unsigned int* sec_slots = nullptr;    // per-workgroup runner-up values (float bits), slots_cap / 2 of them (ensure_slots)
  // the float64 guard (acq_guard.h): per-PRN runner-up, cells, their per-hop values, the slow path's candidate list + count
  GcBuf b_second, b_cells, b_exact, b_list, b_off;
  int guard_ties = 0;          // PRNs of the last search whose runner-up was within gc_acq_tie_eps of the winner (resolved in float64)
  double guard_max_dev = 0.0;  // largest |float32 peak - float64 peak| / float64 peak over the last search's PRNs
  int slots_per_prn = 0;                // workgroups per region in the call under way (0: keys were published directly)
};

// The coarse search's two streams, one pair per device for the whole process (created on first use, never destroyed).  HIP deals
// streams out to a few hardware queues; with a stream pair per context, whether a context's two PRN lanes really ran side by side
// depended on how many streams the process had made before: of six engines in one process the second searched in 3.65 instead of
// 2.77 ms (its lanes one after the other), bench.py's searches ran 20 - 30 % slower than the same searches alone, and streams of
// different priority (the multi.hip remedy) moved the bad case elsewhere and made it worse (5.7 ms).  One pair made back to back and
// used by every context behaves the same for all of them.  (Searches of two contexts on one device at the same time share the
// pair: still correct - every call forks and joins with its own events - and no faster than one after the other.)
struct AcqStreams {
  hipStream_t main = nullptr, lane = nullptr;
};

// ---- acq_fft.hip ------------------------------------------------------------------------------------------------------------------
bool make_plan(int n, Plan* pl);
// tile width C1 of the specialised columns pass for vectors of `len`, `nvec` of them per transform; 0: no specialised pass
int ct_columns_tile(int len, int nvec);
int launch_pass(gc_context* ctx, PassArgs& a, long long nbatch_groups, bool* used_ct = nullptr);
int handover_block(const Plan& pl);
void fill_sub(PassArgs& a, const SubPlan& sp);
int choose_cols(int L, int estride = 1);
// Forward transform of `nbatch` sequences produced by `pre` into `dst` (layout [k1][k2])
int forward(gc_context* ctx, AcqScratch* s, PassArgs base, int pre, long long nbatch, float2* dst);
// GC_ACQ_FUSED (tuning build): the whole inverse side of a 36 000- / 24 000-point search in one launch; false: no fused kernel for this plan
struct FusedArgs {
  const float2* tw;        // exp(-2 pi i m / N), m < N
  const float2* sig;       // [nsrc][N] signal spectra, [k1][k2]
  const float2* codespec;  // [nprn * narms][N]
  unsigned long long* keys;  // [nprn][2]
  int nbins, nhops, narms;
  int shift_q;             // > 0: bin b reads hop spectrum h shifted by b * shift_q bins; 0: spectrum b * nhops + h as it is
  int valid;               // columns that count for the peak (2 * spc; the transform may be longer)
  float inv_n;
  float weight[4];         // per code arm (B1C: sqrt(11/40), sqrt(29/40)); 0 means 1
};
bool launch_fused(gc_context* ctx, const Plan& pl, const FusedArgs& a, int nprn);

// ---- acq_coarse.hip ---------------------------------------------------------------------------------------------------------------
AcqStreams* acq_streams(int device);
bool lane_events(AcqScratch* s);
void free_scratch(AcqScratch* s);
int ensure_slots(AcqScratch* s, size_t want);
int ensure_scratch(gc_context* ctx, int n, long long nbh, int nprn, int nbins, int spc, AcqScratch** out);
// up to four device regions back to host arrays through ONE pinned staging buffer and one synchronisation of the context's stream (a copy
// into pageable memory goes through the runtime's own staging in pieces and blocks: 20 - 30 us each)
struct AcqBack {
  void* dst;
  const void* src;
  size_t bytes;
};
int acq_read_back(gc_context* ctx, AcqScratch* s, const AcqBack* parts, int nparts);
int launch_abs_pass(gc_context* ctx, AcqScratch* s, PassArgs& a, long long nbins, unsigned long long* keys = nullptr, int valid = 0,
                    int ip = 0, int nprn = 1, bool* rows_fused = nullptr, int bin0 = 0, long long nbins_total = 0);

}  // namespace gcacq
