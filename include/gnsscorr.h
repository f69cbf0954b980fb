/*
 * gnsscorr.h — C-ABI of libgnsscorr.so, the MI355X (gfx950) replacement for the hot path of
 * gnsscusdr/CU-SDR-Collection (reference @ 2024_10_08, 100 % MATLAB).
 *
 * The reference has no FFI boundary of its own: the boundary is cut *inside*
 * two MATLAB functions and every entry point below names the reference lines it replaces
 * (paths relative to the reference root; "L1CA/" = GPS/GPS_L1CA/).
 *
 *   tracking.m:226-236   fread + int8->double + de-interleave      -> gc_load_if / gc_attach_if
 *   tracking.m:156-158   padded code table [c(L) c c(1)]           -> gc_set_channel / gc_set_code
 *   tracking.m:247-300   replica ramps, carrier, mix, six sums     -> gc_correlate (+ replay API)
 *   tracking.m:133-368   channel x epoch loops incl. loop closure  -> gc_track  (host C++ filters)
 *   acquisition.m:151-200  sigPower, coarse PCPS search, peak pick -> gc_acquire_coarse
 *   acquisition.m:206-254  fine-frequency search                   -> gc_acquire_fine
 *
 * Conventions: plain C types only; status-code returns (0 = OK, negative = error, text via
 * gc_last_error()); no exceptions cross the ABI; caller-owned host buffers are never retained
 * past the call that receives them; a context is used by one host thread at a time.
 * There is NO CPU fallback: every compute entry point runs HIP kernels on the context's
 * device and fails with GC_E_HIP if that is impossible.
 */
#ifndef GNSSCORR_H
#define GNSSCORR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GC_API_VERSION 4 /* 4: gc_acq_shift_pick.peak / second_peak are doubles (float64 guard); gc_device_count, gc_acq_guard_stats */

typedef struct gc_context gc_context;

enum gc_status {
  GC_OK = 0,
  GC_E_INVALID = -1,     /* bad argument */
  GC_E_RANGE = -2,       /* a block reaches past the loaded IF buffer (tracking.m:241-245) */
  GC_E_NOMEM = -3,
  GC_E_HIP = -4,         /* HIP runtime / device failure */
  GC_E_STATE = -5,       /* call sequence error (no IF loaded, channel not configured ...) */
  GC_E_UNSUPPORTED = -6
};

/* settings.dataType ('schar' | 'int16'), initSettings.m:60 */
enum gc_dtype { GC_I8 = 0, GC_I16 = 1 };
/* settings.fileType (1 = real, 2 = I/Q interleaved), initSettings.m:62-65; GC_QI is the
 * GLONASS sample order (GLO_GL1/include/tracking.m:227). */
enum gc_layout { GC_REAL = 0, GC_IQ = 1, GC_QI = 2 };

#define GC_MAX_ARMS 3
#define GC_TAPS 3  /* early, prompt, late */

/* ---- lifetime ---------------------------------------------------------------------- */
int gc_create(gc_context** ctx, int device_id);
int gc_destroy(gc_context* ctx);
const char* gc_last_error(void);
/* What this library was built as.  GC_BUILD_TUNING: libgnsscorr_tuning.so (-DGC_TUNING=1) - the kernels' A/B switches of
 * docs/KNOBS.md are live environment variables and the experimental kernels are linked in; the library that ships reads none. */
enum { GC_BUILD_TUNING = 1 };
int gc_build_flags(void);
int gc_api_version(void);
/* Device name / CU count of the context's device (for bench reporting). */
int gc_device_info(gc_context* ctx, char* name, int name_len, int* compute_units);
/* How many HIP devices this process sees (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES applied): the valid device_id of gc_create are
 * 0 .. *count - 1 (gc_create refuses an ordinal that is not gfx950).  0 without a driver or a device; needs no context.  The caller that shards channels over the GPUs of a node (tracking.m:133: channels are independent) makes
 * ONE context per device - gc_create(&ctx[i], i), i < *count - instead of SURVEY 8b's gc_create(ctx**, device_ids, ndev): a context
 * is bound to one device, one stream and one host thread ("deviations from SURVEY 8b"). */
int gc_device_count(int* count);
int gc_synchronize(gc_context* ctx);

/* ---- IF samples -> HBM (replaces fread + conversion, tracking.m:226-236) ------------ */
/* Copies `nsamples` samples (complex samples for IQ/QI, real samples for REAL) of raw
 * integers to the device.  The raw bytes stay as they are in HBM; conversion happens in
 * the kernels.  Replaces any previously loaded buffer. */
int gc_load_if(gc_context* ctx, const void* samples, uint64_t nsamples, int dtype, int layout);
/* 2-bit sign-magnitude packed complex records (two samples per byte; the reference converts them to schar files on
 * the CPU with GPS_L5C/include/unpack_cplx.m:32-49): upload the packed bytes and expand them on the GPU into the
 * int8 I/Q record (2 * nbytes samples).  SURVEY.md §8f item 2. */
int gc_load_if_packed2(gc_context* ctx, const void* packed, uint64_t nbytes);
/* As gc_load_if, reading from a file: `skip_bytes` as postProcessing.m:74
 * (fseek(fid, dataAdaptCoeff*skipNumberOfBytes)); nsamples = 0 reads to end of file. */
int gc_open_if_file(gc_context* ctx, const char* path, uint64_t skip_bytes, uint64_t nsamples,
                    int dtype, int layout);
/* Zero-copy: adopt a device pointer that already holds the raw samples (not owned). */
int gc_attach_if(gc_context* ctx, void* device_ptr, uint64_t nsamples, int dtype, int layout);
/* Device pointer and size of the current IF buffer (for tools that fill it in place). */
int gc_if_buffer(gc_context* ctx, void** device_ptr, uint64_t* nsamples);
/* sample type and order of the loaded record (GC_I8 / GC_I16, GC_IQ / GC_QI / GC_REAL) */
int gc_if_format(gc_context* ctx, int* dtype, int* layout);
/* Allocate an uninitialised device IF buffer owned by the context. */
int gc_alloc_if(gc_context* ctx, uint64_t nsamples, int dtype, int layout);
/* Read back raw samples [first, first+n) to the host (tests, CPU-baseline sampling). */
int gc_read_if(gc_context* ctx, uint64_t first, uint64_t n, void* dst);

/* ---- code replicas (replaces tracking.m:156-158) ------------------------------------ */
/* Declares a tracking channel: `arms` code arms (1 = data only; 2 = data+pilot,
 * GPS_L5C/include/tracking.m:151-160; 3 = B1C wide-band, BDS/B1C/include/WB_tracking.m:176-188)
 * whose index ramps are multiplied by `index_scale` R (1; 2 for BOC(1,1) / RZ tables,
 * GAL_E1C/include/tracking.m:236-262). */
int gc_set_channel(gc_context* ctx, int channel, int arms, double index_scale);
/* Padded table exactly as the reference builds it ([c(end) c c(1)], values in {-1,0,+1});
 * `arm_mult` is the extra factor applied to the ramp of this arm before ceil()
 * (6 for the BOC(6,1) arm, WB_tracking.m:293; otherwise 1). */
int gc_set_code(gc_context* ctx, int channel, int arm, const int8_t* table, int n_entries,
                double arm_mult);
/* Optional: stage only `window_entries` table entries (starting at the block's table_offset)
 * into LDS — needed for the 1 534 502-entry GPS L2C CL table, of which one block touches
 * 2*codeLength+2 entries (GPS_L2C/include/tracking.m:261).  0 = whole table (default). */
int gc_set_code_window(gc_context* ctx, int channel, int arm, int window_entries);

/* ---- correlator (replaces tracking.m:247-300) --------------------------------------- */
/* One integrate-and-dump block = the five scalars the reference records per epoch
 * (tracking.m:212-216,249,277,314,332) + block geometry. */
typedef struct gc_block {
  int32_t channel;          /* index given to gc_set_channel */
  int32_t blksize;          /* N, tracking.m:222 */
  int64_t first_sample;     /* 0-based sample index of the block start in the IF buffer
                               (ftell/dataAdaptCoeff, tracking.m:212-216) */
  double rem_code_phase;    /* remCodePhase  (chips),   tracking.m:249 */
  double code_phase_step;   /* codeFreq/fs,             tracking.m:219 */
  double el_spacing;        /* earlyLateSpc (chips),    tracking.m:94  */
  double carr_freq;         /* carrFreq (Hz),           tracking.m:314 */
  double rem_carr_phase;    /* remCarrPhase (rad),      tracking.m:277 */
  int32_t table_offset[GC_MAX_ARMS]; /* per-arm index offset (GPS_L2C tracking.m:261) */
  int32_t reserved;
} gc_block;

/* Test hook: non-zero forces the generic correlator kernel (per-sample table lookup) even where the
 * fast single-transition kernel applies, so both can be checked against the oracle. */
int gc_force_generic_kernel(gc_context* ctx, int on);

/* Sampling frequency used for the carrier replica (settings.samplingFreq, tracking.m:280). */
int gc_set_sampling_freq(gc_context* ctx, double fs);

/* Synchronous correlate: `out` receives arms x [I_E,Q_E,I_P,Q_P,I_L,Q_L] doubles per block,
 * blocks in input order, stride = 6*GC_MAX_ARMS doubles per block.
 * Returns GC_E_RANGE (nothing computed) if any block exceeds the IF buffer. */
#define GC_OUT_STRIDE (6 * GC_MAX_ARMS)
int gc_correlate(gc_context* ctx, int nblocks, const gc_block* blocks, double* out);

/* Replay (batched, open-loop) mode: descriptors stay resident in HBM so that the timed
 * region contains only kernel work (SURVEY.md §7 hard part 1a). */
int gc_replay_prepare(gc_context* ctx, int64_t nblocks, const gc_block* blocks);
int gc_replay_launch(gc_context* ctx);                 /* asynchronous on the context stream */
int gc_replay_fetch(gc_context* ctx, double* out);     /* waits, copies nblocks*GC_OUT_STRIDE */
/* hipEvent timer on the context stream (bench: roofline.achieved). */
int gc_timer_start(gc_context* ctx);
int gc_timer_stop(gc_context* ctx, double* elapsed_ms);

/* ---- closed-loop tracking (replaces tracking.m:133-368; loop filters on the host) ---- */
enum gc_pll_kind {
  GC_PLL_2ND_ORDER = 0, /* L1CA: tracking.m:308-311 with calcLoopCoef.m:41-45 */
  GC_PLL_3_STATE = 1    /* all other packages: GPS_L5C/include/tracking.m:351-353 */
};

typedef struct gc_track_params {
  double sampling_freq;      /* settings.samplingFreq */
  double code_freq_basis;    /* settings.codeFreqBasis */
  double code_length;        /* settings.codeLength (chips per block) */
  double el_spacing;         /* settings.dllCorrelatorSpacing */
  double int_time;           /* settings.intTime (PDIcode = PDIcarr) */
  double dll_noise_bw, dll_damping;   /* settings.dllNoiseBandwidth, dllDampingRatio */
  double pll_noise_bw, pll_damping;   /* settings.pllNoiseBandwidth, pllDampingRatio */
  int32_t pll_kind;          /* gc_pll_kind */
  int32_t pilot_combine;     /* 0 = data arm only; 1 = combine data+pilot discriminators with the
                                pilot rotated by -pi/2 (GPS_L5C tracking.m:336-348);
                                2 = pilot as is (GAL_E1C tracking.m:303-311,326-331);
                                3 = pilot in quadrature, atan(-I/Q) (BDS/B1C NB_tracking.m:340-342);
                                4 = three arms {data, pilot BOC(1,1), pilot BOC(6,1)} folded into one pilot
                                    p = -sqrt(4/33)*p61 + sqrt(29/33)*(Q11, -I11) (WB_tracking.m:364-369),
                                    which is also what the Pilot_* records then hold (:420-425);
                                5 = the same three arms folded IN PHASE for Galileo E1-C CBOC(6,1,1/11):
                                    p = sqrt(10/11)*p11 - sqrt(1/11)*p61, then as 2 (BASELINE config 3; an extension:
                                    the reference's GAL_E1C package uses the BOC(1,1) replica only) */
  double pf1, pf2, pf3;      /* 3-state filter coefficients (calcLoopCoefCarr.m), if used */
  int64_t skip_samples;      /* settings.skipNumberOfBytes, in samples */
  int32_t n_epochs;          /* codePeriods = settings.msToProcess (tracking.m:90) */
  int32_t table_phase_count; /* API version 2.  0 = off.  GPS L2C: 75 — the pilot arm's table is read through a window
                                that advances by code_length entries per epoch, table_offset[1] = code_length *
                                (phase - 1), phase = channel.CLCodePhase, +1 per epoch, back to 1 after
                                table_phase_count (GPS_L2C/include/tracking.m:261,357-360) */
  /* API version 2: discriminator weights {data, pilot}; all-zero pairs mean 1:1 (the plain averages of L5 / E1).
   * B1C NB: pll 11:29, dll 11:29 (NB_tracking.m:342,349); B1C WB: pll 1:3, dll factor:(1-factor) with
   * factor = CalcWeighingFactor(settings) (WB_tracking.m:382,403). */
  double pll_weight[2];
  double dll_weight[2];
  double dll_scale;          /* both DLL discriminators are multiplied by it: 1 - earlyLateSpc for B1C
                                (NB_tracking.m:346-348); 0 means 1 */
  /* API version 3: C/N0 by the variance-summing method inside the loop (tracking.m:351-358, Common/CNoVSM.m:38-47): after
   * every cno_interval epochs (settings.CNo.VSMinterval; 0 = off) CNoVSM(I_P, Q_P, cno_acc_time) of those epochs' data-arm
   * prompt sums goes to the buffer registered with gc_set_cno_output.  It feeds nothing back into the loop. */
  int32_t cno_interval;
  int32_t cno_mode;          /* gc_cno_mode: which estimator runs every cno_interval epochs */
  double cno_acc_time;       /* settings.CNo.accTime (GC_CNO_VSM) / settings.intTime (GC_CNO_PLD*) */
} gc_track_params;

/* GC_CNO_VSM: Common/CNoVSM.m, one value per interval (trackResults.CNo.VSMValue).
 * GC_CNO_PLD*: BDS/B2a + BDS/B1C Calc_CNo_PLD.m with the bookkeeping of B2a tracking.m:409-432 (cno_interval =
 * settings.CNoInterval): GC_CNO_NPLD values per interval - DataCNo, PilotCNo, B2a_CNo / B1C_CNo (each the 0.5/0.5 average of
 * this interval's estimate with the previous one, zeros before the first) and DataPLD, PilotPLD (the narrow-band lock
 * detectors NBD/NBP with the data bits wiped by sign).  _PILOT_SWAPPED reads the pilot prompt pair as (Q, I)
 * (pilotTRKflag == 1, Calc_CNo_PLD.m:72-75), _PILOT as (I, Q) (pilotTRKflag == 2 of BDS/B1C), plain GC_CNO_PLD has no pilot
 * arm (PilotCNo = PilotPLD = 0, the third value = DataCNo's estimate).  The library evaluates these from the epoch records
 * when the tracking call returns, in every loop mode; they feed nothing back. */
enum gc_cno_mode { GC_CNO_VSM = 0, GC_CNO_PLD = 1, GC_CNO_PLD_PILOT_SWAPPED = 2, GC_CNO_PLD_PILOT = 3 };
#define GC_CNO_NPLD 5

typedef struct gc_channel_init {
  int32_t channel;           /* gc_set_channel index holding this PRN's tables */
  int32_t prn;               /* recorded only */
  double acquired_freq;      /* channel.acquiredFreq, preRun.m:68 */
  double code_freq;          /* initial codeFreq: settings.codeFreqBasis (tracking.m:163) or
                                channel.codeFreq (GPS_L5C tracking.m:165) */
  int64_t code_phase;        /* channel.codePhase, 1-based sample index, preRun.m:69 */
  int32_t table_phase;       /* API version 2: channel.CLCodePhase (1-based; GPS_L2C preRun.m:71), else 0 */
  int32_t reserved;
} gc_channel_init;

/* Per-epoch records, one row of n_epochs doubles per field per channel
 * (trackResults fields, tracking.m:47-86).  Layout: out[(ch*GC_TRK_NFIELDS + field)*n_epochs + e]. */
enum gc_track_field {
  GC_TRK_ABSOLUTE_SAMPLE = 0, GC_TRK_CODE_FREQ, GC_TRK_CARR_FREQ,
  GC_TRK_I_E, GC_TRK_Q_E, GC_TRK_I_P, GC_TRK_Q_P, GC_TRK_I_L, GC_TRK_Q_L,
  GC_TRK_DLL_DISCR, GC_TRK_DLL_DISCR_FILT, GC_TRK_PLL_DISCR, GC_TRK_PLL_DISCR_FILT,
  GC_TRK_REM_CODE_PHASE, GC_TRK_REM_CARR_PHASE,
  GC_TRK_PILOT_I_E, GC_TRK_PILOT_Q_E, GC_TRK_PILOT_I_P, GC_TRK_PILOT_Q_P,
  GC_TRK_PILOT_I_L, GC_TRK_PILOT_Q_L,
  GC_TRK_NFIELDS
};

/* Runs the reference's tracking loop for `nch` channels in lock step: one correlator launch
 * per epoch for all channels, discriminators + loop filters on the host between launches.
 * `epochs_done[ch]` receives the number of completed epochs (== n_epochs unless the IF
 * buffer ran out, in which case the call returns GC_E_RANGE after filling what it could —
 * the reference's early return, tracking.m:241-245). */
int gc_track(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
             double* out, int32_t* epochs_done);
/* Where the next tracking calls of this context put trackResults.CNo.VSMValue: cno[ch * (n_epochs / cno_interval) + k] for
 * the k-th completed interval of channel slot ch (caller-owned, `capacity` doubles; NULL unregisters); with a GC_CNO_PLD*
 * mode cno[(ch * (n_epochs / cno_interval) + k) * GC_CNO_NPLD + j].  With the device loop
 * (gc_track_device) the estimator runs in the kernel that closes the loop. */
int gc_set_cno_output(gc_context* ctx, double* cno, int64_t capacity);

/* ---- records larger than the device: tracking window by window ------------------------------------------------------
 * tracking.m reads its file block by block (fread at :226-245) and so handles a record of any length
 * (postProcessing.m:61-96 never loads it whole).  gc_track works on the resident IF buffer; these two entry points carry a
 * channel's loop state from one window of the record to the next. */
typedef struct gc_channel_state {   /* what tracking.m keeps in its local variables between two blocks of a channel */
  int64_t next_sample;       /* first sample of the next block, counted from the start of the RECORD (ftell/dataAdaptCoeff) */
  double code_freq, rem_code_phase;        /* codeFreq, remCodePhase (:219,273) */
  double carr_freq, rem_carr_phase;        /* carrFreq, remCarrPhase (:283,317) */
  double old_code_nco, old_code_error;     /* oldCodeNco, oldCodeError (:326-330) */
  double old_carr_nco, old_carr_error;     /* oldCarrNco, oldCarrError (:308-312) */
  double d_carr_error, d2_carr_error;      /* third-order PLL integrators (GPS_L5C tracking.m:351-353) */
  int32_t table_phase;       /* CLCodePhase (GPS_L2C) */
  int32_t status;            /* 0 running, 2 the record ended inside this channel's next block (tracking.m:241-245) */
  int64_t reserved;
} gc_channel_state;
#define GC_TRACK_RESUME 1        /* `state` holds the end state of the previous window (else it is only written) */
#define GC_TRACK_PAUSE_AT_END 2  /* more of the record follows: stop all channels, in lock step, at the first epoch one of
                                    them cannot read from this window (*paused = 1) instead of ending that channel */
/* gc_track on the window currently in the IF buffer, whose first sample is sample `origin` of the record.  Runs at most
 * p->n_epochs epochs from the given state; `out` / `epochs_done` count from this call's first epoch; absoluteSample is
 * recorded in record coordinates. */
int gc_track_resume(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
                    gc_channel_state* state, int flags, int64_t origin, double* out, int32_t* epochs_done,
                    int32_t* paused);
/* tracking(fid, channel, settings) on a file of any size: the record is read in windows of `window_samples` samples into two
 * device buffers (the next window's read + upload runs while the current one is tracked: at most 2 * window_samples
 * samples are resident), results as gc_track's.  `skip_bytes`: file offset of record sample 0.  Host-closed loop. */
int gc_track_file(gc_context* ctx, const char* path, uint64_t skip_bytes, int dtype, int layout, uint64_t window_samples,
                  const gc_track_params* p, int nch, const gc_channel_init* init, double* out, int32_t* epochs_done);

/* As gc_track, with the loop closed ON THE DEVICE: one persistent cooperative launch runs every
 * epoch; the workgroups sharing a channel exchange self-validating 16-byte {payload, epoch tag} messages (no atomics,
 * no fences) and execute tracking.m:302-335 in float64 themselves (csrc/devloop.h).  Covered: int8 I/Q or Q/I records;
 * single-arm R = 1 channels on the transition-mask kernel (GPS L1 C/A, GLONASS L1OF, BDS B1I) and one- or two-arm
 * channels of any rate and index scale with pilot_combine 0-3 on the lane kernel (GPS L5, BDS B2a / B3I, Galileo E5a /
 * E5b / E1 B+C, BDS B1C narrow-band), and the three-arm Galileo E1-C CBOC fold (pilot_combine 5, third arm derived).
 * Other three-arm or mixed-multiplier channels (pilot_combine 4), windowed tables (GPS L2C CL), int16 and real records
 * return GC_E_UNSUPPORTED: use gc_track. */
int gc_track_device(gc_context* ctx, const gc_track_params* p, int nch, const gc_channel_init* init,
                    double* out, int32_t* epochs_done);

/* ---- several tracking() calls at once (BASELINE config 5: the all-constellation mix) ---------------------------
 * The reference tracks the channels of ONE package per call (tracking.m:133 loops over settings.numberOfChannels with one
 * `settings`); its packages run one after the other.  Channels are independent, so the loops of different packages can
 * run side by side on one GPU - GPS L1 C/A, Galileo E1 and BDS B1C channels on the same L1-band record, L5 / E5a / B2a
 * channels on another - or on several GPUs driven by one host process.  One context per (record, package): own stream,
 * code tables and result buffers; a record loaded into one context is shared with the others by gc_share_if (no copy). */
int gc_share_if(gc_context* dst, gc_context* src);   /* dst reads src's IF record (same device; src must outlive the use) */

typedef struct gc_track_job {
  gc_context* ctx;                /* one job per context */
  const gc_track_params* params;  /* this package's settings */
  const gc_channel_init* init;    /* nch channels, as gc_track */
  double* out;                    /* nch * GC_TRK_NFIELDS * n_epochs doubles, as gc_track */
  int32_t* epochs_done;           /* nch */
  int32_t nch;
  int32_t device_loop;            /* 0: gc_track; 1: gc_track_device, falling back to gc_track where it returns GC_E_UNSUPPORTED */
  int32_t status;                 /* out: the job's return code (GC_E_RANGE = short read, partial records, tracking.m:241-245) */
  int32_t reserved;
  char error[240];                /* out: the job's error text */
} gc_track_job;

/* Runs every job's tracking loop concurrently (one host thread per job for the loop closure, tracking.m:302-335) and
 * returns when all are done: GC_OK, or the first failure (a short read only if nothing worse happened).
 * The jobs' persistent kernels are admitted by a per-device ledger that knows THIS process' kernels only: other processes that
 * run persistent kernels on the same GPU at the same time (several ranks pinned to one device) are invisible to it - serialise
 * their tracking calls (bench.py does, with a file lock) or give every process its own GPU. */
int gc_track_multi(int njobs, gc_track_job* jobs);

/* ---- acquisition (replaces acquisition.m:151-254, resampling off) ------------------- */
typedef struct gc_acq_params {
  double sampling_freq;      /* settings.samplingFreq */
  double code_freq_basis;    /* settings.codeFreqBasis */
  double code_length;        /* settings.codeLength */
  double intermediate_freq;  /* settings.IF */
  double search_band;        /* settings.acqSearchBand (Hz, single-sided) */
  double search_step;        /* settings.acqSearchStep */
  int32_t non_coh_time;      /* settings.acqNonCohTime */
  int32_t source;            /* 0: the IF record (int8 I/Q); GC_ACQ_SOURCE_CONDITIONED: the signal gc_acq_condition left */
  int64_t first_sample;      /* start of longSignal within the IF buffer (or within the conditioned signal) */
  /* API version 3, all optional (0 = the GPS L1 C/A scheme above).  They let the search family whose bins are circular shifts
   * of ONE spectrum (BDS/B1C/include/acquisition.m:137-200) run as a carrier-per-bin search when its transform length cannot be
   * taken by the radix plan - after the input conditioning, where samplesPerCode follows ceil(newFs): circshift(X, k) is the
   * carrier moved by k*fs/N, and the N-point circular correlation with a replica of code_samples samples is the linear one of
   * the block followed by a repeat of its first code_samples samples. */
  int32_t block_len;         /* samples per search block (len10PlusXms, B1C :113); 0: 2*samplesPerCode (acquisition.m:174) */
  int32_t code_samples;      /* samples of each sampled-code row, zeros beyond (samplesXmsLen, B1C :111); 0: samplesPerCode.
                                sigPower is taken over that many samples (B1C :138).  Needs non_coh_time == 1 when set. */
  int32_t n_bins;            /* number of frequency bins; 0: round(2*search_band/search_step) + 1 (acquisition.m:124) */
  int32_t reserved;
  double arm_weight[4];      /* results = sum_arm arm_weight * |ifft(...)| (B1C :186-187: sqrt(11/40), sqrt(29/40)); all 0: 1 */
} gc_acq_params;
#define GC_ACQ_SOURCE_CONDITIONED 1

typedef struct gc_acq_result {
  int32_t coarse_bin;        /* 1-based frequency-bin index, acquisition.m:196 */
  int32_t code_phase;        /* 1-based sample index over 2*samplesPerCode columns, :198 */
  double peak;               /* max(max(results)) */
  double peak_metric;        /* peak/sigPower/acqNonCohTime, :200 */
  double coarse_freq;        /* coarseFreqBin(acqCoarseBin), :169-170 */
} gc_acq_result;

/* ---- acquisition, optional input conditioning (acquisition.m:46-111; the same block with its own
 * bandwidth in ten packages; settings.resamplingflag, off by default): zero-phase band-pass around IF -
 * longSignal = filtfilt(fir1(order, [IF - BW/2, IF + BW/2]*2/fs), 1, longSignal) with filtfilt's 3*order odd-reflected edge
 * samples - then band-pass-sampling decimation by index selection, longSignal(ceil((0:len-1)/newFs*fs)) with the first index
 * forced to 1, newFs = ceil of the centre of the admissible range, and IF -> rem(IF, newFs).  The conditioned complex float
 * signal stays on the device; searches read it with gc_acq_params.source = GC_ACQ_SOURCE_CONDITIONED and the returned
 * sampling frequency / IF (the caller maps code phase and carrier frequency back as acquisition.m:264-276 does). */
typedef struct gc_acq_front_params {
  double sampling_freq;      /* settings.samplingFreq */
  double intermediate_freq;  /* settings.IF */
  double bandwidth;          /* BW (acquisition.m:58: codeFreqBasis*2 + 0.5e6; per-package constants elsewhere) */
  int64_t first_sample;      /* start of longSignal within the IF record */
  int64_t n_samples;         /* length(longSignal) */
  int32_t fir_order;         /* 700 */
  int32_t reserved;
  double band_margin;        /* widening of both normalised band edges: wp = [w1*2/fs - m, w2*2/fs + m]; 0.002 in GPS_L5C / BDS_B2a /
                              * BDS_B1C (GPS_L5C/include/acquisition.m:69), 0 in the other packages (GPS_L1CA :62) */
} gc_acq_front_params;
typedef struct gc_acq_front_result {
  double sampling_freq;      /* settings.samplingFreq after the block (:81) */
  double intermediate_freq;  /* settings.IF after the block (:95) */
  int64_t n_samples;         /* signalLen (:84) */
} gc_acq_front_result;
int gc_acq_condition(gc_context* ctx, const gc_acq_front_params* p, gc_acq_front_result* out);
/* The same device-side signal filled WITHOUT the conditioning block, for what the searches do not read directly (they read int8
 * I/Q records): n samples of the loaded record from first_sample in whatever format it has - int16 files (postProcessing.m:61-96,
 * settings.dataType), Q/I order, real samples - or the caller's own complex samples (re, im interleaved floats;
 * acquisition(longSignal, settings) accepts any complex row).  Then source = GC_ACQ_SOURCE_CONDITIONED and first_sample counts
 * from the signal's start.  gc_acq_condition itself takes the record in any format. */
int gc_acq_signal_from_record(gc_context* ctx, int64_t first_sample, int64_t n);
int gc_acq_set_signal(gc_context* ctx, const float* iq, int64_t n);
/* test hook: the conditioned signal back as n complex floats (re, im interleaved) */
int gc_acq_conditioned(gc_context* ctx, int64_t first, int64_t n, float* dst);

/* `sampled_codes`: nprn rows of samplesPerCode int8 (makeCaTable.m:59-67 output).
 * For each PRN returns the coarse peak (acquisition.m:158-200). */
int gc_acquire_coarse(gc_context* ctx, const gc_acq_params* p, int nprn,
                      const int8_t* sampled_codes, gc_acq_result* out);

/* Data + pilot search: `narms` sampled codes per PRN (row prn*narms + arm); the per-arm correlation
 * magnitudes are added before the non-coherent sum (GPS_L5C/include/acquisition.m:175-216).
 * The PRNs are searched on two lanes; both run on a pair of streams the library keeps per DEVICE for the whole process (forked from and
 * joined into the context's stream inside the call - the call is synchronous as before).  Searches of two contexts on one device at
 * the same time therefore share that pair: correct, and no faster than one after the other. */
int gc_acquire_coarse_multi(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                            const int8_t* sampled_codes, gc_acq_result* out);

/* The same with a centre frequency per row: row ip is searched around p->intermediate_freq + freq_offset[ip] (Hz).  One code on
 * several carriers is GLONASS' FDMA search - GLO/GLO_GL1/include/acquisition.m:146-147 runs the L1 C/A scheme once per frequency
 * number K around IF - freqSpacing*K with the common 511-chip code - here as ONE call with the code repeated per row: the rows share
 * the signal spectra (an offset is a whole-bin shift of them) and run on the two PRN lanes.  Every offset must be a whole number of
 * the search's FFT bins, sampling_freq / N, and the bin spacing a whole (or q / den) number of bins - GC_E_UNSUPPORTED otherwise (the
 * caller then searches row by row).  freq_offset == NULL: gc_acquire_coarse_multi.  out[ip].coarse_freq includes the row's offset. */
int gc_acquire_coarse_offsets(gc_context* ctx, const gc_acq_params* p, int nprn, int narms,
                              const int8_t* sampled_codes, const double* freq_offset, gc_acq_result* out);

/* Fine-frequency stage of GPS L1 C/A (acquisition.m:213-254) for one detected PRN:
 * `code` = 1023 chips (+-1), `code_phase` / `coarse_freq` from gc_acquire_coarse.
 * Returns the fine carrier frequency (with the "0 -> 1 Hz" rule of :258-260 applied). */
int gc_acquire_fine_l1ca(gc_context* ctx, const gc_acq_params* p, const int8_t* code,
                         int code_phase, double coarse_freq, double* carr_freq);
/* The same for `ndet` detections in one launch and one read-back (the loop over the detected PRNs of
 * acquisition.m:206-260): codes = ndet rows of code_length chips. */
int gc_acquire_fine_l1ca_batch(gc_context* ctx, const gc_acq_params* p, int ndet, const int8_t* codes,
                               const int32_t* code_phase, const double* coarse_freq, double* carr_freq);

/* ---- acquisition, generic fine-frequency stage (acquisition.m:206-260 and its per-package variants: GPS_L5C
 * :228-252 Neuman-Hofman search, GAL_E5a 100-code secondary search, BDS/B2a non-coherent data+pilot, ...).
 * out[(bin*ncodes + c)*2 + {0,1}] = sum over code period c of x[n] * code[floor(ts*(n + index_offset)/tc) mod
 * code_len] * exp(-1i*2*pi*f_bin*n/fs), n counted from first_sample, f_bin = f0 - bin*fstep.  The hypothesis
 * search over the per-code sums stays with the caller. */
typedef struct gc_fine_params {
  double sampling_freq;      /* settings.samplingFreq */
  double code_freq;          /* settings.codeFreqBasis (tc = 1/code_freq); 0: `code` is a replica already sampled at the
                                sampling rate, one entry per sample - index (n + index_offset) mod code_len (GLONASS' sampled
                                40-code replica, GLO_GL1 acquisition.m:159-164; BDS B1C's table, acquisition.m:236-250; the CL
                                windows of GPS_L2C acquisition.m:140-163) */
  double f0, fstep;          /* FineFreqBins(k) = f0 - fstep*(k-1) */
  int64_t first_sample;      /* absolute index of longSignal(codePhase) */
  int32_t spc;               /* samplesPerCode */
  int32_t ncodes;            /* code periods (40 L1CA, 20 L5/B2a, 100 E5a) */
  int32_t nbins;
  int32_t code_len;          /* settings.codeLength */
  int32_t index_offset;      /* 0: codeValueIndex over (0:K*spc-1) (L1CA :210); 1: over (1:K*spc) (L5 :231) */
  int32_t source;            /* as gc_acq_params.source */
  double dc_re, dc_im;       /* subtracted from every sample first: sig - mean(sig), GPS_L2C acquisition.m:144 (0: nothing) */
} gc_fine_params;

/* mean(x) and var(x) (MATLAB's: normalised by n - 1, of a complex vector) of the n samples from first_sample of the IF record
 * or the conditioned signal (`source` as in gc_acq_params): sigPower = sqrt(var * n) of BDS/B1C acquisition.m:138, the mean
 * of GPS_L2C acquisition.m:144.  Exact integer sums for the int8 record. */
int gc_acq_signal_stats(gc_context* ctx, int64_t first_sample, int64_t n, int32_t source, double* mean_re, double* mean_im,
                        double* var);

int gc_acquire_fine_sums(gc_context* ctx, const gc_fine_params* p, const int8_t* code, double* out);
/* `ndet` detections per call: detection d uses codes + d*code_len, first_sample[d] and f0[d] instead of the fields of
 * `p`; out[((d*nbins + bin)*ncodes + c)*2 + {0,1}]. */
int gc_acquire_fine_sums_batch(gc_context* ctx, const gc_fine_params* p, int ndet, const int8_t* codes,
                               const int64_t* first_sample, const double* f0, double* out);

/* ---- acquisition, circshift search family (replaces GPS_L2C/include/acquisition.m:40-75,
 * BDS/B1I/include/acquisition.m:76-123, BDS/B1C/include/acquisition.m:137-170): the signal block(s) are mixed with
 * n_carriers carriers and transformed once (gc_acq_shift_prepare); per PRN every Doppler bin is a circular shift of
 * that spectrum by 0 .. n_bins-1 positions before the product with the code spectrum and the inverse transform
 * (gc_acq_shift_search).  Rows are ordered ((carrier * n_signals + signal) * n_bins + bin). */
typedef struct gc_acq_shift_params {
  double sampling_freq;      /* settings.samplingFreq */
  double carrier_f0;         /* first wipe-off carrier in Hz: initFreq */
  double carrier_step;       /* carrier i = carrier_f0 + i*carrier_step (B1I: +freqResolution/Nshifts, L2C: minus) */
  int64_t first_sample;      /* signal block k starts at first_sample + k*n */
  int32_t n;                 /* samplesPerBlock = the reference's transform length; any value: lengths other than 2^a 3^b 5^c run one
                                carrier per row on a longer transform, same rows and results */
  int32_t n_signals;         /* consecutive signal blocks (B1I: signal1, signal2) */
  int32_t n_carriers;        /* Nshifts */
  int32_t n_bins;            /* numberOfFrqBins: circshift(IQfreqDom, bin), bin = 0 .. n_bins-1 */
  int32_t n_arms_max;        /* code components per PRN that gc_acq_shift_search will be given (1..4) */
  int32_t source;            /* as gc_acq_params.source */
} gc_acq_shift_params;

int gc_acq_shift_prepare(gc_context* ctx, const gc_acq_shift_params* p);
/* codes: int8 [narms][n], the local replica already sampled and zero-padded to n by the caller
 * ([cmCodesTable(1:spc) zeros], acquisition.m:44-45).  results(row, :) = sum_arm weight_arm * abs(ifft(shifted
 * spectrum .* conj(fft(code_arm)))) (weights: NULL = 1; B1C sqrt(11/40), sqrt(29/40), acquisition.m:186-187).
 * Returns each row's maximum and its 0-based first position; the rows stay on the device for gc_acq_shift_row. */
int gc_acq_shift_search(gc_context* ctx, int narms, const int8_t* codes, const double* arm_weight,
                        float* row_max, int32_t* row_argmax);
int gc_acq_shift_row(gc_context* ctx, int row, float* out /* n floats */);  /* a row of the last gc_acq_shift_search (GC_E_STATE after a batch call) */
/* A package's whole search in ONE call (replaces the PRN loops BDS/B1I/include/acquisition.m:76-176, GPS/GPS_L2C/include/
 * acquisition.m:40-118, BDS/B1C/include/acquisition.m:170-235 up to the threshold test): codes int8 [nprn][narms][n] go up once, every
 * PRN's transforms are queued back to back, the row maxima of all PRNs come back in one copy, the package's selection rule picks each
 * PRN's row, and - rules with a second peak - the winning rows are transformed again and reduced on the device to
 * {first maximum, second peak}: two read-backs per search instead of two per PRN, no row of n floats crosses the bus.
 *   GC_SHIFT_PICK_GLOBAL            row of the largest row maximum, first column holding it (B1C :193-197); no second peak
 *   GC_SHIFT_PICK_SEQUENTIAL        the sequential rule over (carrier, bin) of GPS_L2C :46-66 (n_signals == 1)
 *   GC_SHIFT_PICK_SEQUENTIAL_PAIRS  the same over the larger of the two signal blocks' maxima, BDS/B1I :87-122 (n_signals == 2)
 * codes: sample_index == NULL: int8 [nprn][narms][n], the local replicas sampled and zero-padded by the caller as for
 * gc_acq_shift_search (code_len, n_index ignored).  Otherwise int8 [nprn][narms][code_len] chip tables and ONE 0-based index vector
 * for all of them: replica(k) = chips(sample_index[k]) for k < n_index, 0 for n_index <= k < n - the make*Table.m gather
 * (ceil(ts*k/tc) depends on the rates only: makeCaTableDMA.m, makeCMTable.m, makeDataTable.m) and the [table zeros] padding done on
 * the device: a few KB per code cross the bus instead of n bytes.
 * second_peak: the largest value among the row's first `period` samples at least `exclude` samples away from the peak, with the
 * reference's three range cases (B1I :141-156, L2C :77-91).  Block lengths without specialised pass kernels (16.368-Msps front ends:
 * padded transforms) run PRN after PRN inside the call, every PRN's rows picked - and guarded in float64, gc_acq_guard_stats - from the
 * written sums (API version 4; version 3 answered GC_E_UNSUPPORTED there).  GC_E_NOMEM: the whole list's buffers do not fit - search
 * PRN by PRN with gc_acq_shift_search / gc_acq_shift_row (float32 rows, the rules and no guard on the caller's side). */
enum { GC_SHIFT_PICK_GLOBAL = 0, GC_SHIFT_PICK_SEQUENTIAL = 1, GC_SHIFT_PICK_SEQUENTIAL_PAIRS = 2 };
typedef struct gc_acq_shift_pick {
  int32_t row;          /* winning row, -1: no value above 0 (the reference leaves the PRN's results at 0) */
  int32_t code_phase;   /* 0-based position of the row's first maximum */
  double peak;          /* its value - float64: the winning cell evaluated again as a float64 correlation (gc_acq_guard_stats) */
  double second_peak;   /* likewise; 0 with GC_SHIFT_PICK_GLOBAL */
} gc_acq_shift_pick;
int gc_acq_shift_search_batch(gc_context* ctx, int nprn, int narms, const int8_t* codes, int code_len, const int32_t* sample_index,
                              int n_index, const double* arm_weight, int rule, int exclude, int period, gc_acq_shift_pick* out /* [nprn] */);
/* What the prepared search writes: samples per row (n), rows of gc_acq_shift_search's outputs (n_carriers * n_signals *
 * n_bins) and the largest narms it takes - for callers that size their buffers (the MEX gateway); any pointer may be NULL. */
int gc_acq_shift_dims(gc_context* ctx, int32_t* n, int32_t* rows, int32_t* n_arms_max);

/* ---- bit synchronisation front end of navigation decoding ------------------------------
 * What every package's NAVdecoding.m does first with a channel's prompt in-phase stream: hard-limit it and cross-correlate it
 * with the sync pattern stretched to the stream's rate - GPS_L1CA/include/NAVdecoding.m:69-85 (160-sample TLM preamble),
 * GAL_E1C :79-88, GAL_E5a :69-95, GAL_E5b :80-100, BDS/B1I :71-105, BDS/B3I :72-97, GLO_GL1 :69-86.
 * out[l] = sum_k s(I_P[l+k]) * pattern[k] for lags l = 0 .. n-1 - the non-negative-lag half of xcorr(bits, pattern),
 * tlmXcorrResult(xcorrLength : 2*xcorrLength - 1) - with s(x) = +1 for x > 0 and -1 otherwise, or with
 * GC_SYNC_ZERO_IS_PLUS +1 for x >= 0 (Galileo E1: bits = (I_P < 0), GAL_E1C/include/NAVdecoding.m:84-85).
 * pattern: m <= 8192 values in {-1, 0, +1}.  The packages' thresholds, spacing rules and word checks (GPS :94-145 ...)
 * stay with the caller (nav_sync.py).  gc_preamble_xcorr is gc_sync_xcorr with flags = 0. */
#define GC_SYNC_ZERO_IS_PLUS 1
int gc_sync_xcorr(gc_context* ctx, const double* i_p, int64_t n, const int8_t* pattern, int m, int flags, float* out);
int gc_preamble_xcorr(gc_context* ctx, const double* i_p, int64_t n, const int8_t* pattern, int m, float* out);

/* Test hook (host only, no GPU): first sample i in [0, n) whose ramp value a + i*step is within eps chips of
 * an integer, or -1 — the exact near-tie analysis that lets the kernels skip their per-chunk filters. */
long long gc_debug_first_sample_near_edge(double a, double step, long long n, double eps);

/* Test hook (host only, no GPU): 1 if the padded table `t6` (n6 entries, read at six times the ramp rate) is the padded
 * table `t1` (n1 entries) with the sign pattern t6[k] = t1[(k + 5) / 6] * (-1)^((k + 5) / 6 + k) - BOC(6,1) next to
 * BOC(1,1) - which lets the lane kernel derive that arm instead of staging its table; 0 otherwise. */
int gc_debug_tables_derivable(const int8_t* t1, int n1, const int8_t* t6, int n6);

/* Test hook: which correlator kernel the last gc_correlate / gc_replay_launch / gc_track launch used:
 * 0 lane kernel (any chipping rate), 1 fast kernel with one-wave workgroups (float2 tables), 2 fast kernel with
 * four-wave workgroups and int8-pair tables, 3 the same with plain float tables, -1 exact per-sample kernel
 * (mixed ramp multipliers); -2 before the first launch.  Lets the parity tests prove which path they covered. */
int gc_debug_last_kernel(const gc_context* ctx);

/* Test hook: how the last gc_track / gc_track_device call on this context ran its loop: 0 a correlator launch per epoch,
 * 1 the persistent host-fed kernel, 2 the device loop; -1 before the first call.  A persistent kernel needs all its workgroups
 * resident: a grid that does not fit the device (many channels), or does not fit it together with other contexts' persistent
 * kernels in flight (gc_track_multi), is retried with half as many workgroups per channel until it fits; only a grid that no
 * team size brings under the limit runs with a launch per epoch instead (same records). */
int gc_debug_last_track_mode(const gc_context* ctx);
/* Test hook for the lane kernel's flush (csrc/corr_common.h: wave_transpose_sum): `k` (1..32) vectors of 64 floats, in[v * 64 + lane];
 * out[v] = the sum over the 64 lanes as the one-wave transposing reduction forms it (the tree's own order of additions). */
int gc_debug_wave_transpose_sum(gc_context* ctx, int k, const float* in, float* out);

/* Test hook: the library's four-step mixed-radix FFT on `nbatch` host sequences of n complex64
 * values (n of the form 2^a 3^b 5^c); output in natural order, unnormalised. */
int gc_debug_fft(gc_context* ctx, int n, int nbatch, const float* in, float* out, int inverse);

/* The float64 guard of the last search of this context (gc_acquire_coarse*, gc_acq_shift_search_batch): the searches transform in
 * float32, the reference decides in float64 (acquisition.m:196-206 max(max(results)), peakMetric > acqThreshold; BDS/B1I acquisition.m:
 * 141-166 max_peak / second).  Every PRN's winning cell is evaluated again in float64 as a circular correlation at one lag, so peak
 * and peak_metric leave the library without the transforms' rounding; a PRN whose runner-up lies within *eps (relative) of its winner
 * has every cell that close re-evaluated and the reference's first-occurrence rule applied to the float64 values.
 * ties: PRNs of the last search resolved that way; max_dev: largest |float32 peak - float64 peak| / float64 peak over its PRNs. */
int gc_acq_guard_stats(gc_context* ctx, int32_t* ties, double* max_dev, double* eps);

#ifdef __cplusplus
}
#endif
#endif /* GNSSCORR_H */
