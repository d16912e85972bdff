/*
 * tetrahip.h -- C-ABI of libtetrahip.so: MI355X (gfx950) TETRA IQ -> symbol front end.
 *
 * This is the drop-in boundary for ONE path of syrex1013/TetraEar:
 *     tetraear/signal/processor.py  SignalProcessor  (processor.py:18-273)
 * The reference has no FFI of its own (it is pure Python over numpy/scipy); the
 * boundary a maintainer binds is this header via ctypes (see INTEGRATION.md and
 * tetraear_amd/signal/processor.py, which keeps the SignalProcessor interface).
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI; every call returns 0 (TDM_OK) or a negative
 *     tdm_status; tdm_last_error() gives the text for the calling thread.
 *   - the caller owns every buffer it passes; a plan owns its device scratch.
 *   - a plan is not thread-safe (one caller at a time, like the reference instance,
 *     ui/modern.py:1857); the library is.
 *   - complex data: interleaved (re, im).  "c128" = two doubles per sample.
 *   - there is NO CPU fallback: without a gfx950 device every compute call fails with
 *     TDM_ERR_NO_DEVICE.
 */
#ifndef TETRAHIP_H
#define TETRAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDM_VERSION 102 /* 0.1.2: + tdm_plan_wait_for (0.1.1: tdm_plan_info grew by gardner_segments) -- a binding checks tdm_version() against the header it was written for */

#if defined(__GNUC__)
#define TDM_API __attribute__((visibility("default")))
#else
#define TDM_API
#endif

typedef enum tdm_status {
    TDM_OK = 0,
    TDM_ERR_INVALID = -1,    /* bad argument */
    TDM_ERR_NO_DEVICE = -2,  /* no usable gfx950 device / HIP runtime failure at init */
    TDM_ERR_HIP = -3,        /* a HIP call failed (text in tdm_last_error) */
    TDM_ERR_NOMEM = -4,
    TDM_ERR_UNSUPPORTED = -5
} tdm_status;

/* IQ sample formats accepted on the wire (signal/capture.py:143-158 hands over complex128
 * made by pyrtlsdr from cu8; cu8 is the RTL-SDR native format). */
typedef enum tdm_fmt {
    TDM_CU8 = 0,  /* uint8 I,Q;  value = u/127.5 - 1   (pyrtlsdr convention) */
    TDM_CS8 = 1,  /* int8  I,Q;  value = s/128.0 */
    TDM_CF32 = 2, /* float I,Q */
    TDM_CF64 = 3  /* double I,Q  (numpy complex128, the reference's own dtype) */
} tdm_fmt;

typedef enum tdm_mode {
    TDM_MODE_REFERENCE = 0, /* reproduces processor.py:221-273 (parity mode) */
    TDM_MODE_TETRA = 1,     /* RRC matched filter + feed-forward timing + Farrow + quadrant slicer on cf32
                               channelised baseband (no reference oracle; defined by oracle/tetra_np.py).
                               Arithmetic: fp32; the matched filter (16-bit coefficients) multiplies samples as sums
                               of two bf16 on the matrix cores with fp32 accumulation: soft symbols within 1e-5 of the
                               largest symbol of the fp64 definition (measured: median 3e-6, max 6e-6), independent of
                               the input's scale and of the alignment of its rows */
    TDM_MODE_TETRA_GARDNER = 2 /* the same receiver with the timing recovery BASELINE.json's north_star names: Gardner
                               timing-error detector -> proportional-integral loop -> period-controlled Farrow
                               interpolation (oracle/tetra_np.py demod_gardner), four lanes per carrier.  Same I/O as
                               TDM_MODE_TETRA; slower by construction (a recurrence over a carrier's symbols) */
} tdm_mode;

typedef struct tdm_plan tdm_plan;

/* Derived constants of a plan (processor.py:245-255, :74-75, :183, :194). */
typedef struct tdm_plan_info {
    double sample_rate;
    double rate_dec;      /* rate after the decimator (== sample_rate if not decimated) */
    int64_t n_samples;    /* samples per carrier per call */
    int64_t n_dec;        /* samples per carrier after the decimator */
    int32_t n_carriers;
    int32_t q;            /* decimation factor actually applied (1 = none) */
    int32_t sps;          /* int(rate_dec / 18000) */
    int32_t phase_step;   /* max(1, sps // 8) */
    int32_t max_soft;     /* capacity per carrier of the soft-symbol output (TDM_MODE_TETRA_GARDNER: room for a symbol clock 2 % fast) */
    int32_t lpf_applied;  /* 0 when n_dec <= 15 (reference falls back to unfiltered) */
    int32_t in_fmt;
    int32_t mode;
    int32_t device;
    int32_t dec_engine;   /* decimator kernel the next call runs: 0 none, 1 cascade engine, 2 parallel form on doubles,
                             3 parallel form on the raw bytes (cu8 batches of at least 8 blocks per CU) */
    int32_t gardner_segments; /* TDM_MODE_TETRA_GARDNER: the number of independently started loops every carrier's chunk is
                             walked as, joined at seams: by default the largest of 2, 4, 8 that leaves every loop its 384
                             warm-up symbols -- a function of the chunk (length, rate, taps) alone, never of the batch --
                             else 1 (oracle/tetra_np.py demod_gardner(segments=K)); 0 in the other modes */
} tdm_plan_info;

/* ---- library ---------------------------------------------------------------------------- */
TDM_API int tdm_version(void);
/* number of HIP devices, or a negative tdm_status */
TDM_API int tdm_device_count(void);
/* copies the calling thread's last error text (NUL-terminated) into buf; returns its length */
TDM_API int tdm_last_error(char *buf, size_t buflen);
/* Debug / experiment switches.  No counterpart in the reference (it has no kernel choices to make); the library never
 * reads the environment, so this call is the only way to override a kernel choice -- tests use it to put small inputs on
 * the kernels large batches take.  Process-wide; an unknown key is TDM_ERR_INVALID.  Every switch selects between
 * COMPLETE code paths with equal results (none skips work):
 *   "no_raw"          1: reference-mode cu8 plans created from now on never take the raw-byte decimator       (default 0)
 *   "raw_min_blocks"  >= 0: decimator blocks below which a batch stays on the double-based kernel, for plans
 *                     created from now on                                                                   (default -1: 8 per CU)
 *   "gardner_fused"   0: TDM_MODE_TETRA_GARDNER as three launches (matched filter -> HBM -> loop -> decisions)  (default 1)
 *   "gardner_segments" what tdm_plan_option "gardner_segments" sets per plan, for TDM_MODE_TETRA_GARDNER plans created from
 *                     now on: 0 whole chunks, 1 the default, K > 1 at most K pieces, -1 fitted to the batch   (default 1)
 *   "pfb_direct"      1: channeliser plans created from now on use the direct-DFT kernel                     (default 0)
 *   "pfb_rounds"      > 0: rounds per channeliser workgroup, for plans created from now on                  (default 0: computed) */
TDM_API int tdm_debug_set(const char *key, int64_t value);
TDM_API int tdm_debug_get(const char *key, int64_t *value);

/* ---- plan: one (sample_rate, n_samples, n_carriers, format) configuration ----------------
 * replaces SignalProcessor.__init__ (processor.py:21-33) + the per-call filter design
 * (processor.py:78, :254).  n_carriers independent streams are processed per call.
 * in_fmt: reference mode takes all four wire formats; the TETRA modes take TDM_CF32 (the channeliser's output) or
 * TDM_CU8 / TDM_CS8 straight off the wire (converted where the kernels stage their window: cu8 as u / 127.5 - 1, cs8 as s / 128;
 * TDM_MODE_TETRA_GARDNER's fused kernel takes bytes at 33 and 35 taps -- 72 and 80 kS/s --, other rates run as three launches,
 * whole chunks).                                                                            */
TDM_API int tdm_plan_create(double sample_rate, int64_t n_samples, int32_t n_carriers, int32_t in_fmt,
                    int32_t mode, int32_t device, tdm_plan **out);
TDM_API int tdm_plan_destroy(tdm_plan *plan);
/* Per-plan options (no counterpart in the reference).
 *   "fast_pre_shift"  (TDM_MODE_REFERENCE, default 0) 1: the input-rate pre-shift of tdm_process* (pre_shift_hz: the
 *       frequency_shift(x, f_k) of processor.py:85-100 fused into the decimator's load) runs its phase as the IDEAL ramp
 *       from an exactly anchored first sample per lane, instead of reproducing the reference's own rounding of
 *       theta_j = fl(ci * fl(j / fs)) sample by sample (a third of that kernel's arithmetic).  The two phases differ by the
 *       reference's rounding error of theta: at most 6e-11 rad at 787.5 kHz x 0.1 s.  Soft symbols then agree with the
 *       reference to 1e-9 instead of 1e-10 (north_star: 1e-5); a HARD decision can differ only where its margin to a
 *       threshold is below that, which the per-carrier min_margin output reports: a caller that needs the reference's
 *       decision there re-runs the carriers with min_margin < 1e-8 on a plan without the option
 *       (tetraear_amd.batch.BatchDemodulator.process does).
 *   "rows_per_chunk"  (TDM_MODE_REFERENCE, default 1) C > 1: the plan's n_carriers rows are T x C -- C carriers out of each of T
 *       CONSECUTIVE CHUNKS of one stream in one call: plan rows r C ... r C + C - 1 all read input row r
 *       (iq + r * carrier_stride_samples), each with its own pre_shift_hz / freq_offset_hz entry (both stay per plan row;
 *       every row's phase starts at its chunk's first sample, as in the reference's per-call frequency_shift).  This is the
 *       reference's caller loop (ui/modern.py:1908-1912: chunk after chunk through process()) for many carriers of one
 *       wideband stream, T iterations per launch instead of one: a 64-carrier step is launch-latency, a 256-row step is not.
 *       C must divide n_carriers.  The raw-byte decimator is not used by such a plan (its rows are rotated samples anyway).
 *   "gardner_segments"  (TDM_MODE_TETRA_GARDNER) how many independently started loops a carrier's chunk is walked as.
 *       1 (the state after tdm_plan_create): the largest of 2, 4, 8 pieces that leaves every loop its 384 warm-up symbols
 *       -- decided by the chunk's length, rate and tap count ALONE, so the same carrier gives the same symbols bit for bit
 *       whatever the plan's carrier count and whatever the device (tap counts above 41, whose fused kernel serves one round
 *       of workgroups only, keep whole chunks); 0: whole chunks; K = 2..8: at most K pieces, by the same rule;
 *       -1: FITTED TO THE BATCH -- the number of pieces that is fastest for this plan's carrier count on this device
 *       (e.g. 2 at 4096 carriers x 32 768 samples, where the default's 8 cost 20 % more time for their warm-ups): the one
 *       setting under which a carrier's soft symbols behind a seam depend on the plan's size.  Waits for the device;
 *       tdm_plan_get_info reports the number now in force.  In pieces the symbols before the first seam are those of the
 *       whole-chunk path bit for bit, behind a seam the soft symbols agree with it to about 1 % of the largest symbol
 *       (DESIGN.md 4.8).
 *   "gardner_ff_start"  (TDM_MODE_TETRA_GARDNER, default 0) 1: the first loop of every chunk starts at a feed-forward
 *       (square-law) timing estimate over the chunk's first 512 filter outputs instead of at sample 1 + sps -- the later
 *       pieces of a chunk always do.  The Gardner detector's error vanishes half a symbol off the eye as well as on it, so a
 *       loop started there can sit for hundreds of symbols before it pulls in; with the option a chunk's first few hundred
 *       symbols are as good as the rest (oracle/tetra_np.py demod_gardner(ff_first=True)).  Needs the fused kernel. */
TDM_API int tdm_plan_option(tdm_plan *plan, const char *key, int64_t value);
/* Host only (no device): the geometry of a TDM_MODE_TETRA_GARDNER chunk of n_samples walked in `pieces` independently started
 * loops -- out[0..5] = piece length, samples from a piece's start to the next one's, the incoming and the outgoing seam in a
 * piece's own coordinates, the seam's distance from a piece's end, warm-up + margin -- the arithmetic of
 * oracle/tetra_np.py gardner_segments (a CPU-tier test holds the two together); TDM_ERR_UNSUPPORTED when the chunk is too
 * short for that many pieces. */
TDM_API int tdm_gardner_geometry(double sample_rate, int64_t n_samples, int32_t pieces, int32_t *out);
TDM_API int tdm_plan_get_info(const tdm_plan *plan, tdm_plan_info *info);
/* Serve another chunk length with the same plan (TDM_MODE_REFERENCE): the reference designs its filters inside every
 * process() call (processor.py:78, :254), so its callers read whatever lengths they like (ui/modern.py:1912 128 Ki,
 * scanner.py:347 min(fs*dwell, 256 Ki), rtl_auto_capture.py:182 args.chunk).  Only the last block's tables and the edge
 * maps depend on the length: the first call for a length builds and uploads those (a fraction of a millisecond, one
 * allocation), a length seen before is a look-up; the large tables are uploaded once per plan and the work buffers are
 * shared between lengths (they grow when a longer chunk arrives).  tdm_plan_get_info then describes the new length.
 * Kernels already enqueued keep running; do not call it concurrently with a process call on the same plan.        */
TDM_API int tdm_plan_resize(tdm_plan *plan, int64_t n_samples);

/* ---- SignalProcessor.process (processor.py:221-273), batched over carriers ----------------
 *  iq            [n_carriers] streams of n_samples in the plan's format; carrier c starts at
 *                iq + c * carrier_stride_samples samples (0 = all carriers share one stream;
 *                TDM_MODE_TETRA: any stride >= n_samples, e.g. the row pitch of tdm_channelise_batch)
 *  n_samples     at most 2^31 per plan
 *  pre_shift_hz  per carrier, or NULL: frequency_shift(x, f) at the INPUT rate before process()
 *                (processor.py:85-100; the composition SURVEY.md 8(d) C3 uses to channelise)
 *  freq_offset_hz per carrier, or NULL: process()'s freq_offset (applied after the decimator)
 *  hard          [n_carriers][max_soft] uint8 symbols 0..3   (n_hard = max(n_soft-1, 0))
 *  soft          [n_carriers][max_soft] c128, = SignalProcessor.symbols (processor.py:268);
 *                in TDM_MODE_TETRA: cf32 (float I,Q) symbol-spaced matched-filter outputs
 *  n_soft        [n_carriers]
 *  best_phase    [n_carriers] timing phase chosen (processor.py:196-210), may be NULL
 *  min_margin    [n_carriers] min |phase - threshold| over the decisions (rad), may be NULL
 * tdm_process:        all pointers are HOST memory; blocking.
 * tdm_process_device: all pointers are DEVICE memory (pre_shift/freq_offset too); enqueues on
 *                     `stream` (a hipStream_t, NULL = default) and returns; tdm_plan_sync waits. */
TDM_API int tdm_process(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples,
                const double *pre_shift_hz, const double *freq_offset_hz, uint8_t *hard, double *soft,
                int32_t *n_soft, int32_t *best_phase, double *min_margin);
TDM_API int tdm_process_device(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples,
                       const double *pre_shift_hz, const double *freq_offset_hz, uint8_t *hard,
                       double *soft, int32_t *n_soft, int32_t *best_phase, double *min_margin,
                       void *stream);
TDM_API int tdm_plan_sync(tdm_plan *plan);
/* The RRC matched filter of a TDM_MODE_TETRA / TDM_MODE_TETRA_GARDNER plan on its own (the receiver's first stage as a
 * stand-alone operator: LDS-tiled sliding window, cf32 in, cf32 out at the sample rate; oracle/tetra_np.py matched_filter
 * with the plan's taps): iq [n_carriers] rows of the plan's chunk length, carrier_stride_samples apart; y [n_carriers]
 * rows, y_pitch samples apart (even, >= the chunk length).  DEVICE pointers; enqueues on `stream` (NULL = the plan's) and
 * returns.  No counterpart in the reference (it has no matched filter). */
TDM_API int tdm_plan_rrc_filter(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, float *y, int64_t y_pitch,
                                void *stream);
/* Stream ordering of the stand-alone entry points.  Plans run on their own (non-blocking) stream, which has no implicit
 * ordering with the null stream.  The device-pointer forms of tdm_spectrum_gate, tdm_channelise(_batch) and
 * tdm_find_sync enqueue on the calling thread's CURRENT stream: the null stream until tdm_set_stream names another.
 * To chain them with tdm_process_device on the device without any host synchronisation, make the plan's stream
 * current:    tdm_plan_stream(plan, &s); tdm_set_stream(s);   (tdm_set_stream(NULL) restores the null stream)
 * Without that, a tdm_dev_sync / tdm_plan_sync is required between stages that run on different streams.       */
TDM_API int tdm_set_stream(void *stream);
TDM_API int tdm_plan_stream(tdm_plan *plan, void **stream);
/* Device-side ordering between two plans of one device (no host synchronisation): what is enqueued on `plan`'s stream from
 * now on starts after everything enqueued so far on `other`'s stream has finished.  Plans are independent objects on their
 * own streams and normally overlap (a large batch run as two plans of half the carriers each is faster than as one, see
 * tetraear_amd/batch.py SplitBatchDemodulator); this call is for the places where they must not -- a consumer plan behind a
 * producer plan, or timing one plan's launches alone on the device.  No counterpart in the reference. */
TDM_API int tdm_plan_wait_for(tdm_plan *plan, tdm_plan *other);
/* Host-fed streaming: n_batches consecutive batches, each laid out like one
 * tdm_process call (n_carriers x n_samples back to back; outputs [n_batches][n_carriers][max_soft]...).
 * The host->device copy of batch i+1 and the device->host copy of batch i-1 overlap the kernels of
 * batch i (two device slots, three streams; the caller's buffers are pinned in place for the call).
 * freq_offset_hz is per carrier and applies to every batch.  Consecutive 256 Ki-sample chunks of one
 * recording are independent (the reference is stateless per read, processor.py:221-273), so a long
 * capture is simply fed as rows of successive batches.                                              */
TDM_API int tdm_process_pipelined(tdm_plan *plan, const void *iq, int64_t n_batches, const double *freq_offset_hz,
                                  uint8_t *hard, void *soft, int32_t *n_soft, int32_t *best_phase,
                                  double *min_margin);

/* ---- the other public methods of SignalProcessor, one call each (host pointers, blocking) ---
 * All take/return c128 host arrays.                                                          */
/* filter_signal (processor.py:51-83). *applied = 0 when the reference's except-branch would
 * return the input unchanged (n <= 15). */
TDM_API int tdm_filter_signal(const double *x, int64_t n, double bandwidth, double fs, double *y,
                      int32_t *applied, int32_t device);
/* frequency_shift (processor.py:85-100) */
TDM_API int tdm_frequency_shift(const double *x, int64_t n, double freq_offset, double fs, double *y,
                        int32_t device);
/* extract_symbols (processor.py:168-219): y has room for n samples */
TDM_API int tdm_extract_symbols(const double *x, int64_t n, double fs, double symbol_rate, double *y,
                        int64_t *n_out, int32_t *best_phase, int32_t device);
/* demodulate_dqpsk (processor.py:102-166): out has room for n-1 symbols */
TDM_API int tdm_demodulate_dqpsk(const double *x, int64_t n, uint8_t *out, int64_t *n_out, double *min_margin,
                         int32_t device);
/* scipy.signal.decimate(x, q) as called at processor.py:254 (y has room for ceil(n/q));
 * returns TDM_ERR_INVALID when n <= 27 (scipy raises there). */
TDM_API int tdm_decimate(const double *x, int64_t n, int32_t q, double *y, int64_t *n_out, int32_t device);
/* resample (processor.py:35-49): scipy.signal.resample (FFT method) to `num` points.  Short inputs as direct sums with exact
 * twiddles; from 2^24 terms on as fast transforms of arbitrary length (radix-2 passes for powers of two, Bluestein's chirp-z form
 * otherwise), both in fp64: 1e-15 from scipy on 131 072-sample arrays. */
TDM_API int tdm_resample(const double *x, int64_t n, int64_t num, double *y, int32_t device);

/* ---- spectrum / AFC / signal gate in front of process() -----------------------
 * The block of CaptureThread.run (tetraear/ui/modern.py:1921-2021) that decides whether process() is
 * called and with which freq_offset: 2048-point Hann FFT of the first 2048 samples of each row, dBFS,
 * +-12.5 kHz band mean / peak / peak bin, out-of-band noise floor, SNR rule.
 *  out [rows][8] = peak_freq_offset_hz, signal_power_db, peak_power_db, noise_floor_db, snr_db,
 *                  strong (0/1), afc_offset_hz, 0        afc [rows] (may be NULL) = afc_offset_hz
 * With device pointers the launch is asynchronous on the current stream (tdm_set_stream) and `afc` can be
 * passed straight to a tdm_process_device that runs on the same stream as freq_offset_hz.             */
TDM_API int tdm_spectrum_gate(const void *iq, int32_t in_fmt, int64_t row_stride, int64_t n_samples, int32_t rows,
                              double sample_rate, double *out, double *afc, int32_t device_pointers,
                              int32_t device);

/* ---- scanner heuristics: TetraSignalDetector.calculate_power,
 * detect_tetra_modulation, detect_sync_pattern (tetraear/signal/scanner.py:42-147), `rows` rows of n
 * complex128 host samples.  out [rows][8] = power_db, is_tetra (0/1), confidence, found_sync (0/1),
 * max_correlation, 0, 0, 0.                                                                          */
TDM_API int tdm_detect(const double *x, int64_t n, int32_t rows, double sample_rate, double *out, int32_t device);

/* ---- burst synchronisation, the immediate consumer of process() -------------
 * TetraDecoder.symbols_to_bits + TetraDecoder.find_sync (tetraear/core/decoder.py:140-169, :171-295),
 * batched over rows.  units = hard symbols 0..3 (from_bits = 0; bit stream = (s>>1, s&1) per symbol) or
 * the bit stream itself, one byte per bit (from_bits = 1).  Row r holds n_units[r] entries at
 * units + r*row_stride.  positions [rows][max_pos] receives the accepted bit positions in order
 * (n_pos[r] may exceed max_pos: only max_pos are stored), max_corr[r] the reference's max correlation.
 * device_pointers != 0: units/n_units/positions/n_pos/max_corr are device memory and the call is asynchronous on the
 * current stream (tdm_set_stream); rows are then bounded by row_stride.  from_bits | 2: n_units holds symbols + 1,
 * i.e. the `n_soft` buffer of tdm_process_device can be passed as it is (with its `hard` buffer as units).      */
TDM_API int tdm_find_sync(const uint8_t *units, int64_t row_stride, const int32_t *n_units, int32_t rows,
                          int32_t from_bits, double threshold, int32_t max_pos, int32_t *positions,
                          int32_t *n_pos, double *max_corr, int32_t device_pointers, int32_t device);

/* ---- tetra-mode channeliser: oversampled polyphase DFT filter bank (no counterpart in the reference) --
 * One wideband stream (cu8 / cs8 / cf32, n_in samples at fs) -> M channels spaced fs/M, each decimated by
 * D (output rate fs/D), out [M][n_out] cf32 with n_out = ceil(n_in/D); channel k is centred on k*fs/M
 * (k >= M/2: negative frequencies).  Built for M in {72, 80, 96, 128, 400}.                          */
TDM_API int tdm_channelise(const void *iq, int32_t in_fmt, int64_t n_in, int32_t M, int32_t D, float *out,
                           int64_t *n_out, int32_t device_pointers, int32_t device);
/* n_streams independent streams in one launch: iq [n_streams][n_in], out [n_streams][M][out_pitch]
 * (out_pitch = row pitch in complex samples, >= n_out; 0 = n_out).  A pitch that is a multiple of 16
 * keeps every 128-byte store of the kernel inside one cache line (1.5x the dense layout's rate).      */
TDM_API int tdm_channelise_batch(const void *iq, int32_t in_fmt, int64_t n_in, int32_t n_streams, int32_t M,
                                 int32_t D, float *out, int64_t out_pitch, int64_t *n_out,
                                 int32_t device_pointers, int32_t device);

/* ---- occupancy gate of the wideband chain (many-carrier form) ------------------------------------
 * Which channel rows of tdm_channelise_batch's output carry a signal: the decision CaptureThread.run makes before it calls
 * process() (tetraear/ui/modern.py:1921-2003), per channel row at the channel rate -- Hann-windowed FFT of the row's first
 * 256 samples, power = 20 log10(|X|/N + 1e-20) (:1926-1934), signal_power / peak_power = mean / max over the bins within
 * 25 kHz around the centre (:1948-1957), occupied = snr > snr_db and peak_power > min_dbfs and peak_power - signal_power > 3
 * (:1992-1995; the reference: 15 dB, -70 dBFS) -- with the noise floor taken as the MEDIAN of signal_power over the stream's
 * M channels (the reference's floor, the bins outside the centre channel, would count neighbouring carriers as noise).
 * Definition: oracle/pfb_np.py occupancy().
 *   chan      [n_streams * M][pitch] cf32 channel rows (n_out >= 256 valid samples each), chan_rate their sample rate
 *   stats     [rows][2] float: signal_power, peak_power in dBFS;  flags [rows] 1 = occupied
 *   row_list  [rows] int32: the occupied rows (no particular order), n_rows [1] their number
 *   n_soft    [rows] or NULL: set to 0 for the rows that are NOT occupied (so that a download after
 *             tdm_process_device_rows reads "no symbols" there)
 * device_pointers != 0: everything is device memory and the two launches are asynchronous on the current stream
 * (tdm_set_stream) -- row_list / n_rows then feed tdm_process_device_rows without a host round trip.                      */
TDM_API int tdm_occupancy_gate(const float *chan, int64_t pitch, int32_t n_streams, int32_t M, int64_t n_out,
                               double chan_rate, double snr_db, double min_dbfs, float *stats, uint8_t *flags,
                               int32_t *row_list, int32_t *n_rows, int32_t *n_soft, int32_t device_pointers, int32_t device);
/* tdm_process_device over the listed rows only (TDM_MODE_TETRA plans): row_list / n_rows are DEVICE memory (the gate's
 * outputs); row r of the batch is read from iq + r * carrier_stride_samples and written to row r of the outputs exactly as
 * tdm_process_device would -- rows that are not listed are not touched.  Replaces the reference's "process() only when a
 * signal is present" (ui/modern.py:2016-2022) for many carriers.                                                          */
TDM_API int tdm_process_device_rows(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, const int32_t *row_list,
                                    const int32_t *n_rows, uint8_t *hard, float *soft, int32_t *n_soft,
                                    int32_t *timing_milli, double *min_margin, void *stream);

/* ---- device memory helpers for callers without a HIP binding (bench, tests) ---------------- */
TDM_API int tdm_dev_alloc(int32_t device, size_t bytes, void **ptr);
TDM_API int tdm_dev_free(int32_t device, void *ptr);
TDM_API int tdm_dev_upload(int32_t device, void *dst_dev, const void *src_host, size_t bytes);
TDM_API int tdm_dev_download(int32_t device, void *dst_host, const void *src_dev, size_t bytes);
TDM_API int tdm_dev_sync(int32_t device);
/* Page-lock a host buffer the caller keeps reusing for tdm_process (a capture loop's read buffers: the recording reader
 * of tetraear_amd/ingest.py keeps two of them, filled in turn, like the read buffer of decrypt_capture.py:101-107):
 * host->device copies out of registered memory run at the link's rate without a bounce through a staging page. */
TDM_API int tdm_host_register(int32_t device, void *ptr, size_t bytes);
TDM_API int tdm_host_unregister(int32_t device, void *ptr);
/* Measured HBM ceilings of the box ("public-spec peaks must be replaced by a measured on-box copy-kernel
 * ceiling for the denominator"): grid-stride kernels with 16-byte accesses over two buffers of `bytes` each (use more
 * than the 256 MiB of the last-level cache), `reps` timed launches after 3, HIP events on the current stream.
 * gbs[0] = copy (bytes read + bytes written per second), gbs[1] = read only, gbs[2] = write only, in GB/s.        */
TDM_API int tdm_hbm_ceiling(int32_t device, size_t bytes, int32_t reps, double *gbs);
/* event timing on the library's own stream for a plan: elapsed milliseconds between two marks */
TDM_API int tdm_plan_time_begin(tdm_plan *plan);
/* the same mark without per-stage events: the launches of the timed region go out back to back exactly as they do
 * untimed (the events around every launch cost a one-carrier call a third of its time) */
TDM_API int tdm_plan_time_begin_total(tdm_plan *plan);
TDM_API int tdm_plan_time_end(tdm_plan *plan, float *elapsed_ms);
/* per-stage kernel time (ms) accumulated between time_begin/time_end; names[i] static strings */
TDM_API int tdm_plan_stage_times(tdm_plan *plan, int32_t max_stages, const char **names, float *ms,
                         int32_t *n_stages);

/* ---- introspection used by the CPU tests (no device needed) -------------------------------- */
/* filter design the plan would use: sos[4][6], soszi[4][2] (zeros if q == 1), b[5], a[5], zi[4] */
TDM_API int tdm_design_dump(double sample_rate, int64_t n_samples, double *sos, double *soszi, double *b,
                    double *a, double *zi, int32_t *q, double *rate_dec);
TDM_API int tdm_design_butter(double bandwidth, double fs, double *b, double *a, double *zi);

#ifdef __cplusplus
}
#endif
#endif /* TETRAHIP_H */
