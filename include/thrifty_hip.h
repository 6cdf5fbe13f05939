/*
 * thrifty_hip.h -- C ABI of the MI355X (gfx950) matched-filter detection engine.
 *
 * This is the drop-in boundary for the `thrifty detect` hot path.  Each entry
 * point names the reference interface it stands in for (paths relative to the
 * swkrueger/Thrifty checkout).  Conventions follow the reference's native twin
 * (fastcard/fastcard.h:46-53, fastdet/corr_detector.h:24-38): an opaque handle
 * created from a caller-owned settings struct, `int` status returns
 * (0 = ok, <0 = error; the message is available from thr_last_error()), the
 * library owns all FFT/work buffers, a handle is single-threaded / not
 * re-entrant, and distinct handles are fully independent (one per device and
 * stream), so a host may drive 8 GPUs from 8 processes or threads.
 *
 * No exceptions cross this boundary and no torch / C++ types appear in it: the
 * entry points are function-try-blocks (csrc/handle.hip: thr::on_exception) -- host
 * memory exhaustion or a thread the OS refuses come back as THR_ERR_DEVICE with a
 * message, a half-built handle or input window is torn down first.
 */
#ifndef THRIFTY_HIP_H
#define THRIFTY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + thr_create_preshift / thr_create_fastdet, thr_detect_stream[_device], thr_detect_card,
 *    thr_identify (additions only: every version-1 entry point is unchanged) */
/* (3: THR_N_KERNEL_SLOTS 4 -> 5 (the arrays of thr_profile_read grow; slot 4 = the long-block
 *    combination kernel); everything else unchanged) */
/* 4: + thr_frame_card (addition only) */
/* 5: + thr_submit / thr_submit_card / thr_submit_stream / thr_collect / thr_inputs_consumed / thr_poll (asynchronous host
 *    boundary), thr_set_stream_default, thr_format_toad (additions only) */
/* 6: + thr_create_ex (explicit variant and kernel-path selection), thr_plan_sections,
 *    thr_host_register / thr_host_unregister, thr_input_window (additions only).
 *    No environment variable changes what a handle computes or how it schedules any more. */
/* 7: + thr_run_card / thr_run_stream (the whole file -> .toad loop in one call, text on a library
 *    thread), thr_get_settings, thr_input_window_ex (populator threads / segment size) / _release,
 *    thr_detect_offsets, thr_set_wait_mode, THR_PATH_GENERIC_ROWS, THR_ERR_INDEX, THR_FLAG_INT_OFFSET
 *    (additions only) */
/* 8: block_len 16384 with ONE template short enough for four 4096-sample sections (and no stddev
 *    term in the correlation threshold): THR_PATH_AUTO runs the correlate stage sectioned
 *    (csrc/detect16k_sec.hip); thr_plan_sections answers for block_len 16384 too,
 *    THR_PATH_UNSECTIONED applies to it, + THR_PATH_UNSECTIONED_GENERIC_ROWS, thr_debug_sections
 *    (additions; records of such handles change in the last bits of their float fields only) */
/* 9: block_len 16384: SEVERAL short templates run the sectioned correlate stage too (one forward
 *    transform per section, one product + inverse per template); + THR_FLAG_FIT_UNCONVERGED (the
 *    carrier fit's MINPACK exit code), thr_get_path_info (which kernels a handle's launches take, and
 *    why); thr_detect_offsets refuses non-finite offsets (additions; records of several-template
 *    handles change in the last bits of their float fields only) */
#define THR_ABI_VERSION 9

/* status codes */
#define THR_OK 0
#define THR_ERR_ARG (-1)      /* bad argument / unsupported configuration      */
#define THR_ERR_DEVICE (-2)   /* HIP runtime error (no GPU, OOM, launch failure) */
#define THR_ERR_STATE (-3)    /* call sequence error                            */
#define THR_ERR_INDEX (-4)    /* thr_run_*: a block on which the reference raises IndexError
                                 (THR_FLAG_INDEX_ERROR) ended the run                */

/* thr_record.flags */
#define THR_FLAG_CARRIER 1u      /* carrier_detect verdict (carrier_detect.py:95)  */
#define THR_FLAG_CORR 2u         /* correlation peak verdict (soa_estimator.py:85) */
#define THR_FLAG_INDEX_ERROR 4u  /* reference raises IndexError here: carrier bin
                                    + 3 >= block_len (carrier_sync.py:187)         */

#define THR_FLAG_FIT_UNCONVERGED 16u /* the Dirichlet carrier fit ended with MINPACK lmdif exit code 5 .. 8
                                    (600 evaluations, or a tolerance below machine precision): SciPy's
                                    curve_fit raises RuntimeError("Optimal parameters not found: ...")
                                    there and the reference does not catch it (carrier_sync.py:189), so
                                    its detect loop dies on that block.  The record is complete -- shift,
                                    FFT#2 and correlation ran with lmdif's last iterate -- and says so;
                                    Detector(strict_fit=True) raises like the reference.  Degenerate
                                    geometries only (templates far shorter than block_len / 24: the seven
                                    fitted points sit on a flat main lobe), and there the exit code hangs
                                    on the last digit of the seven magnitudes: the flagged blocks are the
                                    reference's to within that noise, not block for block. */
#define THR_FLAG_INT_OFFSET 8u   /* PreshiftDetector, cosine interpolator: the reference returned the
                                    Python int 0 (cos(omega) > 1, carrier_interpolators.py:87-88), so
                                    carrier_offset is 0 and prints as "0", not "0.0"                  */

/* input sample formats for thr_detect*() */
#define THR_IN_U8 0   /* interleaved unsigned 8-bit I,Q (RTL-SDR; block_data.py:38-52) */
#define THR_IN_C64 1  /* interleaved float32 re,im (a `Signal` already converted)     */

/*
 * Detector configuration.  Mirrors `DetectorSettings` (thrifty/detect.py:24-31)
 * plus what `Detector.__init__` derives from it (detect.py:40-58); the native
 * twin's equivalent is fargs_t + CorrDetector's ctor (corr_detector.h:26-30).
 */
typedef struct thr_settings {
    int32_t block_len;        /* samples per block, power of two                   */
    int32_t history_len;      /* samples repeated from the previous block          */
    int32_t n_templates;      /* >= 1 (the reference has exactly 1)                */
    int32_t template_len;     /* samples per template (all templates equal length) */
    const double* templates;  /* [n_templates][template_len], real, time domain    */
    int32_t carrier_len;      /* Dirichlet-kernel width; 0 = template_len          */
    int32_t carrier_window[2];/* closed bin interval, negative = wrap
                                 (carrier_detect.py:17-58); {0,-1} = all bins       */
    double carrier_thresh[3]; /* (constant, snr, stddev) carrier_detect.py:110-115 */
    double corr_thresh[3];    /* (constant, snr, stddev) soa_estimator.py:127-134  */
    int32_t device_id;        /* HIP device ordinal                                */
    int32_t max_batch;        /* largest number of blocks per thr_detect*() call   */
} thr_settings;

/*
 * One detection record per (block, template): the payload of
 * `DetectionResult` / `CarrierSyncInfo` / `CorrDetectionInfo`
 * (thrifty/toads_data.py:8-45) and of fastdet's CorrDetection
 * (corr_detector.h:14-22).  64 bytes, written densely as out[block][template].
 */
typedef struct thr_record {
    int64_t block_idx;      /* caller's block index (passed through)              */
    uint32_t flags;         /* THR_FLAG_*                                          */
    int32_t template_id;    /* 0 .. n_templates-1                                  */
    int32_t carrier_bin;    /* CarrierSyncInfo.bin                                 */
    int32_t corr_sample;    /* CorrDetectionInfo.sample (-1 if no carrier)         */
    double carrier_offset;  /* CarrierSyncInfo.offset (0 if no carrier)            */
    double corr_offset;     /* CorrDetectionInfo.offset, clipped to +-0.6          */
    float carrier_energy;   /* CarrierSyncInfo.energy (peak magnitude)             */
    float carrier_noise;    /* CarrierSyncInfo.noise  (rms)                        */
    float corr_energy;      /* CorrDetectionInfo.energy (peak magnitude)           */
    float corr_noise;       /* CorrDetectionInfo.noise  (rms)                      */
    uint64_t reserved;      /* 0 (default detector); see thr_create_preshift       */
} thr_record;

typedef struct thr_handle thr_handle;

/* Library / ABI identification. */
int thr_abi_version(void);
const char* thr_last_error(void);

/*
 * Replaces `Detector.__init__` (detect.py:40-58: DefaultSynchronizer +
 * SoaEstimator construction incl. template zero-pad + FFT,
 * soa_estimator.py:63-76) and fastcard_new()/CorrDetector::CorrDetector
 * (fastcard.c:13-117, corr_detector.cpp:31-86).  `settings` and the template
 * array need only live for the duration of the call.
 */
int thr_create(const thr_settings* settings, thr_handle** out);
/*
 * PreshiftDetector variant -- replaces `PreshiftDetector.__init__` + `TemplateShifts`
 * (thrifty/experimental/detect_preshift.py:24-60; defaults num=21, parabolic carrier
 * interpolator experimental/carrier_interpolators.py:40-45, corr_shift off): the
 * carrier offset is a 3-point parabola on |FFT#1|, FFT#1 is rolled by the rounded shift
 * (freq_shift_integer, carrier_sync.py:241-245) and correlated against the nearest of
 * `num_shifts` template spectra pre-shifted by -0.5 .. +0.5 bin.  Every other entry
 * point then behaves as documented with these semantics: carrier_offset is the float32
 * parabola offset; THR_FLAG_INDEX_ERROR marks carrier_bin + 1 >= block_len (where the
 * reference's fft_mag[peak+1] raises); `reserved` = (uint32 rolled shift << 32) | bank
 * index.  One template only; thr_debug_stage dumps (np.roll(FFT#1, shift) and the correlation,
 * what the reference class returns under yield_data) need a THR_PATH_MULTIPASS handle.  block_len 16384
 * runs ONE fused kernel per block (FFT, verdict, gather-multiply, IFFT); other lengths
 * use the multi-pass pipeline.
 */
int thr_create_preshift(const thr_settings* settings, int num_shifts, thr_handle** out);
/*
 * fastdet-compatible variant -- the algorithm of the reference's NATIVE detector
 * (fastcard/cardet.c:7-41 + fastdet/corr_detector.cpp:88-197), which differs from the Python
 * one: float32 power-domain verdicts `max > c + s * noise_power` for carrier and correlation
 * (carrier_thresh / corr_thresh hold c, s in the POWER domain; the third coefficient must be
 * 0), carrier noise (sum - 2 max)/(N - 1), correlation noise clamped at 0 and computed from
 * the integer-truncated peak power, integer roll by -argmax against the unshifted template
 * (no FFT#2), carrier offset = parabola on sqrt(power) and correlation offset = Gaussian on
 * log sqrt(power), both clipped to +-0.5, window [min, max] non-wrapping (a window with
 * min < 0 <= max is refused like cardet_normalize_window, cardet.c:44-48).  Records then hold
 * what fastdet prints (fastdet.cpp:188-206): energies / noises are the square roots of the
 * powers.  Same kernel structure as the preshift variant (one fused kernel at 16384).
 * Parity status: restated from the C/C++ sources; fastdet itself needs FFTW3f, VOLK and
 * librtlsdr and cannot be built in this environment, so this variant is NOT pinned against
 * reference output.
 */
int thr_create_fastdet(const thr_settings* settings, thr_handle** out);
/*
 * The three constructors above in one call, plus an explicit choice of kernel path.  `variant`:
 * THR_VARIANT_DEFAULT (thr_create), THR_VARIANT_PRESHIFT (thr_create_preshift; variant_arg =
 * num_shifts | THR_INTERP_* << 16: the bank size and the carrier interpolator, one of the reference's
 * thrifty/experimental/carrier_interpolators.py -- parabolic (:40-45, what thr_create_preshift uses
 * and the reference's default), none (:17-18), gaussian (:48-54), cosine (:84-92); float32 like the
 * magnitudes they are given) or THR_VARIANT_FASTDET (thr_create_fastdet).  `path`:
 *   THR_PATH_AUTO         what the other constructors use: the fastest kernels for the block
 *                         length (LDS-resident for 1024 ... 65536, multi-pass otherwise);
 *   THR_PATH_MULTIPASS    the generic multi-pass pipeline (Stockham passes through HBM) whatever
 *                         the block length -- an independent implementation of the same arithmetic,
 *                         kept for cross-checking the fused kernels on identical input;
 *   THR_PATH_UNSECTIONED  the correlate stage as ONE block_len-point transform pair instead of
 *                         overlap-save sections: block_len 32768 / 65536 (decimated sub-transforms,
 *                         detect_long.hip, instead of sections of 16384 points, detect_seg.hip --
 *                         AUTO falls back to it by itself for templates longer than 9361 samples at
 *                         65536) and block_len 16384 (k_correlate instead of up to four sections of
 *                         4096 points, detect16k_sec.hip, which AUTO takes for templates -- one or
 *                         several, ABI 9 -- of at most about 1000 samples with no stddev threshold
 *                         term; thr_get_path_info says which a handle got and why).  thr_debug_stage
 *                         dumps always come from the unsectioned kernels.
 *   THR_PATH_GENERIC_ROWS AUTO, except that the correlate kernel of block_len 16384 (and of the
 *                         sections of longer blocks) is the generic one, with the unique-window test
 *                         in all 16 rows of lags, instead of the window-row specialisation the
 *                         geometry selects (csrc/correlate16k_geom.hpp): same arithmetic, so the
 *                         records are equal byte for byte -- which is what the tests use it for.
 *                         (A sectioned block_len 16384 handle: the same, for its sections' rows.)
 *   THR_PATH_UNSECTIONED_GENERIC_ROWS  both of the above (block_len 16384: the generic k_correlate).
 * All paths implement the same reference semantics and agree to rounding (tests/test_gpu_*.py).
 */
#define THR_VARIANT_DEFAULT 0
#define THR_VARIANT_PRESHIFT 1
#define THR_VARIANT_FASTDET 2
#define THR_INTERP_PARABOLIC 0
#define THR_INTERP_NONE 1
#define THR_INTERP_GAUSSIAN 2
#define THR_INTERP_COSINE 3
#define THR_PATH_AUTO 0
#define THR_PATH_MULTIPASS 1
#define THR_PATH_UNSECTIONED 2
#define THR_PATH_GENERIC_ROWS 3
#define THR_PATH_UNSECTIONED_GENERIC_ROWS 4
int thr_create_ex(const thr_settings* settings, int variant, int variant_arg, int path, thr_handle** out);
/*
 * The overlap-save plan THR_PATH_AUTO uses for the correlate stage of a long block (host-only, no
 * device involved; exported so that the plan itself can be checked against the reference's
 * `despread` / `get_peak`, soa_estimator.py:97-102,137-143, on a CPU).  Section g transforms samples
 * [start[g], start[g] + 16384) of the frequency-shifted block against the template zero-padded to
 * 16384; lag j of that circular correlation, 0 <= j <= 16384 - template_len, is lag start[g] + j of
 * the block's.  In block coordinates section g searches the window lags [win_lo[g], win_hi[g]) and
 * sums (stddev threshold term) the lags [sum_lo[g], sum_hi[g]): the window ranges tile the unique
 * window of soa_estimator.calculate_window (soa_estimator.py:20-39) and the sum ranges tile
 * [0, corr_len), each exactly once, ascending in g, and every searched lag has both neighbours
 * inside its section.  Arrays hold THR_MAX_SECTIONS ints.  *n_sections = 0 when the block is not
 * sectioned (block_len < 16384, or more than THR_MAX_SECTIONS would be needed: templates longer
 * than 9361 samples at block_len 65536).
 * block_len 16384 (ABI 8): sections of 4096 samples, [start[g], start[g] + 4096), starts on multiples
 * of 8 samples, at most FOUR (a fifth would cost more than the 16384-point transform pair: *n_sections
 * = 0 then, as for templates longer than 4081 samples); only the unique window is tiled (win_lo /
 * win_hi; sum_lo = sum_hi = start: a stddev threshold term keeps the unsectioned kernel), and a
 * searched lag has both neighbours inside its section unless it is the first or the last kept lag
 * of the block, where the reference takes no neighbours either (soa_estimator.py:163-164).
 * Geometry only: whether an engine handle uses the plan also depends on its block length, template
 * count, thresholds and path (thr_debug_sections).
 */
#define THR_MAX_SECTIONS 8
int thr_plan_sections(int block_len, int history_len, int template_len, int* n_sections, int* start,
                      int* win_lo, int* win_hi, int* sum_lo, int* sum_hi);
/* Replaces fastcard_free() (fastcard.c:119-146). */
void thr_destroy(thr_handle* h);

/*
 * Replaces the body of the hot loop: `Detector.detect` (detect.py:60-78) ==
 * Synchronizer.sync (carrier_sync.py:52-76) + SoaEstimator.soa_estimate
 * (soa_estimator.py:78-92), and fastcard_process() + CorrDetector::detect()
 * (fastcard.c:177-189, corr_detector.cpp:177-197), for `n_blocks` blocks at once.
 *
 * Host-buffer form: `samples` is n_blocks * block_len samples in `format`
 * (u8: 2 bytes/sample; c64: 8 bytes/sample) in host memory; `block_idx` (may be
 * NULL -> 0,1,2,...) is copied into the records; `out` receives
 * n_blocks * n_templates records, ordered [block][template].  Synchronous.
 */
int thr_detect(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
               size_t n_blocks, thr_record* out);

/*
 * thr_detect() with the sub-bin carrier offsets GIVEN -- replaces `Synchronizer.sync` with a replaced
 * `interpolator` (carrier_sync.py:52-76: `offset = self.interpolator(fft_mag, peak_idx)`, then the
 * shift by -(peak_idx + offset)): what the reference's InterpolationDetector
 * (thrifty/experimental/detect_carrier_interpol.py:17-40) does with any function of
 * thrifty/experimental/carrier_interpolators.py:17-81.  The caller has run the carrier stage once
 * (thr_detect + thr_debug_fft give it the peak bin and |FFT#1|), evaluated its interpolator on the
 * host and hands the results back: block i is shifted by -(its carrier bin + carrier_offset[i])
 * instead of by the Dirichlet fit; everything downstream (FFT#2, correlation, SoA) is unchanged and
 * the record carries carrier_offset[i].  Entries of blocks without a carrier are ignored.  The
 * reference's IndexError rule belongs to ITS interpolator (carrier_sync.py:187) and is not applied:
 * THR_FLAG_INDEX_ERROR is never set here.  One batch (n_blocks <= max_batch), host pointers,
 * synchronous, the default detector only.  A slow path by design (two passes over the block and a
 * host round trip per batch) -- for analysis scripts, not for throughput.
 */
int thr_detect_offsets(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
                       size_t n_blocks, const double* carrier_offset, thr_record* out);

/*
 * .card form (ingest on the device).  `text` holds .card records
 * "<timestamp> <block_idx> <base64 of 2*block_len bytes>" (block_data.py:120-131;
 * fastcard_cli.c:187-192); `payload_off[i]` is the byte offset of the i-th block's base64
 * payload inside `text` (the host only splits lines).  The text crosses PCIe once and is
 * decoded on the device (replaces base64.b64decode + np.fromstring, block_data.py:129, and
 * fastcard/lib/base64.c), then processed exactly like thr_detect().  Host pointers,
 * synchronous.  Invalid base64 -> THR_ERR_ARG.
 */
/*
 * Host-side framing for thr_detect_card(): find the records of a .card text.  Skips what the
 * reference's card_reader skips (block_data.py:101-131: '#' comments, blank lines, fastcard's
 * banner lines "Using Volk machine:" / "linux;"); every other line must be
 * "<timestamp> <block_idx> <payload>" with a payload of exactly 4*ceil(2*block_len/3)
 * characters.  Frames at most `max_records` whole lines starting at `text`; a last line without
 * a newline counts only if `at_eof`.  Outputs per record: timestamp (correctly rounded, like
 * Python's float()), block index, payload offset relative to `text`; `*consumed` = bytes
 * framed (the next call starts there).  No device involved.  Malformed line -> THR_ERR_ARG -- from
 * the call that STARTS at it: a call that meets it after whole records returns those first (the
 * reference's per-line loop had processed them before it raised).
 */
int thr_frame_card(const char* text, size_t text_len, int block_len, int at_eof, size_t max_records,
                   double* timestamps, int64_t* block_idx, int64_t* payload_off, size_t* n_records,
                   size_t* consumed);

/*
 * Page-lock caller memory that the host entry points read from -- typically the mmap of the input
 * file (`thrifty detect rx.card` / `--raw`): the chunk copies of thr_detect / thr_detect_card /
 * thr_detect_stream and their thr_submit_* forms then run as DMA straight from the page cache and
 * RETURN AT ONCE, instead of occupying the calling thread for the length of each copy while the
 * HIP runtime stages pageable memory (same PCIe rate either way -- measured 56 GB/s on MI355X --
 * but the host thread frames the next chunk and formats the previous one's output meanwhile).
 * Purely an optimisation: everything works on unregistered memory, and a failed registration
 * (THR_ERR_DEVICE: locked-memory limit, exotic mapping) leaves nothing behind.  The range is
 * widened to whole pages; read-only and private file mappings are fine.  No handle: the lock
 * belongs to the process.  Unregister (same pointer) before unmapping; inputs of open tickets
 * must have been consumed (thr_inputs_consumed / thr_collect).
 */
int thr_host_register(const void* p, size_t bytes);
int thr_host_unregister(const void* p);
/*
 * The streaming form, for inputs of any size: declare [p, p + bytes) -- the mmap of the input file
 * -- as the window this handle's host entry points will read FRONT TO BACK.  A worker thread of
 * the library page-locks it 128 MiB at a time, at most 1 GiB ahead of the chunk copies, and
 * unlocks what they have left behind, so the locking costs neither the caller's time (it runs
 * beside the copies: measured 10-40 ms per GiB, about what the staging of pageable memory costs the
 * calling thread) nor more locked memory than that, however large the file.  Source ranges
 * outside the window, behind its read position or far ahead of it are copied as ordinary pageable
 * memory, as is everything once a lock is refused -- results never depend on the window.
 * (NULL, 0) closes it; thr_destroy does too.  Not while tickets are open (THR_ERR_STATE).  The
 * mapping must stay valid until the window is closed.
 */
int thr_input_window(thr_handle* h, const void* p, size_t bytes);
/*
 * The same with its two resources stated: `populate_threads` (1 .. 16; 0 = the default, 3) threads
 * map the pages of a segment ahead of the locking worker -- on a node whose CPUs are shared by
 * several ranks a caller gives each rank its share (thrifty_amd.parallel.populate_threads: CPUs /
 * world - 3, at least 1); `segment_bytes` (a power of two >= 64 KiB; 0 = the default, 128 MiB) is
 * the locking granularity -- the tests shrink it to put segment boundaries where they want them.
 */
int thr_input_window_ex(thr_handle* h, const void* p, size_t bytes, int populate_threads, size_t segment_bytes);
/*
 * "The input has been read": no further copy will come out of the window.  Returns at once; what is
 * still page-locked (the segments of the last chunks) is unlocked by the window's own thread in the
 * background -- a few milliseconds of hipHostUnregister that a caller writing its last output line
 * need not wait for.  Source ranges inside the window are copied as pageable memory from here on.
 * The mapping must STILL stay valid until thr_input_window(h, NULL, 0), the next
 * thr_input_window[_ex] on this handle or thr_destroy has returned: those wait for the unlocking.
 * Not while tickets are open (THR_ERR_STATE).
 */
int thr_input_window_release(thr_handle* h);

int thr_detect_card(thr_handle* h, const char* text, size_t text_len, const int64_t* payload_off,
                    const int64_t* block_idx, size_t n_blocks, thr_record* out);

/*
 * Raw-stream framing on the device.  Replaces the overlap bookkeeping of
 * `block_reader` (thrifty/block_data.py:70-98: block = previous block's last
 * `history_len` samples + `block_len - history_len` new ones) and of fastcard's
 * raw_reader_next (fastcard/lib/raw_reader.c:15-46): `stream` is the receiver's
 * interleaved u8 I/Q byte stream and block i is the 2*block_len bytes starting
 * 2*(block_len - history_len)*i bytes into it -- the kernels read overlapping
 * windows in place, nothing is copied or re-framed.  Needs an even
 * block_len - history_len (4-byte aligned block starts), else THR_ERR_ARG.
 * The reference's very first block has an all-zero (0.0, not quantiser-zero)
 * history and therefore no u8 form: callers send that one block through
 * thr_detect(..., THR_IN_C64, ...) (thrifty_amd/block_data.py:RawStream does).
 *
 * thr_detect_stream: host pointer, synchronous; processes the
 *   (n_bytes - 2*block_len) / (2*(block_len - history_len)) + 1 whole blocks the
 *   stream holds (0 if shorter than one block), numbers them first_block_idx,
 *   first_block_idx+1, ..., writes [n_blocks][n_templates] records and stores
 *   the block count in *n_blocks_out.  `out_capacity` is in blocks.
 * thr_detect_stream_device: device pointers, asynchronous like
 *   thr_detect_device; n_blocks <= max_batch; d_stream must hold
 *   (n_blocks-1)*2*(block_len-history_len) + 2*block_len bytes.
 */
int thr_detect_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                      thr_record* out, size_t out_capacity, size_t* n_blocks_out);
int thr_detect_stream_device(thr_handle* h, const uint8_t* d_stream, const int64_t* d_block_idx,
                             size_t n_blocks, thr_record* d_out);

/*
 * Device-resident form (what bench.py times): `d_samples`, `d_block_idx`
 * (may be NULL) and `d_out` are device pointers on the handle's device.  The
 * work is enqueued on the handle's stream and NOT synchronised; call
 * thr_sync() (or synchronise the stream you passed to thr_set_stream()).
 */
int thr_detect_device(thr_handle* h, const void* d_samples, int format,
                      const int64_t* d_block_idx, size_t n_blocks, thr_record* d_out);
int thr_sync(thr_handle* h);
/* Use an externally owned hipStream_t (e.g. a torch side stream); NULL restores the handle's own
 * (non-blocking) stream -- NOTE that torch's DEFAULT stream has the handle value 0, i.e. NULL:
 * passing it here selects the engine's own stream, which does not synchronise with the legacy
 * default stream; to run ON the default stream call thr_set_stream_default() instead
 * (thrifty_amd._native.Engine.set_stream(0) does). */
int thr_set_stream(thr_handle* h, void* hip_stream);
/* Run on the device's legacy default stream (the one torch's default stream and every other
 * blocking stream are ordered with) -- what a caller holding `torch.cuda.current_stream()` on the
 * default stream wants: fills and copies queued there are then ordered before the engine's
 * kernels without any explicit synchronisation. */
int thr_set_stream_default(thr_handle* h);

/*
 * Asynchronous host boundary (precedent: the producer/consumer split fastdet
 * planned and never built, fastdet/fastdet.cpp:179-180).  thr_submit*() is thr_detect*() for ONE
 * batch (n_blocks <= max_batch) cut in two: it stages the inputs, enqueues the copies and kernels,
 * and returns a ticket while the device works; thr_collect(ticket) waits for that batch and fills
 * the `out` array given at submit time (caller-owned; it must stay valid and untouched until the
 * collect returns -- the records travel through pinned staging owned by the library).  The INPUT
 * arrays (samples / text / stream, offsets, indices) must stay valid until
 * thr_inputs_consumed(ticket) -- which waits for the batch's host-to-device copies only, not for
 * its kernels -- or thr_collect(ticket) has returned.  Up to THR_MAX_IN_FLIGHT tickets may be open per
 * handle (a further submit fails with THR_ERR_STATE); tickets may be collected in any order, each
 * exactly once; batches execute in submission order.  Ticket 0 is what an empty batch gets and
 * needs no collect.  The synchronous host entry points refuse to run while tickets are open.
 * thr_poll: *done = 1 if thr_collect(ticket) would not block.
 * Same single-threaded-per-handle rule as everything else.
 */
#define THR_MAX_IN_FLIGHT 3
int thr_submit(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
               size_t n_blocks, thr_record* out, uint64_t* ticket);
int thr_submit_card(thr_handle* h, const char* text, size_t text_len, const int64_t* payload_off,
                    const int64_t* block_idx, size_t n_blocks, thr_record* out, uint64_t* ticket);
int thr_submit_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                      thr_record* out, size_t out_capacity, size_t* n_blocks_out, uint64_t* ticket);
int thr_collect(thr_handle* h, uint64_t ticket);
/*
 * How thr_collect (and the drains of the synchronous host entry points, and thr_run_*) wait for a
 * batch: 0 (default) = hipEventSynchronize -- the HIP runtime polls, the calling thread is busy for
 * the length of the wait: lowest latency, one CPU per handle; 1 = ask (hipEventQuery) and nap 40 us
 * in between -- for hosts whose CPUs are shared by several ranks (eight ranks on a 16-CPU cgroup:
 * eight polling threads would take half of it from the ranks' text and page-locking threads).  With
 * batches in flight ahead of the one waited for, the naps cost no throughput.
 */
int thr_set_wait_mode(thr_handle* h, int sleeping);
int thr_inputs_consumed(thr_handle* h, uint64_t ticket);
int thr_poll(thr_handle* h, uint64_t ticket, int* done);

/*
 * `.toad` text for a batch of DETECTED records -- replaces `DetectionResult.serialize`
 * (thrifty/toads_data.py:47-61) called once per detection by detector_cli (detect.py:217-219).
 * One line per record, each ending in '\n':
 *   [rxid ][txid ]t block soa sample offset energy noise cbin coffset cenergy cnoise
 * t as %.6f, soa = new_len * block_idx + corr_sample + corr_offset as %.8f, the integers as
 * str(int), every other float as Python's repr() of the value widened to a double (what
 * '{}'.format gives for a float and for an np.float32).  with_rxid / with_txid: prepend the
 * ids like serialize() does for values that are not None (txid = the record's template_id:
 * multi-template detection); carrier_offset_f32: 1 = round the carrier offset to float32 first
 * (PreshiftDetector's np.float32 offset), 2 = print it as an integer (its interpolator `none`
 * returns the int 0); a record flagged THR_FLAG_INT_OFFSET prints it as an integer whatever the mode.
 * `out_capacity` must be >= n * THR_TOAD_LINE_MAX.
 * Host only, no device, handle-free.
 */
#define THR_TOAD_LINE_MAX 384
int thr_format_toad(const thr_record* recs, const double* timestamps, size_t n, int64_t new_len,
                    int with_rxid, int64_t rxid, int with_txid, int carrier_offset_f32, char* out,
                    size_t out_capacity, size_t* out_len);

/*
 * The whole `thrifty detect <file> -o <toad>` loop in ONE call -- replaces the reference's per-block
 * loop (detect.py:197-223: card_reader / block_reader -> Detector.detect -> `if detected:
 * print(result.serialize())`; block_data.py:70-131) for an input that lies in memory (the mmap of
 * the file, or a rank's shard of it).  Built on the entry points above and nothing else: the calling
 * thread frames batches (thr_frame_card), keeps up to THR_MAX_IN_FLIGHT of them submitted
 * (thr_submit_card / thr_submit_stream) and collects them in order; a thread of the library keeps the
 * DETECTED records of each collected batch, formats them (thr_format_toad) and write()s the text to
 * `out_fd`, and / or appends the records -- each with its timestamp's bits in `reserved` -- to
 * `rec_out` (what the ranks of a sharded run exchange).  No interpreter in the loop, no hand-over of
 * a lock between the two threads.
 *
 * Semantics are those of the batched Python loop (thrifty_amd/detect.py), i.e. the reference's:
 *  - output order = input order; only blocks whose correlation verdict is positive produce a line;
 *  - a block flagged THR_FLAG_INDEX_ERROR (the reference raises IndexError there,
 *    carrier_sync.py:187) ends the run: the detections before it are written, the call returns
 *    THR_ERR_INDEX and `stats` names the block;
 *  - a malformed .card line (THR_ERR_ARG from thr_frame_card) or an invalid base64 payload ends it
 *    likewise, after everything before the offending batch has been written.
 * thr_run_card: `text` = .card text (whole lines; the last one may lack its newline).
 * thr_run_stream: `stream` = raw interleaved u8 I/Q whose first 2*block_len bytes are block
 *   `first_block_idx` (overlap framing as thr_detect_stream; the caller sends the reference's
 *   zero-history lead-in blocks through thr_detect as before and starts here behind them); every
 *   block of a batch is stamped with the wall clock when the batch is framed (`timestamp` NaN) or
 *   with `timestamp`.
 * If the handle has an input window (thr_input_window) around the input, the copies are DMA from
 * the page cache as usual.  The handle must have no open tickets.
 */
typedef struct thr_run_opts {
    uint32_t struct_bytes;        /* sizeof(thr_run_opts) -- guards the layout                     */
    int32_t batch_blocks;         /* blocks per submitted batch; 0 = the handle's max_batch          */
    int32_t out_fd;               /* >= 0: the .toad text is written here; -1: no text               */
    int32_t with_rxid;            /* the arguments of thr_format_toad ...                            */
    int64_t rxid;
    int32_t with_txid;
    int32_t carrier_offset_mode;  /* ... its carrier_offset_f32                                      */
    double timestamp;             /* thr_run_stream only, see above                                  */
    thr_record* rec_out;          /* NULL, or room for rec_capacity detected records                 */
    size_t rec_capacity;
} thr_run_opts;
typedef struct thr_run_stats {
    uint64_t blocks;              /* blocks whose records were handed to the formatter               */
    uint64_t detections;          /* lines written / records appended                                */
    uint64_t batches;
    uint64_t bytes_in;            /* input bytes framed                                              */
    uint64_t text_bytes;          /* bytes written to out_fd                                         */
    uint64_t index_error_at;      /* position (0-based, in this run) of the block that ended the run
                                     with THR_ERR_INDEX; UINT64_MAX otherwise                        */
    int64_t index_error_block;    /* its block index                                                 */
    int32_t index_error_bin;      /* its carrier bin                                                 */
    int32_t reserved_;
    double total_s;               /* where the calling thread's time went: the whole call,           */
    double frame_s, submit_s, wait_s;     /* framing, thr_submit*, waiting in thr_collect            */
    double format_s, write_s;     /* the library thread: thr_format_toad, write()                    */
} thr_run_stats;
int thr_run_card(thr_handle* h, const char* text, size_t text_len, const thr_run_opts* opts,
                 thr_run_stats* stats);
int thr_run_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                   const thr_run_opts* opts, thr_run_stats* stats);
/* The settings a handle was created with (`templates` is NULL: the array is not retained). */
int thr_get_settings(const thr_handle* h, thr_settings* out);

/*
 * K7: keep only records whose THR_FLAG_CORR is set, preserving order -- what
 * detector_cli's `if detected: print(result.serialize())` does (detect.py:217-219).
 * Device pointers (`d_in` and `d_out` must not overlap); `*n_kept` is written on the host
 * after an internal sync.
 */
int thr_compact_device(thr_handle* h, const thr_record* d_in, size_t n_records,
                       thr_record* d_out, size_t* n_kept);

/*
 * Per-kernel timing with HIP events on the handle's stream (bench.py's
 * roofline leg).  `on` = n > 0 brackets each kernel of every n-th thr_detect*()
 * batch with events (n = 1: all; sampling keeps the ~35 us/batch cost of the event
 * packets out of most steps); 0 disables.  thr_profile_read() syncs and returns accumulated
 * milliseconds and launch counts per kernel slot and resets the accumulators.
 * Slots: 0 = carrier (FFT#1 + peak), 1 = fit, 2 = correlate (FFT#2..peak; long blocks: the
 * fused kernel -- a block's R0 sub-transforms and their combination in one workgroup, one
 * launch per sub-batch), 3 = finish (SoA), 4 = long blocks only: the two-kernel form
 * (sub-transforms, then combination, per chunk of work-list slots) that takes the sub-batches
 * with fewer carrier-positive blocks than workgroups; both forms are launched every time and
 * the one the batch does not belong to returns at once.
 */
#define THR_N_KERNEL_SLOTS 5
int thr_profile_enable(thr_handle* h, int on);
int thr_profile_read(thr_handle* h, double ms[THR_N_KERNEL_SLOTS],
                     int64_t launches[THR_N_KERNEL_SLOTS]);
const char* thr_kernel_name(int slot);

/*
 * Test hooks (tests/ only).  Host pointers, synchronous.
 * thr_debug_fft: forward FFT of n_blocks blocks -> complex64 spectra in natural
 *   order (parity of K1+K2 against np.fft.fft, signal_utils.py:21-25).
 * thr_debug_stage: per-block intermediates of Detector.detect(yield_data=True)
 *   (detect.py:75-78): the frequency-shifted spectrum and the correlation
 *   (first corr_len lags) for template `template_id`; either output may be NULL.
 */
/* thr_debug_correlate_geom: which window-row specialisation of the correlate kernel this handle's
 *   launches take (csrc/correlate16k_geom.hpp): *rows_lo / *rows_hi = rows of 1024 lags compiled out
 *   below / above the unique window, or -1, -1 for the generic kernel (a window outside the table,
 *   a stddev threshold term, another block length or variant).  Decided by the same function the
 *   launcher calls.  (A block_len 16384 handle whose plain launches run sectioned -- thr_debug_sections
 *   -- reports the entry its stage-dump launches of k_correlate take.) */
int thr_debug_correlate_geom(thr_handle* h, int* rows_lo, int* rows_hi);
/*
 * thr_get_path_info: which kernels this handle's plain launches take, and why (no device work).
 * A settings change that looks harmless can cost a sixth of the throughput -- a stddev term in
 * corr_thresh, a history of 1100 instead of 4096 samples -- so the choice is reportable:
 * `Detector.engine_path` and one logging.info line at construction come from here.
 *   n_sections / section_len   the correlate stage's overlap-save sections (0, 0: unsectioned)
 *   rows_lo / rows_hi          window-row specialisation of k_correlate / k_correlate_seg (-1: generic)
 *   why_unsectioned            THR_WHY_*: 0 when sectioned, otherwise the FIRST reason that applies
 *   carrier_kernel, correlate_kernel   kernel family names (as in profiles/ and bench.py)
 *   text                       the same as one sentence
 */
#define THR_WHY_SECTIONED 0   /* the correlate stage runs in sections                                  */
#define THR_WHY_PATH 1        /* the handle was created with an unsectioned / multi-pass kernel path   */
#define THR_WHY_VARIANT 2     /* preshift / fastdet variant: ONE fused kernel per block                */
#define THR_WHY_STDDEV 3      /* corr_thresh has a stddev term: its sums run over every kept lag       */
#define THR_WHY_GEOMETRY 4    /* template / history: the unique window needs more sections than pay
                                 (block_len 16384: more than four of 4096 samples) or than fit         */
#define THR_WHY_BLOCK_LEN 5   /* this block length has no sectioned form (whole blocks sit in LDS, or
                                 the generic multi-pass pipeline)                                      */
typedef struct thr_path_info {
    int32_t n_sections, section_len;
    int32_t rows_lo, rows_hi;
    int32_t why_unsectioned;
    int32_t n_templates;
    char carrier_kernel[48];
    char correlate_kernel[48];
    char text[256];
} thr_path_info;
int thr_get_path_info(thr_handle* h, thr_path_info* out);

/* thr_debug_sections: the overlap-save sections this handle's plain correlate launches run in:
 *   *n_sections (0: unsectioned) of *section_len samples (16384 for long blocks, 4096 for block_len
 *   16384). */
int thr_debug_sections(thr_handle* h, int* n_sections, int* section_len);
/* thr_debug_window: the input window's state, in bytes from its (page-aligned) start:
 *   out[0] = everything below this offset has been released by the chunk copies (may be unlocked),
 *   out[1], out[2] = the range that is page-locked now, out[3] = the segment size (0: no window). */
int thr_debug_window(thr_handle* h, size_t out[4]);
/* thr_debug_window_times: seconds the window's threads have spent since it was opened -- out[0]
 *   populating page tables (summed over the populator threads), out[1] in hipHostRegister, out[2] in
 *   hipHostUnregister, out[3] the CALLER waiting in front of a copy for its segments to be locked;
 *   out[4] = number of such waits, out[5] = chunk copies that went out as pageable memory instead. */
int thr_debug_window_times(thr_handle* h, double out[6]);
/* thr_debug_pipe_times: seconds the calling thread has spent inside the chunks of the host entry
 *   points since the last call (reads and resets): out[0] growing staging buffers, out[1] the input's
 *   host-to-device copy calls, out[2] block indices / offsets, out[3] kernel launches, out[4] the
 *   records' device-to-host calls; out[5] = chunks; out[6] event record / wait between the copy and
 *   the main stream, out[7] filling the index array (sample chunks); out[8 + k] = the longest single
 *   occurrence of phase k. */
int thr_debug_pipe_times(thr_handle* h, double out[16]);
int thr_debug_fft(thr_handle* h, const void* samples, int format, size_t n_blocks,
                  float* spectra_out /* [n_blocks][block_len][2] */);
int thr_debug_stage(thr_handle* h, const void* samples, int format, size_t n_blocks,
                    int template_id, float* shifted_fft_out /* [n][block_len][2] */,
                    float* corr_out /* [n][block_len][2] */);
/* thr_debug_stage_offsets: the same dump with the sub-bin carrier offsets GIVEN (NULL: the engine's
 *   own fit), as thr_detect_offsets takes them -- what a replaced `soa_estimate.interpolate`
 *   (reference experimental/detect_xcorr_interpol.py:36-62) looks at when `sync.interpolator` has
 *   been replaced too. */
int thr_debug_stage_offsets(thr_handle* h, const void* samples, int format, size_t n_blocks,
                            int template_id, const double* carrier_offset /* [n] or NULL */,
                            float* shifted_fft_out, float* corr_out);

/*
 * identify: merge-side post-processing of detections (thrifty/identify.py:26-181) --
 * transmitter classification from the carrier bin, the duplicate filter, output order.
 * Columns in, columns out (host pointers, synchronous, handle-free; `device_id` picks
 * the GPU that sorts).  `map` / `n_map`: the frequency map of load_freqmap
 * (identify.py:184-215) flattened in its iteration order -- inclusive ranges on
 * carrier_bin + carrier_offset, the LAST matching entry wins, -1 if none
 * (classify_transmitters, identify.py:109-121); n_map == 0 selects the automatic mode
 * (detect_transmitter_windows + np.digitize, identify.py:26-106).
 * Outputs: txid_out[n]; keep_out[n] = the mask of identify_duplicates (identify.py:140-172,
 * including its quirks: the neighbour test ignores rxid/txid and wraps around the sorted
 * ends); kept_order_out[0 .. *n_kept_out) = indices of the kept detections sorted by
 * timestamp, stable (filter_duplicates, identify.py:175-181).
 */
typedef struct thr_freq_range {
    int32_t rxid;
    int32_t txid;
    double lo;
    double hi;
} thr_freq_range;

int thr_identify(int device_id, size_t n, const int32_t* rxid, const int32_t* block,
                 const double* timestamp, const int32_t* carrier_bin, const double* carrier_offset,
                 const double* energy, const thr_freq_range* map, size_t n_map, int32_t* txid_out,
                 uint8_t* keep_out, int64_t* kept_order_out, size_t* n_kept_out);

#ifdef __cplusplus
}
#endif
#endif /* THRIFTY_HIP_H */
