// Shared device/host structures of the detection pipeline.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/thrifty_hip.h"

namespace thr {

constexpr int kMaxTemplates = 8;
constexpr int kMaxSections = 8;   // (a work item carries its section in 3 bits)

// Per-launch constants (passed by value as a kernel argument).
struct DevCfg {
    int block_len;
    int history_len;
    int n_templates;
    int carrier_len;   // Dirichlet kernel width W (carrier_sync.py:150-196)
    int win_lo;        // first FFT index of the carrier window (carrier_detect.py:17-58)
    int win_count;     // number of bins in the (wrapping, inclusive) window, <= N
    int corr_lo;       // unique-lag window [corr_lo, corr_hi) (soa_estimator.py:20-39)
    int corr_hi;
    int corr_len;      // block_len - template_len + 1
    int car_want_std;  // carrier threshold has a stddev term
    int cor_want_std;  // correlation threshold has a stddev term
    // threshold coefficients (constant, snr, stddev) exactly as given (Python floats = double).
    // The correlation verdict is float64 arithmetic in the reference (soa_estimator.py:127-134);
    // the carrier verdict combines them with float32 statistics, where NumPy >= 2 (NEP 50)
    // first rounds the Python float to float32 -- the kernels narrow at that point, not here.
    double car_thr[3];
    double cor_thr[3];
    float tmpl_energy[kMaxTemplates];  // sum t^2 (soa_estimator.py:65)
    unsigned long long blk_stride;  // bytes from one block's first sample to the next one's: N * sample
                                    // size for packed blocks, 2 (N - H) for raw-stream framing
    int car_prune;     // pruned FFT#1 (16384 path): 0 off, 1 window+margin inside bins [0,128),
                       // 2 any window of <= 122 bins (samples pre-shifted by win_lo - 3)
    const void* gtw;   // [16][1024] W_16384^(k1 q) for k_correlate's passes 1 and B, followed by the same values in the pairs pass B reads (handle.hip; kernels compiled for the LDS-table form ignore it)
    int variant;       // 0 reference Detector, 1 PreshiftDetector, 2 fastdet-compatible (power-domain verdicts)
    int interp;        // PreshiftDetector: carrier interpolator (THR_INTERP_*: 0 parabolic, 1 none, 2 gaussian, 3 cosine)
    // Overlap-save sections of the correlate stage (0 = none): block_len > 16384 (detect_seg.hip):
    // section g covers samples [seg_start[g], seg_start[g] + 16384) of a block; block_len 16384 with a
    // short template (detect16k_sec.hip): [seg_start[g], seg_start[g] + 4096).  In SECTION
    // coordinates (lag - seg_start[g]): it owns the window lags [seg_lo[g], seg_hi[g]) and, for the
    // stddev term, sums the lags [seg_sum_lo[g], seg_sum_hi[g]); the owned ranges tile the block's
    // [corr_lo, corr_hi) resp. [0, corr_len) exactly once (plan_sections, handle.hip).
    int no_row_geom;   // THR_PATH_GENERIC_ROWS: the correlate launches take the generic kernel (cross-checks)
    int n_seg;
    int seg_start[kMaxSections];
    int seg_lo[kMaxSections], seg_hi[kMaxSections];
    int seg_sum_lo[kMaxSections], seg_sum_hi[kMaxSections];
#ifdef THR_DEV
    unsigned long long* timeline;  // -DTHR_DEV: [8 waves][16] s_memtime stamps of one k_correlate item
#endif
};

// K_A -> K_fit
struct CarStats {
    float sum_mag2;  // sum |X|^2
    float sum_mag;   // sum |X| (only if car_want_std)
    float peak_mag;  // |X[peak]|
    int peak_idx;    // reference's peak_idx (may equal N, carrier_detect.py:151)
    float nb[7];     // |X[peak-3 .. peak+3]| (indices wrapped)
    int pad;
};

// K_B -> K_finish, one per (block, template)
struct CorrStats {
    float pm2;       // |corr[pk]|^2
    float m2[3];     // |corr[pk-1..pk+1]|^2
    float sum_x2;    // sum |X^|^2 of the shifted spectrum (template 0's slot only)
    float sum_mag;   // sum |corr| over [0, corr_len)   (only if cor_want_std)
    float sum_mag2;  // sum |corr|^2 over [0, corr_len) (only if cor_want_std)
    int pk;          // windowed first-max lag
};

// K_fit -> K_B: the frequency-shift phasor exp(2 pi i s (n/N - 1/2)), factored
struct ShiftParams {
    float2 rpow[16];  // exp(2 pi i s j / R1), j = sub-sequence index of pass 1
    float2 r0pow[4];  // long blocks only: exp(2 pi i s j / R0), j = leading sub-sequence index
    float2 c0;        // exp(-pi i s)
    int si_mod;       // round(s) mod N
    float sf_over_n;  // (s - round(s)) / N
    int bank;         // preshift variant (multi-pass pipeline): pre-shifted template index
    int pad_;
    float2 segc0[kMaxSections];  // sectioned correlate stage: exp(2 pi i s (seg_start[g] / N - 1/2))
};

// detect16k_preshift.hip -- PreshiftDetector variant (one fused kernel per block at 16384)
hipError_t prepare_preshift_16k();
hipError_t launch_preshift_16k(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                               const float2* tables, const float2* bank, int num,
                               const long long* block_idx, CorrStats* corr_stats,
                               thr_record* records, int grid, hipStream_t stream);
hipError_t launch_fit_preshift(int n_blocks, const DevCfg& cfg, int num, const CarStats* stats,
                               const long long* block_idx, ShiftParams* shifts,
                               thr_record* records, hipStream_t stream);

// detect16k.hip
hipError_t prepare_16k();
size_t lds_bytes_16k();
hipError_t launch_carrier_16k(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                              const float2* tables, const float2* twn, CarStats* stats,
                              float2* dump_fft, int grid, hipStream_t stream);
hipError_t launch_fit(int n_blocks, const DevCfg& cfg, const CarStats* stats,
                      const long long* block_idx, ShiftParams* shifts, int* work_list,
                      int* work_count, thr_record* records, CorrStats* corr_stats_x2,
                      hipStream_t stream,    // corr_stats_x2: where to park sum |X|^2 (or null)
                      const double* forced_offset = nullptr);   // [n_blocks] sub-bin offsets instead of the fit
hipError_t launch_correlate_16k(int fmt, const void* samples, const DevCfg& cfg,
                                const float2* tables, const float2* twn, const float4* tspec,
                                const ShiftParams* shifts, const int* work_list,
                                const int* work_count, CorrStats* corr_stats,
                                float2* dump_xhat, float2* dump_corr, int dump_template, int grid,
                                hipStream_t stream);
// the window-row specialisation (correlate16k_geom.hpp) the correlate launch of this configuration
// takes; false: the generic kernel
bool correlate_geom_16k(const DevCfg& cfg, int* lo, int* hi);
bool correlate_geom_seg(const DevCfg& cfg, int* lo, int* hi);
// seg_stats (or null): the sectioned correlate stage's [record][section] results, merged here
hipError_t launch_finish(int n_records, const DevCfg& cfg, const CorrStats* corr_stats,
                         thr_record* records, int* work_count, hipStream_t stream,
                         const CorrStats* seg_stats = nullptr);
int compact_tiles(int n_records);   // ints of tile scratch launch_compact needs
hipError_t launch_compact(const thr_record* in, int n, thr_record* out, int* n_out,
                          int* tile_scratch, hipStream_t stream);

// detect_seg.hip (block_len = 2 or 4 x 16384: the correlate stage as overlap-save sections of the
// 16384-point kernel; tspec16k = the templates zero-padded to 16384, k_correlate's layout)
hipError_t prepare_seg();
hipError_t launch_correlate_seg(int fmt, const void* samples, const DevCfg& cfg, const float2* tables,
                                const float2* twn, const float4* tspec16k, const ShiftParams* shifts,
                                const int* work_list, const int* work_count, CorrStats* seg_stats,
                                int grid, hipStream_t stream);

// detect16k_sec.hip (block_len 16384, short templates, no stddev term: the correlate stage as up
// to four overlap-save sections of 4096 samples, a 128-thread workgroup each; seg_stats:
// [block][template][section]; tspec4k = the templates zero-padded to 4096, the short-block kernels'
// layout; ctab_pair / park: several templates only)
size_t lds_bytes_4k();
size_t park_bytes_4k(int grid);
hipError_t launch_correlate_4k(int fmt, const void* samples, const DevCfg& cfg, const float2* tables,
                               const float2* twn, const float4* tspec4k, const ShiftParams* shifts,
                               const int* work_list, const int* work_count, CorrStats* seg_stats,
                               const float4* ctab_pair, float4* park, int grid, hipStream_t stream);

// detect_long.hip (block_len = 2 or 4 x 16384: R0 LDS-resident sub-transforms per block)
bool long_supported(int block_len);
hipError_t prepare_long(int block_len);
hipError_t launch_carrier_long(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                               const float2* tables, const float2* twn, float* win_pow,
                               float* partial, CarStats* stats, float2* dump_fft, int grid,
                               hipStream_t stream);
// correlate stage.  fused: ONE launch over the whole work list; the workgroup that runs a block's
// R0 sub-transforms combines them too (dsub: one row of T x block_len per workgroup); it leaves
// batches with fewer carrier-positive blocks than `grid` alone.  !fused: one chunk of work-list
// slots [base, base + cap): sub-transforms -> dsub, then launch_combine_long; both return at once
// when the work count is >= fused_grid (the fused launch has done the batch).
hipError_t launch_correlate_long(bool fused, int fmt, const void* samples, const DevCfg& cfg,
                                 const float2* tables, const float2* twn, const float4* tspec,
                                 const ShiftParams* shifts, const int* work_list,
                                 const int* work_count, float2* dsub, float4* xhat_scratch,
                                 float2* dump_xhat, CorrStats* corr_stats, float2* dump_corr,
                                 int dump_template, int grid, int base, int cap, int fused_grid,
                                 hipStream_t stream);
hipError_t launch_combine_long(const DevCfg& cfg, const float2* twn, const int* work_list,
                               const int* work_count, const float2* dsub, CorrStats* corr_stats,
                               float2* dump_corr, int dump_template, int base, int cap,
                               int fused_grid, hipStream_t stream);
int long_chunk_blocks(int block_len, int n_templates);

// detect_small.hip (block_len = 1024, 2048, 4096, 8192: 16 / R1 blocks per workgroup in LDS)
bool small_supported(int block_len);
hipError_t prepare_small(int block_len);
// (dump_* non-null: the stage-dump mode, natural order, [block][block_len]; the stddev sums follow
// cfg.car_want_std / cfg.cor_want_std)
hipError_t launch_carrier_small(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                const float2* tables, const float2* gtw, CarStats* stats, float2* dump_fft,
                                int n_cu, hipStream_t stream);
hipError_t launch_correlate_small(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                  const float2* tables, const float2* gtw, const float2* twn,
                                  const float4* tspec, const ShiftParams* shifts, const int* work_list,
                                  const int* work_count, CorrStats* corr_stats, float2* dump_xhat,
                                  float2* dump_corr, int dump_template, int n_cu, hipStream_t stream);

// card_ingest.hip (.card base64 payloads -> u8 IQ on the device)
hipError_t launch_b64_decode(const unsigned char* d_text, const long long* d_payload_off, int n_lines,
                             int out_bytes, unsigned char* d_out, int* d_bad, hipStream_t stream);

// generic.hip (any power-of-two block length; multi-pass through HBM)
size_t generic_scratch_bytes(int n, int n_blocks);
hipError_t generic_carrier(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                           const float2* twn, float2* scratch, CarStats* stats, float2** spectrum,
                           hipStream_t stream);
hipError_t generic_correlate(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                             const float2* twn, const float2* tspec_nat,
                             const ShiftParams* shifts, const thr_record* records, float2* scratch,
                             CorrStats* corr_stats, int dump_template, float2** keep_xhat,
                             float2** keep_corr, hipStream_t stream);
// preshift variant: `spectrum` = generic_carrier's FFT#1 (inside scratch), rolled by
// shifts[b].si_mod and multiplied by bank[shifts[b].bank] (natural order, conj, /N)
hipError_t generic_preshift_correlate(int n_blocks, const DevCfg& cfg, const float2* twn,
                                      const float2* bank_nat, const ShiftParams* shifts,
                                      const thr_record* records, float2* scratch,
                                      const float2* spectrum, CorrStats* corr_stats,
                                      float2* dump_rolled, float2** keep_corr, hipStream_t stream);
// (dump_rolled / keep_corr: the stage dumps of yield_data -- np.roll(FFT#1, round(shift)), natural
// order, [block][n], written for carrier-positive blocks; where the correlation was left)

}  // namespace thr
