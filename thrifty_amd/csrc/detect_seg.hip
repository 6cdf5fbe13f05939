// Correlate stage of long blocks (block_len NL = 2 or 4 x 16384; BASELINE configs[2]) as overlap-save
// sections of the LDS-resident 16384-point transform pair (k_correlate<..., SEG>, correlate16k.hpp).
//
// The reference keeps ifft(X^ conj(T^))[:corr_len] with the template zero-padded to NL
// (soa_estimator.py:97-102): corr[l] = sum_{n < W} y[l + n] t[n] for l < corr_len = NL - W + 1, a
// linear correlation of the frequency-shifted block y (carrier_sync.py:222-238) with the W-sample
// template -- no kept lag wraps.  Section g transforms y[seg_start[g] .. + 16384) against the
// template zero-padded to 16384; its lags 0 .. 16384 - W are exact lags seg_start[g] + (0 ..
// 16384 - W) of the block.  With W = 4094 five sections at stride 12288 cover all 61443 lags of a
// 65536-sample block; each is one work item of the same kernel a 16384-sample block runs, and what
// leaves the CU per section is one 32-byte CorrStats -- against the 0.75 MiB of parked rows per
// block of the decimated form (detect_long.hip), which stays for stage dumps and for templates too
// long to section (handle.hip: plan_sections).  k_finish (detect16k_carrier.hip) keeps, per block and
// template, the first section holding the maximum (soa_estimator.py:137-143: np.argmax takes the
// lowest lag, and the owned lag ranges ascend with the section index).
#include <hip/hip_runtime.h>

#include "correlate16k_geom.hpp"

namespace thr {

using namespace k16;

namespace {
template <int FMT, bool STD>
correlate_fn seg_pick(bool multi) {
    return multi ? &k_correlate<FMT, STD, true, false, -1, -1, true>
                 : &k_correlate<FMT, STD, false, false, -1, -1, true>;
}
correlate_fn seg_variant(int fmt, bool want_std, bool multi) {
    if (fmt == THR_IN_U8) return want_std ? seg_pick<THR_IN_U8, true>(multi) : seg_pick<THR_IN_U8, false>(multi);
    return want_std ? seg_pick<THR_IN_C64, true>(multi) : seg_pick<THR_IN_C64, false>(multi);
}
}  // namespace

hipError_t prepare_seg() {
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int m = 0; m < 2; ++m)
            for (int st = 0; st < 2; ++st) {
                hipError_t e = hipFuncSetAttribute(
                    reinterpret_cast<const void*>(seg_variant(fmt, st != 0, m != 0)),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                if (e != hipSuccess) return e;
            }
    // the window-row specialisations of the sections (correlate16k_geom.hpp): owned lags start at
    // 0 or 1, so RLO = 0; RHI = 0 .. 4 by the template length (BASELINE's 4094 samples: 3)
    return prepare_geom_row<0, true>();
}

// which specialisation launch_correlate_seg takes: one pair must fit the window of EVERY section
// (the launch is one kernel over all (block, section) items)
bool correlate_geom_seg(const DevCfg& cfg, int* lo, int* hi) {
    if (cfg.cor_want_std != 0 || cfg.n_seg <= 0 || cfg.no_row_geom) return false;
    return pick_row_geom(cfg.seg_lo, cfg.seg_hi, cfg.n_seg, 0, lo, hi);
}

// seg_stats: [block of the sub-batch][template][section]
hipError_t launch_correlate_seg(int fmt, const void* samples, const DevCfg& cfg, const float2* tables,
                                const float2* twn, const float4* tspec16k, const ShiftParams* shifts,
                                const int* work_list, const int* work_count, CorrStats* seg_stats,
                                int grid, hipStream_t stream) {
    const bool multi = cfg.n_templates > 1;
    correlate_fn fn = seg_variant(fmt, cfg.cor_want_std != 0, multi);
    int lo = -1, hi = -1;
    if (correlate_geom_seg(cfg, &lo, &hi)) fn = geom_row<0, true>(fmt, multi, hi);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec16k), shifts, work_list, work_count, seg_stats,
                       static_cast<cpx*>(nullptr), static_cast<cpx*>(nullptr), 0);
    return hipGetLastError();
}

}  // namespace thr
