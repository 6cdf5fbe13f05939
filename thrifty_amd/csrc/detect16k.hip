// LDS-resident fused detection kernels for block_len = 16384 on gfx950.
//
// One 512-thread workgroup owns one IQ block at a time; the block's 16384
// complex samples live in LDS (136 KiB padded) between the three register
// passes of a 16 x 32 x 32 decimation-in-frequency FFT.  The spectrum leaves
// the last pass in digit-reversed order *in registers*, which is exactly the
// order the mirrored decimation-in-time inverse consumes, so the template
// product and the first inverse pass need no LDS round trip and no reordering.
//
//   k_carrier   : u8/c64 -> FFT#1 -> |X|^2 sum, windowed first-max, 7-bin
//                 neighbourhood                (signal_utils.py:21-25,
//                 carrier_detect.py:99-154)
//   k_fit       : noise/threshold verdict + Dirichlet LSQ fit + shift phasors
//                 (carrier_detect.py:61-115, carrier_sync.py:150-196)
//   k_correlate : shift -> FFT#2 -> x conj(T) -> IFFT -> |.|^2 windowed
//                 argmax -> noise/threshold -> log-parabola -> record
//                 (carrier_sync.py:222-238, soa_estimator.py:78-170)
// The carrier kernels, k_fit and k_finish live in detect16k_carrier.hip, k_correlate itself in
// correlate16k.hpp (detect_seg.hip instantiates it for the sections of long blocks); this file
// holds its block_len = 16384 instantiations and their launcher.
#include <hip/hip_runtime.h>

#include "correlate16k_geom.hpp"

namespace thr {

using namespace k16;

size_t lds_bytes_16k() { return LDS_BYTES; }

namespace {
#ifdef THR_DEV_MINIMAL  // compile-time experiments only: one variant each, fast rebuilds
correlate_fn correlate_variant(int, bool, bool, bool) {
    return &k_correlate<THR_IN_U8, false, false, false>;
}
#else
template <int FMT, bool STD, bool MULTI>
correlate_fn pick_correlate(bool dump) {
    return dump ? &k_correlate<FMT, STD, MULTI, true> : &k_correlate<FMT, STD, MULTI, false>;
}
template <int FMT>
correlate_fn pick_correlate2(bool want_std, bool multi, bool dump) {
    if (want_std)
        return multi ? pick_correlate<FMT, true, true>(dump) : pick_correlate<FMT, true, false>(dump);
    return multi ? pick_correlate<FMT, false, true>(dump) : pick_correlate<FMT, false, false>(dump);
}
correlate_fn correlate_variant(int fmt, bool want_std, bool multi, bool dump) {
    return fmt == THR_IN_U8 ? pick_correlate2<THR_IN_U8>(want_std, multi, dump)
                            : pick_correlate2<THR_IN_C64>(want_std, multi, dump);
}
#endif

// the window-row specialisation for this geometry (no stddev term, no dumps), or nullptr
correlate_fn geom_variant(int fmt, int lo, int hi, bool multi) {
#ifdef THR_DEV_MINIMAL
    (void)fmt;
    (void)lo;
    (void)hi;
    (void)multi;
    return nullptr;
#else
    switch (lo) {
        case 0: return geom_variant_lo0(fmt, multi, hi);
        case 1: return geom_variant_lo1(fmt, multi, hi);
        case 2: return geom_variant_lo2(fmt, multi, hi);
    }
    return nullptr;
#endif
}
}  // namespace

hipError_t prepare_16k_carrier();   // detect16k_carrier.hip

hipError_t prepare_16k() {
    // > 64 KiB of dynamic LDS must be opted into, per device and per kernel variant
    hipError_t e = prepare_16k_carrier();
    if (e != hipSuccess) return e;
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int st = 0; st < 2; ++st)
            for (int d = 0; d < 2; ++d)
                for (int m = 0; m < 2; ++m) {
                    e = hipFuncSetAttribute(
                        reinterpret_cast<const void*>(correlate_variant(fmt, st, m, d)),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                    if (e != hipSuccess) return e;
                }
#ifndef THR_DEV_MINIMAL
    if ((e = prepare_geom_lo0()) != hipSuccess) return e;
    if ((e = prepare_geom_lo1()) != hipSuccess) return e;
    if ((e = prepare_geom_lo2()) != hipSuccess) return e;
#endif
    return hipSuccess;
}

// which window-row specialisation launch_correlate_16k takes for this configuration (false: the
// generic kernel) -- the launcher and thr_debug_correlate_geom share this one decision
bool correlate_geom_16k(const DevCfg& cfg, int* lo, int* hi) {
    if (cfg.cor_want_std != 0 || cfg.no_row_geom) return false;
    return pick_row_geom(&cfg.corr_lo, &cfg.corr_hi, 1, kGeomLoMax, lo, hi);
}

hipError_t launch_correlate_16k(int fmt, const void* samples, const DevCfg& cfg,
                                const float2* tables, const float2* twn, const float4* tspec,
                                const ShiftParams* shifts, const int* work_list,
                                const int* work_count, CorrStats* corr_stats,
                                float2* dump_xhat, float2* dump_corr, int dump_template, int grid,
                                hipStream_t stream) {
    const bool dump = dump_xhat != nullptr || dump_corr != nullptr;
    correlate_fn fn = correlate_variant(fmt, cfg.cor_want_std != 0, cfg.n_templates > 1, dump);
    int lo = -1, hi = -1;
    if (!dump && correlate_geom_16k(cfg, &lo, &hi)) {
        correlate_fn g = geom_variant(fmt, lo, hi, cfg.n_templates > 1);
        if (g != nullptr) fn = g;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list, work_count,
                       corr_stats, reinterpret_cast<cpx*>(dump_xhat),
                       reinterpret_cast<cpx*>(dump_corr), dump_template);
    return hipGetLastError();
}

}  // namespace thr
