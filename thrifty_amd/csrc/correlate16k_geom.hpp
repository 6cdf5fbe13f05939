// The window-row specialisations of k_correlate (correlate16k.hpp: RLO, RHI) as a small closed table.
//
// The peak search visits the 16384 lags of a transform pair in 16 rows of 1024.  A variant <RLO, RHI>
// compiles out the rows that lie entirely outside the unique window [w_lo, w_hi) of
// soa_estimator.calculate_window (soa_estimator.py:20-39) -- rows < RLO and > 15 - RHI, not even
// their share of pass C remains -- and drops the window test from the rows entirely inside it; the
// two boundary rows keep the test.  Which pair applies follows from the history and template
// lengths alone: w_lo = (H - W + 1) / 2 lies in row RLO, w_hi = N - W + 1 - ceil((H - W + 1) / 2) in
// row 15 - RHI.  The table covers RLO = 0 .. 2 and RHI = 0 .. 4 -- windows that start in the first
// 3072 lags and end in the last 5120: BASELINE's (1, 2) (history 4096, 1023-sample template), the
// example detector.cfg's (0, 4), a template as long as its history (0, 0..4), every history up to
// about 6000 samples beyond the template -- so that a user's geometry gets the kernel quality of the
// benchmarked ones; anything outside it (and every stddev-threshold or dump launch) runs the
// generic form with the test in all 16 rows.  Sections of long blocks (SEG, detect_seg.hip) start
// their owned lags at 0 or 1: RLO = 0 only.
//
// One translation unit per RLO (detect16k_geom{0,1,2}.hip) so that the 60 instantiations compile in
// parallel with everything else.
#pragma once
#include <algorithm>

#include "correlate16k.hpp"

namespace thr {

constexpr int kGeomLoMax = 2, kGeomHiMax = 4;

template <int FMT, bool MULTI, int LO, bool SEG>
correlate_fn geom_row_pick(int hi) {
    switch (hi) {
        case 0: return &k_correlate<FMT, false, MULTI, false, LO, 0, SEG>;
        case 1: return &k_correlate<FMT, false, MULTI, false, LO, 1, SEG>;
        case 2: return &k_correlate<FMT, false, MULTI, false, LO, 2, SEG>;
        case 3: return &k_correlate<FMT, false, MULTI, false, LO, 3, SEG>;
        case 4: return &k_correlate<FMT, false, MULTI, false, LO, 4, SEG>;
    }
    return nullptr;
}

template <int LO, bool SEG>
correlate_fn geom_row(int fmt, bool multi, int hi) {
    if (fmt == THR_IN_U8)
        return multi ? geom_row_pick<THR_IN_U8, true, LO, SEG>(hi) : geom_row_pick<THR_IN_U8, false, LO, SEG>(hi);
    return multi ? geom_row_pick<THR_IN_C64, true, LO, SEG>(hi) : geom_row_pick<THR_IN_C64, false, LO, SEG>(hi);
}

// > 64 KiB of dynamic LDS is opted into per kernel: every variant of one row
template <int LO, bool SEG>
hipError_t prepare_geom_row() {
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int m = 0; m < 2; ++m)
            for (int hi = 0; hi <= kGeomHiMax; ++hi) {
                const hipError_t e =
                    hipFuncSetAttribute(reinterpret_cast<const void*>(geom_row<LO, SEG>(fmt, m != 0, hi)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)k16::LDS_BYTES);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

// The pair of the table that applies to EVERY one of the `n` windows [w_lo[g], w_hi[g]) -- one
// window for a 16384-sample block, one per section for a long one -- preferring the most rows
// compiled out; false if none does (the generic form runs).
inline bool pick_row_geom(const int* w_lo, const int* w_hi, int n, int lo_max, int* lo_out, int* hi_out) {
    for (int skip = lo_max + kGeomHiMax; skip >= 0; --skip)
        for (int lo = std::min(skip, lo_max); lo >= 0; --lo) {
            const int hi = skip - lo;
            if (hi > kGeomHiMax) break;
            bool all = true;
            for (int g = 0; g < n && all; ++g) all = row_geom_applies(lo, hi, w_lo[g], w_hi[g]);
            if (all) {
                *lo_out = lo;
                *hi_out = hi;
                return true;
            }
        }
    return false;
}

// detect16k_geom{0,1,2}.hip
correlate_fn geom_variant_lo0(int fmt, bool multi, int hi);
correlate_fn geom_variant_lo1(int fmt, bool multi, int hi);
correlate_fn geom_variant_lo2(int fmt, bool multi, int hi);
hipError_t prepare_geom_lo0();
hipError_t prepare_geom_lo1();
hipError_t prepare_geom_lo2();

}  // namespace thr
