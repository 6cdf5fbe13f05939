// k_correlate<..., RLO = 2, RHI = 0 .. 4>: one row of the table of window-row specialisations
// (correlate16k_geom.hpp) -- block_len 16384, u8 / complex64 input, one / several templates.
#include <hip/hip_runtime.h>

#include "correlate16k_geom.hpp"

namespace thr {

correlate_fn geom_variant_lo2(int fmt, bool multi, int hi) { return geom_row<2, false>(fmt, multi, hi); }
hipError_t prepare_geom_lo2() { return prepare_geom_row<2, false>(); }

}  // namespace thr
