// Helpers shared by the LDS-resident kernels (both workgroup geometries).
#pragma once
#include <hip/hip_runtime.h>

// (Measured, not adopted: `__attribute__((amdgpu_waves_per_eu(2, 2)))` on the LDS-resident
// kernels -- it tells the compiler that only 2 waves per SIMD can ever be resident, which
// they are, one 512-thread workgroup per CU -- raised k_correlate from 119 to 238 VGPRs but
// did not make it faster: 0.55 ms per 8192 blocks either way; pruned carrier kernel -2 %,
// long-block correlate +13 %.)

namespace thr {

// Thread id the optimiser cannot see through: stops LICM from hoisting every
// per-thread LDS address / window predicate out of the persistent block loop
// (that cost ~220 SGPR + ~70 VGPR spills).
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// A wave-uniform pointer the optimiser cannot see through, back in SGPRs (two v_mov + two
// v_readfirstlane): address arithmetic on it stays scalar AND stays inside the persistent block
// loop -- hoisted, a kernel's row bases (15 x 2 SGPRs for one twiddle table) are spilled to VGPR
// lanes and read back with v_readlane every iteration.
template <class T>
__device__ __forceinline__ const T* opaque_uniform(const T* p) {
    unsigned lo = unsigned(reinterpret_cast<unsigned long long>(p));
    unsigned hi = unsigned(reinterpret_cast<unsigned long long>(p) >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return reinterpret_cast<const T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// sin / cos for |x| <= 0.2 rad: Taylor to x^7 / x^6 (truncation < 4e-13 / 2e-11, far below the
// float rounding of the result) -- 10 VALU ops instead of sincosf's ~60 with its large-argument
// reduction compiled in.  (The 73 shift-phasor factors of a 16384-sample block: |x| <= 0.172.)
__device__ __forceinline__ void sincos_small(float x, float* s, float* c) {
    const float x2 = x * x;
    float ps = fmaf(x2, -1.0f / 5040.0f, 1.0f / 120.0f);
    ps = fmaf(x2, ps, -1.0f / 6.0f);
    *s = fmaf(x * x2, ps, x);
    float pc = fmaf(x2, -1.0f / 720.0f, 1.0f / 24.0f);
    pc = fmaf(x2, pc, -0.5f);
    *c = fmaf(x2, pc, 1.0f);
}

// Dev-only cycle timeline (-DTHR_DEV, never set by thrifty_amd/build.py): workgroup 0 records
// s_memtime at phase boundaries of its 4th work item, one row of 16 stamps per wave, into
// cfg.timeline (scripts/timeline.py).  The default build compiles the stamps to nothing.
#ifdef THR_DEV
#define THR_STAMP(slot)                                                                   \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (tl_on) {                                                                      \
            const unsigned long long _t = __builtin_amdgcn_s_memtime();                   \
            if ((threadIdx.x & 63) == 0) cfg.timeline[(threadIdx.x >> 6) * 16 + (slot)] = _t; \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#else
#define THR_STAMP(slot) do { } while (0)
#endif

// ---------------------------------------------------------------- reductions
// Wave-level reductions on the VALU's DPP path (row_shr 1/2/4/8 inside each 16-lane
// row, then row_bcast15 / row_bcast31 across rows): ~10 VALU ops instead of six
// LDS-crossbar ds_bpermute round trips.  The result is valid in lane 63 and is
// broadcast from there with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned identity, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114,
              DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// sum over the four lanes of a quad (every lane gets it): quad_perm [1,0,3,2], then [2,3,0,1]
__device__ __forceinline__ float quad_sum(float v) {
    v += __uint_as_float(dpp_u32<0xB1, 0xf>(0u, __float_as_uint(v)));
    v += __uint_as_float(dpp_u32<0x4E, 0xf>(0u, __float_as_uint(v)));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#define THR_STEP(CTRL, MASK) v += __uint_as_float(dpp_u32<CTRL, MASK>(0u, __float_as_uint(v)))
    THR_STEP(DPP_ROW_SHR1, 0xf);
    THR_STEP(DPP_ROW_SHR2, 0xf);
    THR_STEP(DPP_ROW_SHR4, 0xf);
    THR_STEP(DPP_ROW_SHR8, 0xf);
    THR_STEP(DPP_ROW_BCAST15, 0xa);
    THR_STEP(DPP_ROW_BCAST31, 0xc);
#undef THR_STEP
    return __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v) {
#define THR_STEP(CTRL, MASK)                                                       \
    {                                                                              \
        const unsigned lo = dpp_u32<CTRL, MASK>(0u, (unsigned)v);                  \
        const unsigned hi = dpp_u32<CTRL, MASK>(0u, (unsigned)(v >> 32));          \
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;          \
        v = w > v ? w : v;                                                         \
    }
    THR_STEP(DPP_ROW_SHR1, 0xf)
    THR_STEP(DPP_ROW_SHR2, 0xf)
    THR_STEP(DPP_ROW_SHR4, 0xf)
    THR_STEP(DPP_ROW_SHR8, 0xf)
    THR_STEP(DPP_ROW_BCAST15, 0xa)
    THR_STEP(DPP_ROW_BCAST31, 0xc)
#undef THR_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// max of two keys whose high word is the bit pattern of a float that is not negative (a power,
// +inf and NaN patterns included): read as IEEE doubles such keys are non-negative and finite
// (exponent field <= 0x7FC) and order exactly like the integers, so ONE v_max_f64 does what
// v_cmp_gt_u64 + two v_cndmask do (fp64 denormals are never flushed in HIP kernels).
__device__ __forceinline__ unsigned long long max_power_key(unsigned long long a,
                                                            unsigned long long b) {
    unsigned long long r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned long long wave_max_power_key(unsigned long long v) {
#define THR_STEP(CTRL, MASK)                                                       \
    {                                                                              \
        const unsigned lo = dpp_u32<CTRL, MASK>(0u, (unsigned)v);                  \
        const unsigned hi = dpp_u32<CTRL, MASK>(0u, (unsigned)(v >> 32));          \
        v = max_power_key(((unsigned long long)hi << 32) | lo, v);                 \
    }
    THR_STEP(DPP_ROW_SHR1, 0xf)
    THR_STEP(DPP_ROW_SHR2, 0xf)
    THR_STEP(DPP_ROW_SHR4, 0xf)
    THR_STEP(DPP_ROW_SHR8, 0xf)
    THR_STEP(DPP_ROW_BCAST15, 0xa)
    THR_STEP(DPP_ROW_BCAST31, 0xc)
#undef THR_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// Combined block reduction: NS float sums (returned as double) + one u64 max,
// ONE barrier.  `scratch` is double-buffered by `parity` (flip it on every call)
// so a fast wave's next reduction cannot overwrite slots a slow wave still reads.
template <int NW>
constexpr int red_slot_bytes() { return NW * 32; }  // per parity: NW waves x (3 doubles + u64)
// One float sum over the workgroup (returned as double), ONE barrier; same scratch layout and
// parity rule as block_reduce.
template <int NW>
__device__ __forceinline__ double block_sum(float s, unsigned char* scratch, int parity) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* sd = reinterpret_cast<double*>(scratch + parity * red_slot_bytes<NW>());
    s = wave_sum(s);
    if (lane == 0) sd[wv * 3] = (double)s;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += sd[w * 3];
    return t;
}

// POWER_KEY: the keys are power keys (max_power_key above): v_max_f64 instead of 64-bit compares.
template <int NS, int NW, bool POWER_KEY = false>
__device__ __forceinline__ void block_reduce(float (&s)[NS], double (&out)[NS],
                                             unsigned long long& m, unsigned char* scratch,
                                             int parity) {
    static_assert(NS <= 3, "scratch layout holds 3 sums");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* sd = reinterpret_cast<double*>(scratch + parity * red_slot_bytes<NW>());
    unsigned long long* su = reinterpret_cast<unsigned long long*>(sd + 3 * NW);
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = wave_sum(s[i]);
    m = POWER_KEY ? wave_max_power_key(m) : wave_max(m);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) sd[wv * 3 + i] = (double)s[i];
        su[wv] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sd[w * 3 + i];
        out[i] = t;
    }
    unsigned long long t = su[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = POWER_KEY ? max_power_key(su[w], t) : (su[w] > t ? su[w] : t);
    m = t;
}

// Wave-wide float maximum / unsigned minimum on the DPP path (result in every lane).
__device__ __forceinline__ float wave_max_f32(float v) {
    // (identity -1: the callers' values are powers, or -1 for "no candidate")
#define THR_STEP(CTRL, MASK) \
    v = __builtin_fmaxf(v, __uint_as_float(dpp_u32<CTRL, MASK>(0xBF800000u, __float_as_uint(v))))
    THR_STEP(DPP_ROW_SHR1, 0xf);
    THR_STEP(DPP_ROW_SHR2, 0xf);
    THR_STEP(DPP_ROW_SHR4, 0xf);
    THR_STEP(DPP_ROW_SHR8, 0xf);
    THR_STEP(DPP_ROW_BCAST15, 0xa);
    THR_STEP(DPP_ROW_BCAST31, 0xc);
#undef THR_STEP
    return __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define THR_STEP(CTRL, MASK) \
    { const unsigned o = dpp_u32<CTRL, MASK>(0xFFFFFFFFu, v); v = o < v ? o : v; }
    THR_STEP(DPP_ROW_SHR1, 0xf)
    THR_STEP(DPP_ROW_SHR2, 0xf)
    THR_STEP(DPP_ROW_SHR4, 0xf)
    THR_STEP(DPP_ROW_SHR8, 0xf)
    THR_STEP(DPP_ROW_BCAST15, 0xa)
    THR_STEP(DPP_ROW_BCAST31, 0xc)
#undef THR_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Block maximum of one power key per WAVE (already uniform in the wave): ONE barrier, same
// scratch layout and parity rule as block_reduce.
template <int NW>
__device__ __forceinline__ void block_reduce_wave_keys(unsigned long long& m, unsigned char* scratch,
                                                       int parity) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* su =
        reinterpret_cast<unsigned long long*>(scratch + parity * red_slot_bytes<NW>()) + 3 * NW;
    if (lane == 0) su[wv] = m;
    __syncthreads();
    unsigned long long t = su[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = max_power_key(su[w], t);
    m = t;
}

// The same with no sums: one max of power keys (max_power_key), ONE barrier (same scratch layout
// and parity rule).
template <int NW>
__device__ __forceinline__ void block_reduce_max(unsigned long long& m, unsigned char* scratch,
                                                 int parity) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long* su =
        reinterpret_cast<unsigned long long*>(scratch + parity * red_slot_bytes<NW>()) + 3 * NW;
    m = wave_max_power_key(m);
    if (lane == 0) su[wv] = m;
    __syncthreads();
    unsigned long long t = su[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = max_power_key(su[w], t);
    m = t;
}

}  // namespace thr
