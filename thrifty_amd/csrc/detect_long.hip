// Long blocks: block_len NL = R0 * 16384 (R0 = 2, 4; BASELINE config C3 is 65536).
//
// A block no longer fits the CU's 160 KiB LDS, so the transform is split by one
// decimation-in-frequency stage in front of the LDS-resident 16384-point kernels
// (passes_w8.hpp):
//     n = n0*M + m,  k = k0 + R0*q        (M = 16384)
//     X[k0 + R0 q] = FFT_M{ W_NL^(m k0) * sum_n0 x[n0 M + m] W_R0^(n0 k0) }[q]
// i.e. R0 independent 16384-point sub-transforms per block whose inputs are formed on
// the fly by the sample loader (R0 coalesced loads per element, a radix-R0 butterfly,
// one twiddle) -- the sub-transform itself, the template product and the mirrored
// inverse run exactly as for N = 16384, with the digit-reversed spectrum staying in
// registers.  What crosses sub-transforms goes through small global arrays:
//   k_carrier_sub  -> |X|^2 of the window bins (+-3) and partial sums -> k_select -> k_fit
//   k_correlate_sub-> d_k0[m] = IFFT_M{...} per (block, template, k0)  -> k_combine:
//                     corr[n0 M + m] = sum_k0 W_R0^(-n0 k0) conj(W_NL^(m k0)) d_k0[m]
// Reference lines as in detect16k.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;  // N == M == 16384 here; NL = R0 * N is the block length

namespace {

constexpr int M = N;

// z * exp(-i*pi/2 * q), q in 0..3 (wave-uniform q)
__device__ __forceinline__ cpx rot_quarter_neg(cpx z, int q) {
    switch (q & 3) {
        case 1: return cpx{z.y, -z.x};
        case 2: return -z;
        case 3: return cpx{-z.y, z.x};
        default: return z;
    }
}

// Sample source of sub-transform k0: the radix-R0 combination of the R0 samples that
// alias onto m.  g[n0] (LDS, wave-uniform) = [shift phasor step r0^n0] * W_R0^(n0 k0);
// GEN = false means g is a pure quarter-turn (carrier stage): adds only.
template <int FMT, int R0, bool GEN>
struct RawLong {
    const void* blk;
    const float2* g;  // [R0]
    int t, k0;
    // u8: the quantiser's affine map (v - 127.4)/128 is applied AFTER the radix-R0 combination
    // (it is linear): y = sum g[n0] (sc u + of (1+i)) = sc * sum g[n0] u + of (1+i) sum g[n0]
    // -- one packed fma per output instead of one fma per byte.
    cpx kadd = cpx{0.f, 0.f};
    // u8, HELD: the block's R0 x 16 raw words stay in registers across the R0 sub-transforms
    // of the block (fetch() once per block, 16 R0 VGPRs) instead of being re-read from L2 / the
    // Infinity Cache for each of them.
    static constexpr bool HELD = FMT == THR_IN_U8 && GEN;
    unsigned w[HELD ? R0 : 1][HELD ? R1 : 1];
    __device__ __forceinline__ void fetch() {
        if constexpr (HELD) {
#pragma unroll
            for (int n0 = 0; n0 < R0; ++n0)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1)
                    w[n0][n1] = reinterpret_cast<const unsigned*>(blk)[size_t(n0) * (M / 2) +
                                                                       size_t(n1) * (S1 / 2) + t];
        }
    }
    __device__ __forceinline__ void prepare() {
        if constexpr (FMT == THR_IN_U8) {
            constexpr float of = -127.4f / 128.0f;
            cpx gs = cpx{1.f, 0.f};
#pragma unroll
            for (int n0 = 1; n0 < R0; ++n0) {
                if constexpr (GEN) gs += cpx{g[n0].x, g[n0].y};
                else gs += rot_quarter_neg(cpx{1.f, 0.f}, (n0 * k0 * (4 / R0)) & 3);
            }
            kadd = cpx{of * (gs.x - gs.y), of * (gs.x + gs.y)};
        }
    }
    __device__ __forceinline__ void pair(int n0, int n1, cpx& a, cpx& b) const {
        const size_t idx = size_t(n0) * (M / 2) + size_t(n1) * (S1 / 2) + t;
        if constexpr (FMT == THR_IN_U8) {
            unsigned v;
            if constexpr (HELD) v = w[n0][n1];
            else v = reinterpret_cast<const unsigned*>(blk)[idx];
            a = cpx{float(v & 0xffu), float((v >> 8) & 0xffu)};      // raw bytes: see prepare()
            b = cpx{float((v >> 16) & 0xffu), float(v >> 24)};
        } else {
            const f4 v = reinterpret_cast<const f4*>(blk)[idx];
            a = cpx{v.x, v.y};
            b = cpx{v.z, v.w};
        }
    }
    __device__ __forceinline__ void get(int n1, cpx& a, cpx& b) const {
        pair(0, n1, a, b);
#pragma unroll
        for (int n0 = 1; n0 < R0; ++n0) {
            cpx x, y;
            pair(n0, n1, x, y);
            if constexpr (GEN) {
                const cpx wg = cpx{g[n0].x, g[n0].y};
                a += cmul(x, wg);
                b += cmul(y, wg);
            } else {
                const int q = (n0 * k0 * (4 / R0)) & 3;  // W_R0^(n0 k0) as quarter turns
                a += rot_quarter_neg(x, q);
                b += rot_quarter_neg(y, q);
            }
        }
        if constexpr (FMT == THR_IN_U8) {
            constexpr float sc = 1.0f / 128.0f;
            a = __builtin_elementwise_fma(a, cpx{sc, sc}, kadd);
            b = __builtin_elementwise_fma(b, cpx{sc, sc}, kadd);
        }
        // complex64 input: R0 x 16 float4 loads would all be hoisted (256 VGPRs) -- fence per n1
        if constexpr (FMT == THR_IN_C64) __builtin_amdgcn_sched_barrier(0);
    }
};

// Per-item uniform factors into LDS scratch: rp[n1] = rpow[n1] * W_NL^(1024 n1 k0) and
// g[n0] (see RawLong); `sp` == nullptr for the carrier stage (no frequency shift).
template <int R0>
__device__ __forceinline__ void item_factors(float2* rp, float2* g, const ShiftParams* sp,
                                             const cpx* __restrict__ twn, int k0, int nl_mask) {
    const int i = threadIdx.x;
    if (i < 16) {
        const cpx u = twn[(1024 * i * k0) & nl_mask];
        const cpx r = sp ? cmul(cpx{sp->rpow[i].x, sp->rpow[i].y}, u) : u;
        rp[i] = float2{r.x, r.y};
    } else if (i >= 64 && i < 64 + R0) {
        const int n0 = i - 64;
        const cpx w = rot_quarter_neg(cpx{1.f, 0.f}, n0 * k0 * (4 / R0));
        const cpx r = sp ? cmul(cpx{sp->r0pow[n0].x, sp->r0pow[n0].y}, w) : w;
        g[n0] = float2{r.x, r.y};
    }
}

// =========================================================================
// carrier stage of one sub-transform
// =========================================================================
template <int FMT, int R0, bool WANT_STD, bool DUMP>
__global__ __launch_bounds__(NT) void k_carrier_sub(const void* __restrict__ samples, int n_blocks,
                                                    DevCfg cfg, const cpx* __restrict__ tables,
                                                    const cpx* __restrict__ twn,
                                                    float* __restrict__ win_pow,   // [b][win_w]
                                                    float* __restrict__ partial,   // [b][R0][2]
                                                    cpx* __restrict__ dump_fft) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float2* sc_rp = reinterpret_cast<float2*>(sc_red + 2 * red_slot_bytes<NT / 64>());  // [16]
    float2* sc_g = sc_rp + 16;                                                            // [R0]

    load_tables(lds, tables);
    __syncthreads();
    const int NL = R0 * M, nl_mask = NL - 1;
    const size_t blk_bytes = cfg.blk_stride;
    const int win_w = min(cfg.win_count + 6, NL);
    const int win_base = cfg.win_lo - 3;
    int parity = 0;

    // With enough blocks a workgroup runs the R0 sub-transforms of ONE block back to back:
    // each of them reads all of the block's samples, so k0 >= 1 finds them in L2.
    const bool per_block = n_blocks >= int(gridDim.x);
    const int n_iter = per_block ? ((n_blocks - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x)) * R0
                                 : (n_blocks * R0 - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    for (int it = 0; it < n_iter; ++it) {
        const int item = per_block ? (int(blockIdx.x) + (it / R0) * int(gridDim.x)) * R0 + it % R0
                                   : int(blockIdx.x) + it * int(gridDim.x);
        const int b = item / R0, k0 = item % R0;
        const int t = opaque_tid();
        __syncthreads();  // scratch factors of the previous item are no longer read
        item_factors<R0>(sc_rp, sc_g, nullptr, twn, k0, nl_mask);
        cpx p[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) p[e] = twn[((2 * t + e) * k0) & nl_mask];  // W_NL^(m' k0)
        __syncthreads();
        // (complex64 input takes the branch-free multiply form: the quarter-turn switch plus
        // R0 x 16 hoisted float4 loads spills ~150 VGPRs)
        RawLong<FMT, R0, FMT == THR_IN_C64> raw{
            static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes, sc_g, t, k0};
        raw.prepare();
        fwd_pass1<true>(lds, raw, sc_rp, p[0], p[1]);
        __syncthreads();
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx v[R3];
        fwd_pass3(lds, v);

        const int kbase = (t >> 5) + 16 * (t & 31);
        float sums[2] = {0.f, 0.f};
        float* wp = win_pow + size_t(b) * win_w;
        static_for<R3>([&](auto K) {
            constexpr int k3 = decltype(K)::value;
            const float pw = cnorm(v[brev(k3, R3)]);
            sums[0] += pw;
            if constexpr (WANT_STD) sums[1] += __builtin_amdgcn_sqrtf(pw);
            const int k = k0 + R0 * (kbase + 512 * k3);
            const unsigned wi = unsigned(k - win_base) & unsigned(nl_mask);
            if (wi < unsigned(win_w)) wp[wi] = pw;
            if constexpr (DUMP) dump_fft[size_t(b) * NL + k] = v[brev(k3, R3)];
        });
        double tot[2] = {0, 0};
        unsigned long long dummy = 0;
        block_reduce<WANT_STD ? 2 : 1, NT / 64>(reinterpret_cast<float(&)[WANT_STD ? 2 : 1]>(sums),
                                                reinterpret_cast<double(&)[WANT_STD ? 2 : 1]>(tot),
                                                dummy, sc_red, parity);
        parity ^= 1;
        if (t == 0) {
            partial[(size_t(b) * R0 + k0) * 2 + 0] = (float)tot[0];
            partial[(size_t(b) * R0 + k0) * 2 + 1] = WANT_STD ? (float)tot[1] : 0.f;
        }
    }
}

// Pruned variant (cf. k_carrier_pruned in detect16k.hip): when the window plus its 3-bin
// fit margin lies inside bins [0, 128 R0) every sub-transform only needs q = k1 + 16 k2 with
// k2 < 8, k3 = 0 -- pass 2 keeps 8 of 32 outputs, pass 3 is a 32-term sum in 128 threads --
// and its share of sum |X|^2 is M * sum |y_k0[m]|^2 (Parseval over the sub-transform's input).
template <int FMT, int R0>
__global__ __launch_bounds__(NT) void k_carrier_sub_pruned(const void* __restrict__ samples,
                                                           int n_blocks, DevCfg cfg,
                                                           const cpx* __restrict__ tables,
                                                           const cpx* __restrict__ twn,
                                                           float* __restrict__ win_pow,
                                                           float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float2* sc_rp = reinterpret_cast<float2*>(sc_red + 2 * red_slot_bytes<NT / 64>());  // [16]
    float2* sc_g = sc_rp + 16;                                                            // [R0]

    load_tables(lds, tables);
    __syncthreads();
    const int NL = R0 * M, nl_mask = NL - 1;
    const size_t blk_bytes = cfg.blk_stride;
    const int win_w = cfg.win_count + 6;
    const int win_base = cfg.win_lo - 3;
    int parity = 0;
    const bool per_block = n_blocks >= int(gridDim.x);
    const int n_iter = per_block ? ((n_blocks - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x)) * R0
                                 : (n_blocks * R0 - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    for (int it = 0; it < n_iter; ++it) {
        const int item = per_block ? (int(blockIdx.x) + (it / R0) * int(gridDim.x)) * R0 + it % R0
                                   : int(blockIdx.x) + it * int(gridDim.x);
        const int b = item / R0, k0 = item % R0;
        const int t = opaque_tid();
        __syncthreads();  // scratch factors of the previous item are no longer read
        item_factors<R0>(sc_rp, sc_g, nullptr, twn, k0, nl_mask);
        cpx p[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) p[e] = twn[((2 * t + e) * k0) & nl_mask];  // W_NL^(m' k0)
        __syncthreads();
        RawLong<FMT, R0, FMT == THR_IN_C64> raw{
            static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes, sc_g, t, k0};
        raw.prepare();
        float sums[1];
        fwd_pass1<true>(lds, raw, sc_rp, p[0], p[1], &sums[0]);
        __syncthreads();
        fwd_pass2<8>(lds);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = t & 31;
        if (k2 < 8) {
            const f4* src = reinterpret_cast<const f4*>(lds + (t >> 5) * ROW + k2 * CHUNK);
            f4 acc = src[0];
#pragma unroll
            for (int j = 1; j < R3 / 2; ++j) acc += src[j];
            const cpx x = cpx{acc.x + acc.z, acc.y + acc.w};
            const int k = k0 + R0 * ((t >> 5) + 16 * k2);
            const unsigned wi = unsigned(k - win_base);
            if (wi < unsigned(win_w)) win_pow[size_t(b) * win_w + wi] = cnorm(x);
        }
        double tot[1];
        unsigned long long dummy = 0;
        block_reduce<1, NT / 64>(sums, tot, dummy, sc_red, parity);
        parity ^= 1;
        if (t == 0) {
            partial[(size_t(b) * R0 + k0) * 2 + 0] = (float)(tot[0] * double(M));  // Parseval
            partial[(size_t(b) * R0 + k0) * 2 + 1] = 0.f;
        }
    }
}

// Decimation-in-time carrier stage: when window + margin lie inside bins [0, 128) the R0
// sub-transforms can run over the DECIMATED sequences x[R0 m + r] instead -- every sample is
// converted once (the radix-R0 pre-stage above re-reads and re-converts the whole block for
// each k0), each pruned transform yields F_r[k], k < 128, and k_select_dit finishes
// X[k] = sum_r W_NL^(r k) F_r[k] for the ~110 window bins.  sum |X|^2 = NL * sum |x|^2.
template <int FMT, int R0>
struct RawDecim;
template <int R0>
struct RawDecim<THR_IN_U8, R0> {
    // Sample R0*m + r is 2 bytes at byte 2*R0*m + 2r, so ONE aligned 4*R0-byte load per n1
    // covers m = 2t, 2t+1 for every r: the block is fetched once (16 x uint4 = 64 VGPRs for
    // R0 = 4) and each of the R0 decimated transforms picks its two samples out of it.
    static_assert(R0 == 2 || R0 == 4, "radix");
    typedef unsigned word_t __attribute__((ext_vector_type(R0)));
    word_t w[R1];
    unsigned q[R1];   // the selected pair: sample 2t in the low half, 2t+1 in the high half
    __device__ __forceinline__ void fetch(const void* __restrict__ blk, int t) {
        const word_t* p = reinterpret_cast<const word_t*>(blk) + t;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) w[n1] = p[n1 * (S1 / 2)];
    }
    __device__ __forceinline__ void select(int r) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            unsigned lo, hi;
            if constexpr (R0 == 4) {
                lo = (r & 2) ? w[n1].y : w[n1].x;
                hi = (r & 2) ? w[n1].w : w[n1].z;
            } else {
                lo = w[n1].x;
                hi = w[n1].y;
            }
            q[n1] = (r & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
    }
    __device__ __forceinline__ void get(int n1, cpx& a, cpx& b) const {
        const unsigned v = q[n1];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;
        a = cpx{fmaf(float(v & 0xffu), sc, of), fmaf(float((v >> 8) & 0xffu), sc, of)};
        b = cpx{fmaf(float((v >> 16) & 0xffu), sc, of), fmaf(float(v >> 24), sc, of)};
    }
};
template <int R0>
struct RawDecim<THR_IN_C64, R0> {
    const cpx* base;
    const cpx* p;
    __device__ __forceinline__ void fetch(const void* __restrict__ blk, int t) {
        base = reinterpret_cast<const cpx*>(blk) + size_t(2 * t) * R0;
    }
    __device__ __forceinline__ void select(int r) { p = base + r; }
    __device__ __forceinline__ void get(int n1, cpx& a, cpx& b) const {
        a = p[size_t(n1) * S1 * R0];
        b = p[size_t(n1) * S1 * R0 + R0];
        __builtin_amdgcn_sched_barrier(0);  // see RawLong: keep the loads from being hoisted en bloc
    }
};

template <int FMT, int R0>
__global__ __launch_bounds__(NT) void k_carrier_dit(const void* __restrict__ samples, int n_blocks,
                                                    DevCfg cfg, const cpx* __restrict__ tables,
                                                    cpx* __restrict__ win_f,      // [b][R0][win_w]
                                                    float* __restrict__ partial)  // [b][R0][2]
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;
    const int win_w = cfg.win_count + 6, win_base = cfg.win_lo - 3;
    int parity = 0;
    cpx tw0[R1], tw1[R1];   // block-invariant pass-1 twiddles of this thread's two columns
    pass1_twiddles(lds, tw0, tw1);
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
      RawDecim<FMT, R0> raw;
      raw.fetch(static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes, opaque_tid());
#pragma nounroll
      for (int r = 0; r < R0; ++r) {
        const int t = opaque_tid();
        raw.select(r);
        float sums[1];
        // (the previous item's pass-3 LDS reads precede its reduction barrier)
        fwd_pass1_pre(lds, raw, tw0, tw1, &sums[0]);
        __syncthreads();
        fwd_pass2<8>(lds);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = t & 31;
        if (k2 < 8) {
            const f4* src = reinterpret_cast<const f4*>(lds + (t >> 5) * ROW + k2 * CHUNK);
            f4 acc = src[0];
#pragma unroll
            for (int j = 1; j < R3 / 2; ++j) acc += src[j];
            const int k = (t >> 5) + 16 * k2;
            const unsigned wi = unsigned(k - win_base);
            if (wi < unsigned(win_w))
                win_f[(size_t(b) * R0 + r) * win_w + wi] = cpx{acc.x + acc.z, acc.y + acc.w};
        }
        double tot[1];
        unsigned long long dummy = 0;
        block_reduce<1, NT / 64>(sums, tot, dummy, sc_red, parity);
        parity ^= 1;
        if (t == 0) {
            partial[(size_t(b) * R0 + r) * 2 + 0] = (float)(tot[0] * double(R0 * M));  // Parseval
            partial[(size_t(b) * R0 + r) * 2 + 1] = 0.f;
        }
      }
    }
}

// combine the decimated transforms on the window bins, then select as k_select does
__global__ __launch_bounds__(128) void k_select_dit(int r0, DevCfg cfg, const cpx* __restrict__ win_f,
                                                    const cpx* __restrict__ twn,
                                                    const float* __restrict__ partial,
                                                    CarStats* __restrict__ stats) {
    __shared__ float pw[128];
    __shared__ unsigned long long sh[2];
    const int b = blockIdx.x, nl = cfg.block_len;
    const int win_w = cfg.win_count + 6, win_base = cfg.win_lo - 3;
    const int i = threadIdx.x;
    unsigned long long best = 0;
    if (i < win_w) {
        const int k = win_base + i;
        cpx x = win_f[size_t(b) * r0 * win_w + i];
        for (int r = 1; r < r0; ++r)
            x += cmul(win_f[(size_t(b) * r0 + r) * win_w + i], twn[(r * k) & (nl - 1)]);
        const float p = cnorm(x);
        pw[i] = p;
        const int wi = i - 3;
        if (wi >= 0 && wi < cfg.win_count)
            best = ((unsigned long long)__float_as_uint(sqrtf(p)) << 32) | (0xFFFFFFFFu - unsigned(wi));
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = sh[1] > sh[0] ? sh[1] : sh[0];
        const int wi = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        CarStats st;
        float s2 = 0.f;
        for (int r = 0; r < r0; ++r) s2 += partial[(size_t(b) * r0 + r) * 2 + 0];
        st.sum_mag2 = s2;
        st.sum_mag = 0.f;
        st.peak_mag = __uint_as_float(unsigned(best >> 32));   // the key carries |X| (ties: see k_carrier_pruned)
        st.peak_idx = wi + cfg.win_lo;   // < 128: the reference's wrap quirk cannot trigger
        for (int d = 0; d < 7; ++d) st.nb[d] = sqrtf(pw[wi + d]);
        st.pad = 0;
        stats[b] = st;
    }
}

// window first-max + neighbourhood + totals, one workgroup per block
__global__ __launch_bounds__(256) void k_select(int r0, DevCfg cfg, const float* __restrict__ win_pow,
                                                const float* __restrict__ partial,
                                                CarStats* __restrict__ stats) {
    __shared__ unsigned long long sh[4];
    const int b = blockIdx.x, nl = cfg.block_len;
    const int win_w = min(cfg.win_count + 6, nl);
    const float* wp = win_pow + size_t(b) * win_w;
    unsigned long long best = 0;
    for (int wi = threadIdx.x; wi < cfg.win_count; wi += blockDim.x) {
        // with a full-length window the +-3 margin cannot be stored: index modulo the array
        const float pw = wp[(wi + 3) % win_w];
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(sqrtf(pw)) << 32) | (0xFFFFFFFFu - unsigned(wi));
        best = key > best ? key : best;
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = sh[i] > best ? sh[i] : best;
        const int wi = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        int peak_idx = wi + cfg.win_lo;
        if (peak_idx > nl) peak_idx -= nl;  // sic (carrier_detect.py:151)
        CarStats st;
        float s2 = 0.f, s1 = 0.f;
        for (int k0 = 0; k0 < r0; ++k0) {
            s2 += partial[(size_t(b) * r0 + k0) * 2 + 0];
            s1 += partial[(size_t(b) * r0 + k0) * 2 + 1];
        }
        st.sum_mag2 = s2;
        st.sum_mag = s1;
        st.peak_mag = __uint_as_float(unsigned(best >> 32));
        st.peak_idx = peak_idx;
        for (int d = 0; d < 7; ++d) st.nb[d] = sqrtf(wp[(wi + d) % win_w]);
        st.pad = 0;
        stats[b] = st;
    }
}

// =========================================================================
// correlation stage of one sub-transform: shift, FFT, x template, inverse -> d_k0[m]
// =========================================================================
template <int FMT, int R0, bool DUMP>
__global__ __launch_bounds__(NT) void k_correlate_sub(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, int slot_base, int slot_cap,
    f4* __restrict__ dsub,           // [slot - slot_base][tpl][k0][M/2] float4
    f4* __restrict__ xhat_scratch, cpx* __restrict__ dump_xhat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float2* sc_rp0 = reinterpret_cast<float2*>(sc_red + 2 * red_slot_bytes<NT / 64>());  // 2 x (16 + R0)

    load_tables(lds, tables);
    __syncthreads();
    const int NL = R0 * M, nl_mask = NL - 1;
    const size_t blk_bytes = cfg.blk_stride;
    cpx ph[2] = {cpx{1.f, 0.f}, cpx{1.f, 0.f}};
    int ph_block = -1;
    RawLong<FMT, R0, true> raw{};
    int raw_block = -1;
    // this launch owns work-list slots [slot_base, slot_base + slot_cap): the correlate stage
    // runs in chunks small enough for the d_k0 exchange to stay in the Infinity Cache
    // (`work_list` already points at slot_base)
    const int n_work = max(0, min(*work_count - slot_base, slot_cap));
    const int T = cfg.n_templates;

    const bool per_block = n_work >= int(gridDim.x);
    const int n_iter = per_block ? ((n_work - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x)) * R0
                                 : (n_work * R0 - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    for (int it = 0; it < n_iter; ++it) {
        const int item = per_block ? (int(blockIdx.x) + (it / R0) * int(gridDim.x)) * R0 + it % R0
                                   : int(blockIdx.x) + it * int(gridDim.x);
        const int slot = item / R0, k0 = item % R0;
        const int b = work_list[slot];
        const int t = opaque_tid();
        const ShiftParams* sp = shifts + b;
        // item factors are double-buffered: this item's writers cannot collide with the previous
        // item's readers, and the buffer written two items ago has barriers in between
        float2* sc_rp = sc_rp0 + (it & 1) * (16 + R0);
        float2* sc_g = sc_rp + 16;
        item_factors<R0>(sc_rp, sc_g, sp, twn, k0, nl_mask);
        // per-thread phasor for m' = 2t, 2t+1: c0 * exp(2 pi i s m'/NL) [per block] * W_NL^(m' k0)
        if (b != ph_block) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = 2 * t + e;
                const int q = int(((long long)sp->si_mod * m) & nl_mask);
                float sn, cs;
                sincosf(6.283185307179586f * (sp->sf_over_n * float(m)), &sn, &cs);
                ph[e] = cmul(cmul(cconj(twn[q]), cpx{cs, sn}), cpx{sp->c0.x, sp->c0.y});
            }
            ph_block = b;
        }
        cpx p[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) p[e] = cmul(ph[e], twn[((2 * t + e) * k0) & nl_mask]);
        __syncthreads();
        raw.blk = static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes;
        raw.g = sc_g;
        raw.t = t;
        raw.k0 = k0;
        if (b != raw_block) {   // block-major work order: once per R0 sub-transforms
            raw.fetch();
            raw_block = b;
        }
        raw.prepare();
        fwd_pass1<true>(lds, raw, sc_rp, p[0], p[1]);
        __syncthreads();
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx xh[R3];
        fwd_pass3(lds, xh);

        const int kbase = (t >> 5) + 16 * (t & 31);
        if constexpr (DUMP) {
            if (dump_xhat != nullptr)
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    dump_xhat[size_t(b) * NL + k0 + R0 * (kbase + 512 * k3)] = xh[brev(k3, R3)];
                });
        }
        // (sum |X^|^2 for the noise estimate == sum |X|^2 of the carrier stage: k_fit parks it)
        // the spectrum stays in registers across templates (64 VGPRs; parking it in an L2
        // scratch row between templates measured slower, as in k_correlate)
        for (int tpl = 0; tpl < T; ++tpl) {
            const int t = opaque_tid();
            const f4* ts = tspec + (size_t(tpl) * R0 + k0) * (M / 2) + t;
            cpx z[R3];
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = ts[j * NT];
                z[brev(2 * j, R3)] = cmul(xh[brev(2 * j, R3)], cpx{q.x, q.y});
                z[brev(2 * j + 1, R3)] = cmul(xh[brev(2 * j + 1, R3)], cpx{q.z, q.w});
            });
            if (tpl > 0) __syncthreads();  // the previous template's pass-C reads of other waves are done
            inv_passA(lds, z);
            __builtin_amdgcn_sched_barrier(0);
            inv_passB(lds);
            __syncthreads();
            cpx c0[R1], c1[R1];
            inv_passC(lds, c0, c1);
            f4* out = dsub + ((size_t(slot) * T + tpl) * R0 + k0) * (M / 2) + t;
            static_for<R1>([&](auto K) {
                constexpr int n1 = decltype(K)::value;
                out[n1 * (S1 / 2)] = f4{c0[brev(n1, R1)].x, c0[brev(n1, R1)].y, c1[brev(n1, R1)].x,
                                        c1[brev(n1, R1)].y};
            });
        }
    }
}

// corr[n0 M + m] = sum_k0 W_R0^(-n0 k0) conj(W_NL^(m k0)) d_k0[m]; windowed first-max, sums
template <int R0>
__device__ __forceinline__ void combine_at(const cpx* __restrict__ d, const cpx* __restrict__ twn,
                                           int m, int nl_mask, cpx (&out)[R0]) {
    cpx u[R0];
    u[0] = d[m];
#pragma unroll
    for (int k0 = 1; k0 < R0; ++k0) u[k0] = cmulc(d[size_t(k0) * M + m], twn[(m * k0) & nl_mask]);
    dft_dif<R0, +1>(u);
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) out[n0] = u[brev(n0, R0)];
}

// HBM / Infinity-Cache-bound byte work (8 R0 M bytes read per (slot, template), ~30 flop per
// 32 bytes): 512 threads, two adjacent lags per thread and step (one 16-byte load per
// sub-transform).  (Measured and dropped: running it on a second stream under the
// sub-transforms of the next chunk -- 1.74 M blocks/s at N = 65536 against 1.71 M serial; with
// the raw block held in registers by k_correlate_sub, 230 VGPRs, the two kernels no longer fit
// on a CU together and serial order wins: 1.78 M against 1.18 M.)
constexpr int CMB_T = 512;
template <int R0>
__global__ __launch_bounds__(CMB_T) void k_combine(DevCfg cfg, const cpx* __restrict__ twn,
                                                   const cpx* __restrict__ dsub,
                                                   const int* __restrict__ work_list,
                                                   const int* __restrict__ work_count,
                                                   int slot_base, CorrStats* __restrict__ corr_stats,
                                                   cpx* __restrict__ dump_corr, int dump_template) {
    __shared__ __attribute__((aligned(16))) unsigned char scratch[2 * (CMB_T / 64) * 32];
    const int T = cfg.n_templates;
    const int slot = blockIdx.x / T, tpl = blockIdx.x % T;   // chunk-local; work_list points at slot_base
    if (slot_base + slot >= *work_count) return;
    const int b = work_list[slot];
    const int NL = R0 * M, nl_mask = NL - 1;
    const cpx* d = dsub + (size_t(slot) * T + tpl) * NL;
    float sums[2] = {0.f, 0.f};
    const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
    // n0-major order would visit lags out of order; ties are resolved through the key instead
    unsigned long long best = 0;
    for (int m = 2 * threadIdx.x; m < M; m += 2 * CMB_T) {
        cpx u0[R0], u1[R0];
        {
            const f4 q = *reinterpret_cast<const f4*>(d + m);
            u0[0] = cpx{q.x, q.y};
            u1[0] = cpx{q.z, q.w};
        }
#pragma unroll
        for (int k0 = 1; k0 < R0; ++k0) {
            const f4 q = *reinterpret_cast<const f4*>(d + size_t(k0) * M + m);
            u0[k0] = cmulc(cpx{q.x, q.y}, twn[(m * k0) & nl_mask]);
            u1[k0] = cmulc(cpx{q.z, q.w}, twn[((m + 1) * k0) & nl_mask]);
        }
        dft_dif<R0, +1>(u0);
        dft_dif<R0, +1>(u1);
#pragma unroll
        for (int n0 = 0; n0 < R0; ++n0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const cpx c = e ? u1[brev(n0, R0)] : u0[brev(n0, R0)];
                const int n = n0 * M + m + e;
                const float pw = cnorm(c);
                if (unsigned(n - cfg.corr_lo) < win_w) {
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(pw) << 32) | (0xFFFFFFFFu - unsigned(n));
                    best = key > best ? key : best;
                }
                if (cfg.cor_want_std && n < cfg.corr_len) {
                    sums[1] += pw;
                    sums[0] += __builtin_amdgcn_sqrtf(pw);
                }
                if (dump_corr != nullptr && tpl == dump_template) dump_corr[size_t(b) * NL + n] = c;
            }
        }
    }
    double tot[2];
    block_reduce<2, CMB_T / 64>(sums, tot, best, scratch, 0);
    if (threadIdx.x == 0) {
        CorrStats* cs = corr_stats + size_t(b) * T + tpl;
        const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        cs->pm2 = __uint_as_float(unsigned(best >> 32));
        cs->pk = pk;
        for (int dd = 0; dd < 3; ++dd) {
            const int n = pk - 1 + dd;
            float v = 0.f;
            if (n >= 0 && n < NL) {
                cpx c[R0];
                combine_at<R0>(d, twn, n % M, nl_mask, c);
                cpx sel = c[0];
#pragma unroll
                for (int n0 = 1; n0 < R0; ++n0) sel = (n / M == n0) ? c[n0] : sel;
                v = cnorm(sel);
            }
            cs->m2[dd] = v;
        }
        cs->sum_mag = (float)tot[0];
        cs->sum_mag2 = (float)tot[1];
    }
}

template <int R0>
hipError_t prepare_r0() {
    const void* fns[] = {
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_U8, R0, false, false>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_U8, R0, true, false>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_U8, R0, false, true>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_U8, R0, true, true>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_C64, R0, false, false>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_C64, R0, true, false>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_C64, R0, false, true>),
        reinterpret_cast<const void*>(&k_carrier_sub<THR_IN_C64, R0, true, true>),
        reinterpret_cast<const void*>(&k_carrier_dit<THR_IN_U8, R0>),
        reinterpret_cast<const void*>(&k_carrier_dit<THR_IN_C64, R0>),
        reinterpret_cast<const void*>(&k_carrier_sub_pruned<THR_IN_U8, R0>),
        reinterpret_cast<const void*>(&k_carrier_sub_pruned<THR_IN_C64, R0>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, true>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, true>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int R0>
hipError_t carrier_r0(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                      const float2* tables, const float2* twn, float* win_pow, float* partial,
                      CarStats* stats, float2* dump_fft, int grid, hipStream_t stream) {
    typedef void (*fn_t)(const void*, int, DevCfg, const cpx*, const cpx*, float*, float*, cpx*);
    const bool st = cfg.car_want_std != 0, dump = dump_fft != nullptr;
    if (cfg.car_prune == 1 && !st && !dump && cfg.win_lo + cfg.win_count + 3 <= 128) {
        typedef void (*dfn_t)(const void*, int, DevCfg, const cpx*, cpx*, float*);
        dfn_t dfn = fmt == THR_IN_U8 ? &k_carrier_dit<THR_IN_U8, R0> : &k_carrier_dit<THR_IN_C64, R0>;
        hipLaunchKernelGGL(dfn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                           reinterpret_cast<const cpx*>(tables), reinterpret_cast<cpx*>(win_pow), partial);
        hipLaunchKernelGGL(k_select_dit, dim3(n_blocks), dim3(128), 0, stream, R0, cfg,
                           reinterpret_cast<const cpx*>(win_pow), reinterpret_cast<const cpx*>(twn),
                           partial, stats);
        return hipGetLastError();
    }
    if (cfg.car_prune == 1 && !st && !dump) {
        typedef void (*pfn_t)(const void*, int, DevCfg, const cpx*, const cpx*, float*, float*);
        pfn_t pfn = fmt == THR_IN_U8 ? &k_carrier_sub_pruned<THR_IN_U8, R0>
                                     : &k_carrier_sub_pruned<THR_IN_C64, R0>;
        hipLaunchKernelGGL(pfn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                           reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                           win_pow, partial);
        hipLaunchKernelGGL(k_select, dim3(n_blocks), dim3(256), 0, stream, R0, cfg, win_pow, partial,
                           stats);
        return hipGetLastError();
    }
    fn_t fn;
    if (fmt == THR_IN_U8)
        fn = st ? (dump ? &k_carrier_sub<THR_IN_U8, R0, true, true> : &k_carrier_sub<THR_IN_U8, R0, true, false>)
                : (dump ? &k_carrier_sub<THR_IN_U8, R0, false, true> : &k_carrier_sub<THR_IN_U8, R0, false, false>);
    else
        fn = st ? (dump ? &k_carrier_sub<THR_IN_C64, R0, true, true> : &k_carrier_sub<THR_IN_C64, R0, true, false>)
                : (dump ? &k_carrier_sub<THR_IN_C64, R0, false, true> : &k_carrier_sub<THR_IN_C64, R0, false, false>);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn), win_pow,
                       partial, reinterpret_cast<cpx*>(dump_fft));
    hipLaunchKernelGGL(k_select, dim3(n_blocks), dim3(256), 0, stream, R0, cfg, win_pow, partial, stats);
    return hipGetLastError();
}

template <int R0>
hipError_t correlate_r0(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                        const float2* tables, const float2* twn, const float4* tspec,
                        const ShiftParams* shifts, const int* work_list, const int* work_count,
                        float2* dsub, float4* xhat_scratch, float2* dump_xhat, int grid, int base,
                        int cap, hipStream_t stream) {
    typedef void (*fn_t)(const void*, DevCfg, const cpx*, const cpx*, const f4*, const ShiftParams*,
                         const int*, const int*, int, int, f4*, f4*, cpx*);
    const bool dump = dump_xhat != nullptr;
    fn_t fn = fmt == THR_IN_U8
                  ? (dump ? &k_correlate_sub<THR_IN_U8, R0, true> : &k_correlate_sub<THR_IN_U8, R0, false>)
                  : (dump ? &k_correlate_sub<THR_IN_C64, R0, true> : &k_correlate_sub<THR_IN_C64, R0, false>);
    // ONE slot chunk [base, base + cap): dsub / partial_x2 hold one chunk (see long_chunk_blocks);
    // the caller loops over the chunks and follows each with launch_combine_long
    hipLaunchKernelGGL(fn, dim3(std::min(grid, cap * R0)), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list + base, work_count, base,
                       cap, reinterpret_cast<f4*>(dsub), reinterpret_cast<f4*>(xhat_scratch),
                       reinterpret_cast<cpx*>(dump_xhat));
    return hipGetLastError();
}

template <int R0>
hipError_t combine_r0(const DevCfg& cfg, const float2* twn, const int* work_list,
                      const int* work_count, const float2* dsub, CorrStats* corr_stats, float2* dump_corr, int dump_template, int base, int cap,
                      hipStream_t stream) {
    hipLaunchKernelGGL(k_combine<R0>, dim3(cap * cfg.n_templates), dim3(CMB_T), 0, stream, cfg,
                       reinterpret_cast<const cpx*>(twn), reinterpret_cast<const cpx*>(dsub),
                       work_list + base, work_count, base, corr_stats,
                       reinterpret_cast<cpx*>(dump_corr), dump_template);
    return hipGetLastError();
}

}  // namespace

bool long_supported(int block_len) { return block_len == 2 * M || block_len == 4 * M; }

hipError_t prepare_long(int block_len) {
    return block_len == 2 * M ? prepare_r0<2>() : prepare_r0<4>();
}

hipError_t launch_carrier_long(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                               const float2* tables, const float2* twn, float* win_pow,
                               float* partial, CarStats* stats, float2* dump_fft, int grid,
                               hipStream_t stream) {
    return cfg.block_len == 2 * M
               ? carrier_r0<2>(fmt, samples, n_blocks, cfg, tables, twn, win_pow, partial, stats,
                               dump_fft, grid, stream)
               : carrier_r0<4>(fmt, samples, n_blocks, cfg, tables, twn, win_pow, partial, stats,
                               dump_fft, grid, stream);
}

hipError_t launch_correlate_long(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                 const float2* tables, const float2* twn, const float4* tspec,
                                 const ShiftParams* shifts, const int* work_list,
                                 const int* work_count, float2* dsub, float4* xhat_scratch,
                                 float2* dump_xhat, int grid, int base, int cap, hipStream_t stream) {
    return cfg.block_len == 2 * M
               ? correlate_r0<2>(fmt, samples, n_blocks, cfg, tables, twn, tspec, shifts, work_list,
                                 work_count, dsub, xhat_scratch, dump_xhat, grid, base, cap, stream)
               : correlate_r0<4>(fmt, samples, n_blocks, cfg, tables, twn, tspec, shifts, work_list,
                                 work_count, dsub, xhat_scratch, dump_xhat, grid, base, cap, stream);
}

hipError_t launch_combine_long(const DevCfg& cfg, const float2* twn, const int* work_list,
                               const int* work_count, const float2* dsub, CorrStats* corr_stats,
                               float2* dump_corr, int dump_template, int base, int cap,
                               hipStream_t stream) {
    return cfg.block_len == 2 * M
               ? combine_r0<2>(cfg, twn, work_list, work_count, dsub, corr_stats, dump_corr,
                               dump_template, base, cap, stream)
               : combine_r0<4>(cfg, twn, work_list, work_count, dsub, corr_stats, dump_corr,
                               dump_template, base, cap, stream);
}

// blocks per correlate-stage chunk: the d_k0 exchange (8 * block_len * T bytes per block) of one
// chunk is kept near 128 MiB, half the Infinity Cache
int long_chunk_blocks(int block_len, int n_templates) {
    const size_t per_block = size_t(8) * block_len * n_templates;
    return int(std::max<size_t>(16, (size_t(128) << 20) / per_block));
}

}  // namespace thr
