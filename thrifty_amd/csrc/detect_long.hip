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
    __device__ __forceinline__ void fetch() { fetch_from(blk, t); }
    __device__ __forceinline__ void fetch_from(const void* src, int tt) {
        if constexpr (HELD) {
#pragma unroll
            for (int n0 = 0; n0 < R0; ++n0)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1)
                    w[n0][n1] = reinterpret_cast<const unsigned*>(src)[size_t(n0) * (M / 2) +
                                                                       size_t(n1) * (S1 / 2) + tt];
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
                a = cmul_acc(a, x, wg);
                b = cmul_acc(b, y, wg);
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
    static constexpr bool kBytes = true;   // (fwd_pass1_pre: affine map after the butterfly)
    __device__ __forceinline__ unsigned word(int n1) const { return q[n1]; }
    __device__ __forceinline__ void get_bytes(int n1, cpx& a, cpx& b) const {
        const unsigned v = q[n1];
        a = cpx{float(v & 0xffu), float((v >> 8) & 0xffu)};
        b = cpx{float((v >> 16) & 0xffu), float(v >> 24)};
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
    static constexpr bool kBytes = false;
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
    pass1_twiddles(lds, tw0, tw1, pass1_scale<RawDecim<FMT, R0>>());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
      RawDecim<FMT, R0> raw;
      raw.fetch(static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes, opaque_tid());
      float energy = 0.f;
      // u8: the loop over the R0 decimated sequences is unrolled -- picking the sequence's pair out
      // of the held words is then register naming plus one byte permute per word instead of
      // selects on r; complex64 (loads at use): rolled, or the compiler hoists R0 x 32 loads
      auto item = [&](int r) __attribute__((always_inline)) {
        const int t = opaque_tid();
        raw.select(r);
        float sums[1];
        // (the previous item's pass-3 LDS reads precede its reduction barrier)
        fwd_pass1_pre(lds, raw, tw0, tw1, &sums[0]);
        __syncthreads();
        fwd_pass2<8>(lds);
        __builtin_amdgcn_sched_barrier(0);
        // (four threads per kept chunk, 8 terms each, then a quad sum: as k_carrier_pruned)
        {
            const int k2 = (t >> 2) & 7, k1 = t >> 5, part = t & 3;
            const f4* src = reinterpret_cast<const f4*>(lds + k1 * ROW + k2 * CHUNK) + part * (R3 / 8);
            f4 acc = src[0];
#pragma unroll
            for (int j = 1; j < R3 / 8; ++j) acc += src[j];
            const cpx x = cpx{quad_sum(acc.x + acc.z), quad_sum(acc.y + acc.w)};
            const int k = k1 + 16 * k2;
            const unsigned wi = unsigned(k - win_base);
            if (part == 0 && wi < unsigned(win_w)) win_f[(size_t(b) * R0 + r) * win_w + wi] = x;
        }
        // sum |x|^2: the R0 decimated sequences' shares add up in the thread; ONE reduction per
        // block (k_select_dit adds the R0 partial entries: the total sits in the last, zeros before)
        energy += sums[0];
        if (r < R0 - 1) {
            __syncthreads();   // (this item's pass-3 LDS reads precede the next item's pass-1 writes)
            if (t == 0) {
                partial[(size_t(b) * R0 + r) * 2 + 0] = 0.f;
                partial[(size_t(b) * R0 + r) * 2 + 1] = 0.f;
            }
        } else {
            const double tot = block_sum<NT / 64>(energy, sc_red, parity);
            parity ^= 1;
            if (t == 0) {
                partial[(size_t(b) * R0 + r) * 2 + 0] = (float)(tot * double(R0 * M));  // Parseval
                partial[(size_t(b) * R0 + r) * 2 + 1] = 0.f;
            }
        }
      };
      if constexpr (FMT == THR_IN_U8) {
          static_for<R0>([&](auto R) { item(decltype(R)::value); });
      } else {
#pragma nounroll
          for (int r = 0; r < R0; ++r) item(r);
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

constexpr int CMB_T = 512;   // threads of the combination (== NT: the fused form runs it in place)

// corr[n0 M + m] = sum_k0 W_R0^(-n0 k0) conj(W_NL^(m k0)) d_k0[m]; windowed first-max, sums
template <int R0>
__device__ __forceinline__ void combine_at(const cpx* __restrict__ d, const cpx* __restrict__ twn,
                                           int m, int nl_mask, cpx (&out)[R0]) {
    cpx u[R0];
    u[0] = d[m];
#pragma unroll
    for (int k0 = 1; k0 < R0; ++k0) u[k0] = cmulc(d[size_t(k0) * M + m], twn[(m * k0) & nl_mask]);
    dft_reg<R0, +1>(u);
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) out[n0] = u[brev(n0, R0)];
}

// HBM / Infinity-Cache-bound byte work (8 R0 M bytes read per (slot, template), ~30 flop per
// 32 bytes): 512 threads, two adjacent lags per thread and step (one 16-byte load per
// sub-transform).  (Measured and dropped: running it on a second stream under the
// sub-transforms of the next chunk -- 1.74 M blocks/s at N = 65536 against 1.71 M serial; with
// the raw block held in registers by k_correlate_sub, 230 VGPRs, the two kernels no longer fit
// on a CU together and serial order wins: 1.78 M against 1.18 M.)
// The combination twiddles that are the same for every thread: stab[i (R0 - 1) + k0 - 1] =
// W_NL^(1024 i k0), i = 0 .. 15 -- 16 (R0 - 1) numbers in LDS, written once per kernel.
template <int R0>
__device__ __forceinline__ void combine_table(cpx* stab, const cpx* __restrict__ twn) {
    const int i = threadIdx.x;
    if (i < 16 * (R0 - 1))
        stab[i] = twn[(2 * CMB_T * (i / (R0 - 1)) * (i % (R0 - 1) + 1)) & (R0 * M - 1)];
}

// the R0 running maxima of a thread as one (power, -lag) key
template <int R0>
__device__ __forceinline__ unsigned long long combine_key(const float (&bp)[R0], const int (&bn)[R0]) {
    unsigned long long best = 0;
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) {
        // key 0 for "no lag inside the window" (bp = -1: the only negative power), as mask
        // arithmetic: a 64-bit select here becomes a branch, and LLVM sinks the arithmetic of the
        // other R0 - 1 chains below it -- with every radix-R0 intermediate spilled across
        const unsigned bits = __float_as_uint(bp[n0]);
        const unsigned valid = ~unsigned(int(bits) >> 31);
        const unsigned long long key =
            ((unsigned long long)(bits & valid) << 32) | ((0xFFFFFFFFu - unsigned(bn[n0])) & valid);
        best = key > best ? key : best;
    }
    return best;
}

// The peak and its two neighbours, recombined from the parked rows by three threads (one lag
// each): request() issues their loads -- R0 row values and R0 - 1 root-table twiddles, one
// round trip of ~2 us -- and complete() turns them into the CorrStats entries.  The fused
// kernel calls complete() one barrier into the NEXT sub-transform: wave 0 then reaches that
// barrier without having waited for the round trip, and nobody else waits for wave 0.
template <int R0>
struct PeakTail {
    cpx dv[R0], wv[R0];
    unsigned long long best = 0;
    int b = -1, tpl = 0;
    __device__ __forceinline__ void request(const cpx* __restrict__ twn, const cpx* d, int b_, int tpl_,
                                            unsigned long long best_, int tid) {
        b = b_;
        tpl = tpl_;
        best = best_;
        if (tid < 3) {
            const int NL = R0 * M;
            const int n = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu)) - 1 + tid;
            const int m = (n >= 0 && n < NL) ? n % M : 0;
#pragma unroll
            for (int k0 = 0; k0 < R0; ++k0) {
                dv[k0] = d[size_t(k0) * M + m];
                wv[k0] = twn[(m * k0) & (NL - 1)];
            }
        }
    }
    __device__ __forceinline__ void complete(const DevCfg& cfg, CorrStats* __restrict__ corr_stats,
                                             int tid, float sum_mag, float sum_mag2) {
        if (b >= 0 && tid < 3) {
            const int NL = R0 * M;
            CorrStats* cs = corr_stats + size_t(b) * cfg.n_templates + tpl;
            const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
            const int n = pk - 1 + tid;
            float v = 0.f;
            if (n >= 0 && n < NL) {
                cpx u[R0];
                u[0] = dv[0];
#pragma unroll
                for (int k0 = 1; k0 < R0; ++k0) u[k0] = cmulc(dv[k0], wv[k0]);
                dft_reg<R0, +1>(u);
                cpx sel = u[0];
#pragma unroll
                for (int n0 = 1; n0 < R0; ++n0) sel = (n / M == n0) ? u[brev(n0, R0)] : sel;
                v = cnorm(sel);
            }
            cs->m2[tid] = v;
            if (tid == 0) {
                cs->pm2 = __uint_as_float(unsigned(best >> 32));
                cs->pk = pk;
                cs->sum_mag = sum_mag;
                cs->sum_mag2 = sum_mag2;
            }
        }
        b = -1;
    }
};

// End of a combination: the R0 running maxima of a thread meet in one (power, -lag) key, the
// workgroup reduces it (ONE barrier), and three threads write the peak and its two neighbours
// (recombined from the parked rows `d`: one lag each, nobody waits for them).
template <int R0, bool WANT_STD>
__device__ __forceinline__ void combine_finish(const DevCfg& cfg, const cpx* __restrict__ twn,
                                               const cpx* d, int b, int tpl,
                                               CorrStats* __restrict__ corr_stats,
                                               const float (&bp)[R0], const int (&bn)[R0],
                                               float (&sums)[2], int tid, unsigned char* scratch,
                                               int parity) {
    unsigned long long best = combine_key<R0>(bp, bn);
    double tot[2] = {0, 0};
    if constexpr (WANT_STD) {
        block_reduce<2, CMB_T / 64>(sums, tot, best, scratch, parity);
    } else {
        block_reduce_max<CMB_T / 64>(best, scratch, parity);
    }
    PeakTail<R0> tail;
    tail.request(twn, d, b, tpl, best, tid);
    tail.complete(cfg, corr_stats, tid, (float)tot[0], (float)tot[1]);
}

// One (block, template): combine the R0 sub-transform outputs `d` ([R0][M]), windowed first-max,
// optional std sums, the peak's neighbours -> corr_stats.  All CMB_T threads of the workgroup.
template <int R0, bool WANT_STD, bool DUMPC>
__device__ __forceinline__ void combine_impl(const DevCfg& cfg, const cpx* __restrict__ twn,
                                             const cpx* stab, const cpx* d, int b, int tpl,
                                             CorrStats* __restrict__ corr_stats,
                                             cpx* __restrict__ dump_corr, unsigned char* scratch,
                                             int parity) {
    const int T = cfg.n_templates;
    const int NL = R0 * M, nl_mask = NL - 1;
    float sums[2] = {0.f, 0.f};
    const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
    // The thread's lags are n = n0 M + m (+ 1), m = 2 tid + 1024 i, i = 0 .. 15: inside one n0 they
    // come in increasing order, so a running (power, lag) per n0 with a strict '>' keeps the
    // first maximum; the R0 of them meet in one key at the end.
    float bp[R0];
    int bn[R0];
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) { bp[n0] = -1.f; bn[n0] = 0; }
    // Combination twiddles W_NL^(m k0) = W_NL^(2 tid k0) * W_NL^(1024 i k0): the first factor is
    // gathered once per block (2 (R0 - 1) loads), the second is the same for every thread (stab,
    // LDS broadcast) -- one complex product per use instead of a gather from the 8 NL-byte root
    // table.  And the loads of UN steps are issued together: with one workgroup per CU (the fused
    // form) nothing else hides their latency.
    const int tid = opaque_tid();   // (keeps the per-thread twiddles out of the caller's block loop)
    cpx wb0[R0], wb1[R0];
#pragma unroll
    for (int k0 = 1; k0 < R0; ++k0) {
        wb0[k0] = twn[(2 * tid * k0) & nl_mask];
        wb1[k0] = twn[((2 * tid + 1) * k0) & nl_mask];
    }
    constexpr int STEPS = M / (2 * CMB_T), UN = 4;
    static_assert(STEPS % UN == 0, "combination steps come in groups of UN");
#pragma unroll 1
    for (int i0 = 0; i0 < STEPS; i0 += UN) {
        f4 q[UN][R0];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k0 = 0; k0 < R0; ++k0)
                q[u][k0] = *reinterpret_cast<const f4*>(d + size_t(k0) * M + 2 * tid +
                                                        2 * CMB_T * (i0 + u));
        cpx s[UN][R0];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k0 = 1; k0 < R0; ++k0) s[u][k0] = stab[(i0 + u) * (R0 - 1) + k0 - 1];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = 2 * tid + 2 * CMB_T * (i0 + u);
            cpx u0[R0], u1[R0];
            u0[0] = cpx{q[u][0].x, q[u][0].y};
            u1[0] = cpx{q[u][0].z, q[u][0].w};
#pragma unroll
            for (int k0 = 1; k0 < R0; ++k0) {
                u0[k0] = cmulc(cpx{q[u][k0].x, q[u][k0].y}, cmul(wb0[k0], s[u][k0]));
                u1[k0] = cmulc(cpx{q[u][k0].z, q[u][k0].w}, cmul(wb1[k0], s[u][k0]));
            }
            dft_reg<R0, +1>(u0);
            dft_reg<R0, +1>(u1);
#pragma unroll
            for (int n0 = 0; n0 < R0; ++n0) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const cpx c = e ? u1[brev(n0, R0)] : u0[brev(n0, R0)];
                    const int n = n0 * M + m + e;
                    const float pw = cnorm(c);
                    const bool take = (unsigned(n - cfg.corr_lo) < win_w) & (pw > bp[n0]);   // (&: no branch)
                    bp[n0] = take ? pw : bp[n0];
                    bn[n0] = take ? n : bn[n0];
                    if constexpr (WANT_STD) {
                        if (n < cfg.corr_len) {
                            sums[1] += pw;
                            sums[0] += __builtin_amdgcn_sqrtf(pw);
                        }
                    }
                    if constexpr (DUMPC) dump_corr[size_t(b) * NL + n] = c;
                }
            }
        }
    }
    combine_finish<R0, WANT_STD>(cfg, twn, d, b, tpl, corr_stats, bp, bn, sums, tid, scratch, parity);
}

template <int R0>
__device__ __forceinline__ void combine_block(const DevCfg& cfg, const cpx* __restrict__ twn,
                                              const cpx* stab, const cpx* d, int b, int tpl,
                                              CorrStats* __restrict__ corr_stats,
                                              cpx* __restrict__ dump_corr, int dump_template,
                                              unsigned char* scratch, int parity) {
    // (the two run-time options once per block, not once per lag)
    if (dump_corr != nullptr && tpl == dump_template)
        combine_impl<R0, true, true>(cfg, twn, stab, d, b, tpl, corr_stats, dump_corr, scratch, parity);
    else if (cfg.cor_want_std)
        combine_impl<R0, true, false>(cfg, twn, stab, d, b, tpl, corr_stats, nullptr, scratch, parity);
    else
        combine_impl<R0, false, false>(cfg, twn, stab, d, b, tpl, corr_stats, nullptr, scratch, parity);
}

// The fused form's combination, in the thread that produced the rows: after pass C of a block's
// LAST sub-transform a thread holds d_{R0-1}[m] for its 32 lags m = n1 1024 + 2t (+ 1) in
// registers (c0, c1), and the same lags of d_0 .. d_{R0-2} are what IT wrote to `rows` when those
// sub-transforms ended -- own writes, program order, no barrier.  They come back four n1 at a time,
// the next four requested before the current four are combined; the caller ends with
// combine_finish.  wb0 / wb1: W_NL^((2t + e) k0), the thread's own factors of the combination
// twiddles (kept for the whole kernel).  (No std sums, no stage dump:
// batches that want either take the from-memory form -- one variant here, on purpose: with three
// behind a run-time choice LLVM hoists their common arithmetic above the branch and spills all
// 128 combined lags.)
template <int R0>
__device__ __forceinline__ void combine_own(const DevCfg& cfg, const cpx* stab, const f4* rows,
                                            const cpx* c0, const cpx* c1, const cpx (&wb0)[R0],
                                            const cpx (&wb1)[R0], float (&bp)[R0], int (&bn)[R0]) {
    const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) { bp[n0] = -1.f; bn[n0] = 0; }
    const int tid = opaque_tid();
    const int rel = 2 * tid - cfg.corr_lo;
    constexpr int G = 2;     // n1 per group
    constexpr int AHEAD = 2; // groups requested ahead of the one being combined (their latency --
                             // rows this CU wrote tens of microseconds ago, long out of L2 -- is
                             // 1.5 - 2 us, a group's arithmetic 0.7 us)
    constexpr int NG = R1 / G;
    f4 q[AHEAD + 1][G][R0 - 1];
    auto request = [&](auto GRP) {
        constexpr int g = decltype(GRP)::value;
        constexpr int buf = g % (AHEAD + 1);
        static_for<G>([&](auto JI) {
            constexpr int j = decltype(JI)::value;
            static_for<R0 - 1>([&](auto KI) {
                constexpr int k0 = decltype(KI)::value;
                q[buf][j][k0] = rows[size_t(k0) * (M / 2) + (g * G + j) * (S1 / 2) + tid];
            });
        });
    };
    static_for<AHEAD>([&](auto GI) { request(GI); });
    static_for<NG>([&](auto GI) {
        constexpr int g = decltype(GI)::value;
        constexpr int buf = g % (AHEAD + 1);
        if constexpr (g + AHEAD < NG) request(std::integral_constant<int, g + AHEAD>{});
        __builtin_amdgcn_sched_barrier(0);   // (AHEAD groups ahead, not all sixteen rows at once)
        static_for<G>([&](auto JI) {
            constexpr int n1 = g * G + decltype(JI)::value;
            constexpr int j = decltype(JI)::value;
            cpx u0[R0], u1[R0];
            u0[0] = cpx{q[buf][j][0].x, q[buf][j][0].y};
            u1[0] = cpx{q[buf][j][0].z, q[buf][j][0].w};
            cpx sv[R0];   // (the uniform factors of this n1 together: one wait, not one per read)
#pragma unroll
            for (int k0 = 1; k0 < R0; ++k0) sv[k0] = stab[n1 * (R0 - 1) + k0 - 1];
#pragma unroll
            for (int k0 = 1; k0 < R0; ++k0) {
                const cpx s = sv[k0];
                const cpx a0 = k0 < R0 - 1 ? cpx{q[buf][j][k0 < R0 - 1 ? k0 : 0].x, q[buf][j][k0 < R0 - 1 ? k0 : 0].y}
                                           : c0[brev(n1, R1)];
                const cpx a1 = k0 < R0 - 1 ? cpx{q[buf][j][k0 < R0 - 1 ? k0 : 0].z, q[buf][j][k0 < R0 - 1 ? k0 : 0].w}
                                           : c1[brev(n1, R1)];
                u0[k0] = cmulc(a0, cmul(wb0[k0], s));
                u1[k0] = cmulc(a1, cmul(wb1[k0], s));
            }
            dft_reg<R0, +1>(u0);
            dft_reg<R0, +1>(u1);
#pragma unroll
            for (int n0 = 0; n0 < R0; ++n0) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const cpx c = e ? u1[brev(n0, R0)] : u0[brev(n0, R0)];
                    // lag n = n0 M + n1 1024 + 2 tid + e: inside the window iff
                    // unsigned(rel + constant) < win_w; remembered as the inline constant 2 n1 + e
                    const float pw = cnorm(c);
                    const bool take = (unsigned(rel + (n0 * M + n1 * S1 + e)) < win_w) & (pw > bp[n0]);   // (&: no branch)
                    bp[n0] = take ? pw : bp[n0];
                    bn[n0] = take ? 2 * n1 + e : bn[n0];
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int n0 = 0; n0 < R0; ++n0) bn[n0] = n0 * M + (bn[n0] >> 1) * S1 + 2 * tid + (bn[n0] & 1);
}

// =========================================================================
// correlation stage of one sub-transform: shift, FFT, x template, inverse -> d_k0[m]
// =========================================================================
// FUSED: one launch over the whole work list; a workgroup runs the R0 sub-transforms of a block
// back to back (block-major order), parks their outputs in ITS OWN exchange rows (dsub row
// blockIdx.x: written and read back by the same CU, L2 / Infinity-Cache traffic that overlaps
// the other CUs' transforms) and combines them at once -- no second kernel, no chunking, and the
// bandwidth-bound combination of one CU runs under the VALU/LDS-bound transforms of the others.
// Needs at least one block per workgroup (n_work >= gridDim.x); smaller batches spread the
// sub-transforms of a block over several workgroups and take the two-kernel form (!FUSED, then
// k_combine), which returns at once when the fused kernel has done the batch.
// MULTI = false (fused only): exactly one template, no std sums, no stage dump -- the last
// sub-transform's outputs are combined straight from the registers of the thread that produced
// them (combine_own); otherwise (several templates: the shifted spectrum stays live across the
// template loop, 64 VGPRs) the rows are combined from memory after it (combine_block), as the
// two-kernel form does.
template <int FMT, int R0, bool DUMP, bool FUSED, bool MULTI = true>
__global__ __launch_bounds__(NT) void k_correlate_sub(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, int slot_base, int slot_cap, int fused_grid,
    f4* dsub,           // [slot - slot_base | workgroup][tpl][k0][M/2] float4
    f4* __restrict__ xhat_scratch, cpx* __restrict__ dump_xhat, CorrStats* __restrict__ corr_stats,
    cpx* __restrict__ dump_corr, int dump_template) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float2* sc_rp0 = reinterpret_cast<float2*>(sc_red + 2 * red_slot_bytes<NT / 64>());  // 2 x (16 + R0)

    load_tables(lds, tables);
    // (scratch map: [0, 512) reductions, [512, 832) item factors, [1024, 1024 + 128 (R0 - 1)) the
    // combination's uniform twiddles)
    cpx* stab = reinterpret_cast<cpx*>(sc_red + 1024);
    if constexpr (FUSED) combine_table<R0>(stab, twn);
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
    const int T = MULTI ? cfg.n_templates : 1;
    if constexpr (FUSED) {
        if (n_work < int(gridDim.x)) return;        // (the two-kernel form does this batch)
    } else {
        if (*work_count >= fused_grid) return;      // (the fused kernel has done it)
    }

    const bool per_block = FUSED || n_work >= int(gridDim.x);
    int parity = 0;
    PeakTail<R0> tail;      // fused, one template: the previous block's peak, completed a barrier later
    cpx wb0[R0], wb1[R0];   // combine_own's per-thread twiddle factors
    if constexpr (FUSED && !MULTI) {
#pragma unroll
        for (int k0 = 1; k0 < R0; ++k0) {
            wb0[k0] = twn[(2 * int(threadIdx.x) * k0) & nl_mask];
            wb1[k0] = twn[((2 * int(threadIdx.x) + 1) * k0) & nl_mask];
        }
    }
    const int n_iter = per_block ? ((n_work - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x)) * R0
                                 : (n_work * R0 - int(blockIdx.x) + int(gridDim.x) - 1) / int(gridDim.x);
    // An item's uniform preparation -- its block index (a scalar load), the item factors (two
    // dependent gathers, then LDS) and, on a new block, the block's shift phasor -- is done ONE
    // ITEM AHEAD, after the pass-1 barrier of the item before: at the top of an item nothing
    // waits for a chain of loads any more.
    auto item_of = [&](int it) {
        return per_block ? (int(blockIdx.x) + (it / R0) * int(gridDim.x)) * R0 + it % R0
                         : int(blockIdx.x) + it * int(gridDim.x);
    };
    auto block_phasor = [&](const ShiftParams* sp, int t) {
        // per-thread phasor for m' = 2t, 2t+1: c0 * exp(2 pi i s m'/NL) [per block]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = 2 * t + e;
            const int q = int(((long long)sp->si_mod * m) & nl_mask);
            float sn, cs;   // |2 pi sf m| <= 2 pi (0.5 / NL) 1023 < 0.1 rad
            sincos_small(6.283185307179586f * (sp->sf_over_n * float(m)), &sn, &cs);
            ph[e] = cmul(cmul(cconj(twn[q]), cpx{cs, sn}), cpx{sp->c0.x, sp->c0.y});
        }
    };
    int b_cur = 0;
    if (n_iter > 0) {
        const int item0 = item_of(0);
        b_cur = work_list[item0 / R0];
        item_factors<R0>(sc_rp0, sc_rp0 + 16, shifts + b_cur, twn, item0 % R0, nl_mask);
        block_phasor(shifts + b_cur, opaque_tid());
        ph_block = b_cur;
    }
    for (int it = 0; it < n_iter; ++it) {
        const int item = item_of(it);
        const int slot = item / R0, k0 = item % R0;
        const int b = b_cur;
        const int t = opaque_tid();
        // item factors are double-buffered: written one item ahead (below), after the pass-1 barrier
        // of the item that reads the other buffer
        float2* sc_rp = sc_rp0 + (it & 1) * (16 + R0);
        float2* sc_g = sc_rp + 16;
        cpx p[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) p[e] = cmul(ph[e], twn[((2 * t + e) * k0) & nl_mask]);
        __syncthreads();
        raw.blk = static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes;
        raw.g = sc_g;
        raw.t = t;
        raw.k0 = k0;
        // block-major work order: the raw block is fetched once per R0 sub-transforms -- fused: the
        // workgroup's first block here, every later one at the end of the block before it (below)
        if (FUSED ? it == 0 : b != raw_block) {
            raw.fetch();
            raw_block = b;
        }
        raw.prepare();
        fwd_pass1<true>(lds, raw, sc_rp, p[0], p[1]);
        __syncthreads();
        if constexpr (FUSED && !MULTI) tail.complete(cfg, corr_stats, opaque_tid(), 0.f, 0.f);
        if (it + 1 < n_iter) {   // the next item's uniform preparation (p of THIS item is formed)
            const int item_n = item_of(it + 1);
            const int b_n = work_list[item_n / R0];
            float2* rp_n = sc_rp0 + ((it + 1) & 1) * (16 + R0);
            item_factors<R0>(rp_n, rp_n + 16, shifts + b_n, twn, item_n % R0, nl_mask);
            if (b_n != ph_block) {
                block_phasor(shifts + b_n, t);
                ph_block = b_n;
            }
            b_cur = b_n;
        }
        // template 0's spectrum slice: requested here, used after pass 3 -- at the point of use its
        // L2 latency stood in front of every sub-transform (1.62 -> 1.38 ms for the fused kernel).
        // (One template only: with several, the spectrum and the slices together do not fit.)
        f4 tq[R3 / 2];
        if constexpr (!MULTI) {
            const f4* ts = tspec + size_t(k0) * (M / 2) + t;
            static_for<R3 / 2>([&](auto J) { tq[decltype(J)::value] = ts[decltype(J)::value * NT]; });
        }
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
            if (MULTI || tpl > 0) {
                const f4* ts = tspec + (size_t(tpl) * R0 + k0) * (M / 2) + t;
                static_for<R3 / 2>([&](auto J) { tq[decltype(J)::value] = ts[decltype(J)::value * NT]; });
            }
            cpx z[R3];
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = tq[j];
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
            const int row = FUSED ? int(blockIdx.x) : slot;
            f4* out = dsub + ((size_t(row) * T + tpl) * R0 + k0) * (M / 2) + t;
            // (fused, last sub-transform: this row is read back only by the three threads that
            // recombine the peak's neighbours, behind the reduction barrier)
            static_for<R1>([&](auto K) {
                constexpr int n1 = decltype(K)::value;
                out[n1 * (S1 / 2)] = f4{c0[brev(n1, R1)].x, c0[brev(n1, R1)].y, c1[brev(n1, R1)].x,
                                        c1[brev(n1, R1)].y};
            });
            if constexpr (FUSED && !MULTI) {
                if (k0 == R0 - 1) {
                    const f4* rows = dsub + (size_t(blockIdx.x) * T + tpl) * (size_t(R0) * (M / 2));
                    float bp[R0];
                    int bn[R0];
                    combine_own<R0>(cfg, stab, rows, c0, c1, wb0, wb1, bp, bn);
                    // the workgroup's next block: its raw words are requested here, between the
                    // lags and the reduction -- their latency hides under the reduction, the
                    // neighbour recombination and the next block's preamble (earlier, their 16 R0
                    // registers would be live through the combination)
                    // (after the last block: the same block again, harmless -- a branch here
                    // makes LLVM sink the tails of the R0 lag chains below it, spills and all)
                    raw.fetch_from(static_cast<const unsigned char*>(samples) +
                                       size_t(work_list[it + 1 < n_iter ? slot + int(gridDim.x) : slot]) *
                                           blk_bytes,
                                   opaque_tid());
                    unsigned long long best = combine_key<R0>(bp, bn);
                    block_reduce_max<CMB_T / 64>(best, sc_red, parity);
                    parity ^= 1;
                    tail.request(twn, reinterpret_cast<const cpx*>(rows), b, tpl, best, opaque_tid());
                }
            }
        }
        if constexpr (FUSED && MULTI) {
            if (k0 == R0 - 1) {
                // all R0 x T rows of this block are written; the barrier makes them visible to the
                // workgroup (same CU, same L1).  The next block's first stores come several
                // barriers after these reads.
                __syncthreads();
                for (int tpl = 0; tpl < T; ++tpl) {
                    combine_block<R0>(cfg, twn, stab,
                                      reinterpret_cast<const cpx*>(dsub) +
                                          (size_t(blockIdx.x) * T + tpl) * (size_t(R0) * M),
                                      b, tpl, corr_stats, dump_corr, dump_template, sc_red, parity);
                    parity ^= 1;
                }
            }
        }
        if constexpr (FUSED && MULTI) {
            // the workgroup's next block: its raw words are requested at the end of the block
            // before it
            if (k0 == R0 - 1 && it + 1 < n_iter)
                raw.fetch_from(static_cast<const unsigned char*>(samples) +
                                   size_t(work_list[slot + int(gridDim.x)]) * blk_bytes, opaque_tid());
        }
    }
    if constexpr (FUSED && !MULTI) tail.complete(cfg, corr_stats, opaque_tid(), 0.f, 0.f);   // (the last block's)
}

// Combination as its own kernel: one workgroup per (slot, template) of a chunk.  Batches with
// fewer carrier-positive blocks than `fused_grid` only -- larger ones are combined by the
// workgroup that ran the block's sub-transforms (k_correlate_sub<..., FUSED>).
template <int R0>
__global__ __launch_bounds__(CMB_T) void k_combine(DevCfg cfg, const cpx* __restrict__ twn,
                                                   const cpx* __restrict__ dsub,
                                                   const int* __restrict__ work_list,
                                                   const int* __restrict__ work_count,
                                                   int slot_base, int fused_grid,
                                                   CorrStats* __restrict__ corr_stats,
                                                   cpx* __restrict__ dump_corr, int dump_template) {
    __shared__ __attribute__((aligned(16))) unsigned char scratch[2 * (CMB_T / 64) * 32];
    __shared__ cpx stab[16 * (R0 - 1)];
    const int T = cfg.n_templates;
    const int slot = blockIdx.x / T, tpl = blockIdx.x % T;   // chunk-local; work_list points at slot_base
    if (*work_count >= fused_grid || slot_base + slot >= *work_count) return;
    combine_table<R0>(stab, twn);
    __syncthreads();
    combine_block<R0>(cfg, twn, stab, dsub + (size_t(slot) * T + tpl) * (R0 * M), work_list[slot],
                      tpl, corr_stats, dump_corr, dump_template, scratch, 0);
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
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, false, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, true, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, false, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, true, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, false, true>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, true, true>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, false, true>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, true, true>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, false, true, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_U8, R0, true, true, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, false, true, false>),
        reinterpret_cast<const void*>(&k_correlate_sub<THR_IN_C64, R0, true, true, false>)};
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

template <int FMT, int R0>
hipError_t correlate_launch(bool fused, const void* samples, const DevCfg& cfg, const float2* tables,
                            const float2* twn, const float4* tspec, const ShiftParams* shifts,
                            const int* work_list, const int* work_count, float2* dsub,
                            float4* xhat_scratch, float2* dump_xhat, CorrStats* corr_stats,
                            float2* dump_corr, int dump_template, int grid, int base, int cap,
                            int fused_grid, hipStream_t stream) {
    typedef void (*fn_t)(const void*, DevCfg, const cpx*, const cpx*, const f4*, const ShiftParams*,
                         const int*, const int*, int, int, int, f4*, f4*, cpx*, CorrStats*, cpx*, int);
    const bool dump = dump_xhat != nullptr;
    fn_t fn = !fused ? (dump ? &k_correlate_sub<FMT, R0, true, false> : &k_correlate_sub<FMT, R0, false, false>)
              : (cfg.n_templates > 1 || cfg.cor_want_std != 0 || dump_corr != nullptr)
                  ? (dump ? &k_correlate_sub<FMT, R0, true, true> : &k_correlate_sub<FMT, R0, false, true>)
                  : (dump ? &k_correlate_sub<FMT, R0, true, true, false>
                          : &k_correlate_sub<FMT, R0, false, true, false>);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list + base, work_count, base,
                       cap, fused_grid, reinterpret_cast<f4*>(dsub), reinterpret_cast<f4*>(xhat_scratch),
                       reinterpret_cast<cpx*>(dump_xhat), corr_stats, reinterpret_cast<cpx*>(dump_corr),
                       dump_template);
    return hipGetLastError();
}

// fused: ONE launch over the whole work list, `grid` workgroups, dsub holds `grid` rows.
// !fused: ONE slot chunk [base, base + cap) (dsub holds one chunk, see long_chunk_blocks); the
// caller follows it with launch_combine_long.
template <int R0>
hipError_t correlate_r0(bool fused, int fmt, const void* samples, const DevCfg& cfg,
                        const float2* tables, const float2* twn, const float4* tspec,
                        const ShiftParams* shifts, const int* work_list, const int* work_count,
                        float2* dsub, float4* xhat_scratch, float2* dump_xhat, CorrStats* corr_stats,
                        float2* dump_corr, int dump_template, int grid, int base, int cap,
                        int fused_grid, hipStream_t stream) {
    const int g = fused ? grid : std::min(grid, cap * R0);
    return fmt == THR_IN_U8
               ? correlate_launch<THR_IN_U8, R0>(fused, samples, cfg, tables, twn, tspec, shifts,
                                                 work_list, work_count, dsub, xhat_scratch, dump_xhat,
                                                 corr_stats, dump_corr, dump_template, g, base, cap,
                                                 fused_grid, stream)
               : correlate_launch<THR_IN_C64, R0>(fused, samples, cfg, tables, twn, tspec, shifts,
                                                  work_list, work_count, dsub, xhat_scratch, dump_xhat,
                                                  corr_stats, dump_corr, dump_template, g, base, cap,
                                                  fused_grid, stream);
}

template <int R0>
hipError_t combine_r0(const DevCfg& cfg, const float2* twn, const int* work_list,
                      const int* work_count, const float2* dsub, CorrStats* corr_stats,
                      float2* dump_corr, int dump_template, int base, int cap, int fused_grid,
                      hipStream_t stream) {
    hipLaunchKernelGGL(k_combine<R0>, dim3(cap * cfg.n_templates), dim3(CMB_T), 0, stream, cfg,
                       reinterpret_cast<const cpx*>(twn), reinterpret_cast<const cpx*>(dsub),
                       work_list + base, work_count, base, fused_grid, corr_stats,
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

hipError_t launch_correlate_long(bool fused, int fmt, const void* samples, const DevCfg& cfg,
                                 const float2* tables, const float2* twn, const float4* tspec,
                                 const ShiftParams* shifts, const int* work_list,
                                 const int* work_count, float2* dsub, float4* xhat_scratch,
                                 float2* dump_xhat, CorrStats* corr_stats, float2* dump_corr,
                                 int dump_template, int grid, int base, int cap, int fused_grid,
                                 hipStream_t stream) {
    return cfg.block_len == 2 * M
               ? correlate_r0<2>(fused, fmt, samples, cfg, tables, twn, tspec, shifts, work_list,
                                 work_count, dsub, xhat_scratch, dump_xhat, corr_stats, dump_corr,
                                 dump_template, grid, base, cap, fused_grid, stream)
               : correlate_r0<4>(fused, fmt, samples, cfg, tables, twn, tspec, shifts, work_list,
                                 work_count, dsub, xhat_scratch, dump_xhat, corr_stats, dump_corr,
                                 dump_template, grid, base, cap, fused_grid, stream);
}

hipError_t launch_combine_long(const DevCfg& cfg, const float2* twn, const int* work_list,
                               const int* work_count, const float2* dsub, CorrStats* corr_stats,
                               float2* dump_corr, int dump_template, int base, int cap,
                               int fused_grid, hipStream_t stream) {
    return cfg.block_len == 2 * M
               ? combine_r0<2>(cfg, twn, work_list, work_count, dsub, corr_stats, dump_corr,
                               dump_template, base, cap, fused_grid, stream)
               : combine_r0<4>(cfg, twn, work_list, work_count, dsub, corr_stats, dump_corr,
                               dump_template, base, cap, fused_grid, stream);
}

// blocks per correlate-stage chunk: the d_k0 exchange (8 * block_len * T bytes per block) of one
// chunk is kept near 128 MiB, half the Infinity Cache
int long_chunk_blocks(int block_len, int n_templates) {
    const size_t per_block = size_t(8) * block_len * n_templates;
    return int(std::max<size_t>(16, (size_t(128) << 20) / per_block));
}

}  // namespace thr
