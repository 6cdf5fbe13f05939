// Short blocks: block_len N = R1 * 1024, R1 in {1, 2, 4, 8}  (1024 ... 8192 samples).
//
// `block_size` is a free setting of the reference (settings.py:62; its own tests run 8192,
// tests/test_carrier_detect.py:57).  These lengths use the SAME LDS-resident machinery as the
// 16384 kernels (passes_w8.hpp): one 512-thread workgroup holds 16384 samples in LDS -- here
// G = 16 / R1 whole blocks side by side.  With N = R1 * 32 * 32,
//     n = n1 * 1024 + 32 n2 + m',   k = k1 + R1 * k2 + 32 R1 * k3,
// LDS row r = g * R1 + k1 holds sub-sequence k1 of block g of the group, so passes 2, 3, A and
// B (the row-local 32 x 32 part) are the 16384 code unchanged; only the first forward pass
// (radix R1 over n1, A = 32 / R1 adjacent columns per thread) and the last inverse pass are
// written per R1, the twiddles W_N^(k1 q) = W_16384^(k1 (16 / R1) q) come from the same
// L2-resident table, and reductions run per block (R1 half-waves each) instead of per workgroup.
//
//   k_carrier_small   : u8/c64 -> FFT#1 -> sum |X|^2, windowed first-max over float32 |X|,
//                       7-bin neighbourhood              (carrier_detect.py:99-154)
//   k_correlate_small : shift -> FFT#2 -> x conj(T) -> IFFT -> |.|^2 windowed first-max
//                       (carrier_sync.py:222-238, soa_estimator.py:97-143)
// k_fit / k_finish (detect16k_carrier.hip) are shared.  MODE 1 adds the sums the stddev threshold
// terms need (carrier_detect.py:110-115, soa_estimator.py:127-134), MODE 2 also the stage dumps of
// Detector.detect(yield_data=True) (detect.py:75-78) and of the test hooks.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;

namespace {

// scratch carve of the small kernels (bytes from OFF_S): [0, 512) reductions (2 parities x 16
// half-wave slots x 16 B); the per-block phasor steps e[g][j] live in the (unused) A / Bt table
// area: G * A <= 512 complex = 4 KiB of its 8 KiB
constexpr int RED_SLOT = 16;   // bytes per half-wave partial: u64 key + float sum (+ pad)

template <int R1>
struct Geo {
    static_assert(R1 == 1 || R1 == 2 || R1 == 4 || R1 == 8, "block_len = R1 * 1024");
    static constexpr int NB = R1 * 1024;   // block length
    static constexpr int A = 32 / R1;      // adjacent columns m per thread in the first / last pass
    static constexpr int G = 16 / R1;      // blocks per workgroup pass
    static constexpr int TB = 32 * R1;     // threads per block
    static constexpr int TROW = 16 / R1;   // gtw table row of sub-sequence k1: k1 * TROW
};

// ---------------------------------------------------------------- sample source
template <int FMT, int R1>
struct RawSmall;

template <int R1>
struct RawSmall<THR_IN_U8, R1> {
    static constexpr int A = Geo<R1>::A;
    unsigned q[16];   // [n1][A / 2] dwords: samples m0 + 2i, m0 + 2i + 1 of sub-sequence n1
    __device__ __forceinline__ void load(const void* __restrict__ blk, int tb) {
        const unsigned char* p = static_cast<const unsigned char*>(blk) + size_t(tb) * A * 2;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1)   // A * 2 contiguous bytes (8 ... 64): wide unaligned loads
            __builtin_memcpy(&q[n1 * (A / 2)], p + size_t(n1) * 2048, A * 2);
    }
    __device__ __forceinline__ void pair(int n1, int i, cpx& a, cpx& b) const {   // samples m0+2i, +1
        const unsigned w = q[n1 * (A / 2) + i];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;  // == (v - 127.4f) / 128 exactly
        a = cpx{fmaf(float(w & 0xffu), sc, of), fmaf(float((w >> 8) & 0xffu), sc, of)};
        b = cpx{fmaf(float((w >> 16) & 0xffu), sc, of), fmaf(float(w >> 24), sc, of)};
    }
};

template <int R1>
struct RawSmall<THR_IN_C64, R1> {
    static constexpr int A = Geo<R1>::A;
    const f4* p;
    __device__ __forceinline__ void load(const void* __restrict__ blk, int tb) {
        p = reinterpret_cast<const f4*>(blk) + size_t(tb) * (A / 2);
    }
    __device__ __forceinline__ void pair(int n1, int i, cpx& a, cpx& b) const {
        const f4 w = p[size_t(n1) * 512 + i];
        a = cpx{w.x, w.y};
        b = cpx{w.z, w.w};
    }
};

// ---------------------------------------------------------------- first forward pass
// radix R1 over n1 for the thread's A adjacent columns -> rows g*R1 + k1 of the LDS image.
// PH: x[n1] is pre-rotated by rpow[n1] and column m by p0 * e[m - m0] (the frequency shift).
// `pre` (optional, PH == false): the block-invariant twiddles already in registers
// (small_twiddles), [(k1 - 1) * A / 2 + i] = W_N^(k1 (m0 + 2i)), W_N^(k1 (m0 + 2i + 1)).
template <int R1>
__device__ __forceinline__ void small_twiddles(int tb, const cpx* __restrict__ gtw, f4* pre) {
    constexpr int A = Geo<R1>::A, TROW = Geo<R1>::TROW;
    const int m0 = tb * A;
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1)
#pragma unroll
        for (int i = 0; i < A / 2; ++i)
            pre[(k1 - 1) * (A / 2) + i] = *reinterpret_cast<const f4*>(gtw + (k1 * TROW) * 1024 + m0 + 2 * i);
}

template <int R1, bool PH, class RAW>
__device__ __forceinline__ void small_pass1(cpx* lds, const RAW& raw, int g, int tb,
                                            const float2* __restrict__ rpow, cpx p0,
                                            const cpx* __restrict__ e, const cpx* __restrict__ gtw,
                                            float* energy, const f4* pre = nullptr) {
    constexpr int A = Geo<R1>::A, TROW = Geo<R1>::TROW;
    const int m0 = tb * A;
    cpx* out = lds + (g * R1) * ROW + (m0 >> 5) * CHUNK + (m0 & 31);
    float en = 0.f;
    cpx r[R1];
    if constexpr (PH) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) r[n1] = cpx{rpow[n1].x, rpow[n1].y};
    }
    static_for<A / 2>([&](auto I) {
        constexpr int i = decltype(I)::value;
        cpx v0[R1], v1[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            raw.pair(n1, i, v0[n1], v1[n1]);
            en += cnorm(v0[n1]) + cnorm(v1[n1]);
            if constexpr (PH) {
                if (n1 != 0) {   // rpow[0] == 1
                    v0[n1] = cmul(v0[n1], r[n1]);
                    v1[n1] = cmul(v1[n1], r[n1]);
                }
            }
        }
        if constexpr (R1 > 1) {
            dft_reg<R1, -1>(v0);
            dft_reg<R1, -1>(v1);
        }
        cpx pa, pb;
        if constexpr (PH) {
            const f4 ee = *reinterpret_cast<const f4*>(e + 2 * i);   // e[2i], e[2i + 1]
            pa = cmul(p0, cpx{ee.x, ee.y});
            pb = cmul(p0, cpx{ee.z, ee.w});
        }
        static_for<R1>([&](auto K) {
            constexpr int k1 = decltype(K)::value;
            cpx y0 = v0[brev(k1, R1)], y1 = v1[brev(k1, R1)];
            if constexpr (k1 == 0) {
                if constexpr (PH) {
                    y0 = cmul(y0, pa);
                    y1 = cmul(y1, pb);
                }
            } else {
                const f4 ww = pre != nullptr
                                  ? pre[(k1 - 1) * (A / 2) + i]
                                  : *reinterpret_cast<const f4*>(gtw + (k1 * TROW) * 1024 + m0 + 2 * i);
                cpx w0 = cpx{ww.x, ww.y}, w1 = cpx{ww.z, ww.w};
                if constexpr (PH) {
                    w0 = cmul(w0, pa);
                    w1 = cmul(w1, pb);
                }
                y0 = cmul(y0, w0);
                y1 = cmul(y1, w1);
            }
            *reinterpret_cast<f4*>(out + k1 * ROW + 2 * i) = f4{y0.x, y0.y, y1.x, y1.y};
        });
    });
    if (energy != nullptr) *energy = en;
}

// ---------------------------------------------------------------- last inverse pass
// c[n1 * A + j] = corr[n1 * 1024 + m0 + j]
template <int R1>
__device__ __forceinline__ void small_passC(const cpx* lds, int g, int tb, cpx* c) {
    constexpr int A = Geo<R1>::A;
    const int m0 = tb * A;
    const cpx* in = lds + (g * R1) * ROW + (m0 >> 5) * CHUNK + (m0 & 31);
    static_for<A / 2>([&](auto I) {
        constexpr int i = decltype(I)::value;
        cpx v0[R1], v1[R1];
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const f4 q = *reinterpret_cast<const f4*>(in + k1 * ROW + 2 * i);
            v0[k1] = cpx{q.x, q.y};
            v1[k1] = cpx{q.z, q.w};
        }
        if constexpr (R1 > 1) {
            dft_reg<R1, +1>(v0);
            dft_reg<R1, +1>(v1);
        }
        static_for<R1>([&](auto K) {
            constexpr int n1 = decltype(K)::value;
            c[n1 * A + 2 * i] = v0[brev(n1, R1)];
            c[n1 * A + 2 * i + 1] = v1[brev(n1, R1)];
        });
    });
}

// ---------------------------------------------------------------- per-block reductions
// A block is R1 consecutive half-waves.  DPP reduce inside each half-wave (row_shr 1/2/4/8,
// row_bcast15: lanes 31 and 63 then hold their half's result), one LDS slot per half-wave,
// ONE barrier, every thread combines its block's R1 slots.
__device__ __forceinline__ unsigned long long halfwave_max(unsigned long long v) {
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
#undef THR_STEP
    return v;   // valid in lanes 31 and 63
}
__device__ __forceinline__ float halfwave_sum(float v) {
#define THR_STEP(CTRL, MASK) v += __uint_as_float(dpp_u32<CTRL, MASK>(0u, __float_as_uint(v)))
    THR_STEP(DPP_ROW_SHR1, 0xf);
    THR_STEP(DPP_ROW_SHR2, 0xf);
    THR_STEP(DPP_ROW_SHR4, 0xf);
    THR_STEP(DPP_ROW_SHR8, 0xf);
    THR_STEP(DPP_ROW_BCAST15, 0xa);
#undef THR_STEP
    return v;   // valid in lanes 31 and 63
}

// NS (0, 1, 2): float sums reduced along with the key (slot: u64 key, float, float)
template <int R1, int NS>
__device__ __forceinline__ void group_reduce(float (&s)[2], unsigned long long& m, unsigned char* scratch,
                                             int parity, int g) {
    unsigned char* base = scratch + parity * 16 * RED_SLOT;
    const int hw = threadIdx.x >> 5;   // half-wave index 0..15
    m = halfwave_max(m);
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = halfwave_sum(s[i]);
    if ((threadIdx.x & 31) == 31) {
        *reinterpret_cast<unsigned long long*>(base + hw * RED_SLOT) = m;
#pragma unroll
        for (int i = 0; i < NS; ++i) *reinterpret_cast<float*>(base + hw * RED_SLOT + 8 + 4 * i) = s[i];
    }
    __syncthreads();
    unsigned long long mm = 0;
    float ss[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < R1; ++i) {
        const unsigned char* slot = base + (g * R1 + i) * RED_SLOT;
        const unsigned long long v = *reinterpret_cast<const unsigned long long*>(slot);
        mm = v > mm ? v : mm;
#pragma unroll
        for (int j = 0; j < NS; ++j) ss[j] += *reinterpret_cast<const float*>(slot + 8 + 4 * j);
    }
    m = mm;
    s[0] = ss[0];
    s[1] = ss[1];
}

// kernel modes: 0 plain, 1 + the sums of the stddev threshold terms, 2 + stage dumps (and the sums)
constexpr int MODE_PLAIN = 0, MODE_STD = 1, MODE_DUMP = 2;

// =========================================================================
// carrier stage
// =========================================================================
template <int FMT, int R1, int MODE>
__global__ __launch_bounds__(NT) void k_carrier_small(const void* __restrict__ samples, int n_blocks,
                                                      DevCfg cfg, const cpx* __restrict__ tables,
                                                      const cpx* __restrict__ gtw,
                                                      CarStats* __restrict__ stats,
                                                      cpx* __restrict__ dump_fft) {
    using GE = Geo<R1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;
    const int n_groups = (n_blocks + GE::G - 1) / GE::G;
    int parity = 0;
    // block-invariant pass-1 twiddles of this thread's columns: loaded once per kernel
    f4 tw[R1 > 1 ? (R1 - 1) * (GE::A / 2) : 1];
    if constexpr (R1 > 1) small_twiddles<R1>(opaque_tid() % GE::TB, gtw, tw);
    RawSmall<FMT, R1> cur;
    {
        const int t = opaque_tid();
        const int b0 = min(int(blockIdx.x) * GE::G + t / GE::TB, n_blocks - 1);
        cur.load(static_cast<const unsigned char*>(samples) + size_t(max(b0, 0)) * blk_bytes, t % GE::TB);
    }
    for (int gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const int t = opaque_tid();
        const int g = t / GE::TB, tb = t % GE::TB;
        const int b = gi * GE::G + g;
        const bool valid = b < n_blocks;   // (tail group: the extra lanes redo the last block, store nothing)
        // (the previous group's pass-3 LDS reads all precede its reduction barrier)
        small_pass1<R1, false>(lds, cur, g, tb, nullptr, cpx{}, nullptr, gtw, nullptr, R1 > 1 ? tw : nullptr);
        // next group's samples, into the registers pass 1 has just consumed: issued now, used one
        // iteration later
        if (gi + int(gridDim.x) < n_groups) {
            const int bn = min((gi + int(gridDim.x)) * GE::G + g, n_blocks - 1);
            cur.load(static_cast<const unsigned char*>(samples) + size_t(bn) * blk_bytes, tb);
        }
        __syncthreads();
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx v[R3];
        fwd_pass3(lds, v);

        // thread (row = t >> 5 = g*R1 + k1, k2 = t & 31) holds bins k1 + R1 k2 + 32 R1 k3
        const int k1 = (t >> 5) % R1;
        const int kbase = k1 + R1 * (t & 31);
        float sums[2] = {0.f, 0.f};   // sum |X|^2, sum |X| (stddev term: MODE >= 1)
        float mg[R3];
        float bestm = -1.0f;
        unsigned bestwi = 0;
        static_for<R3>([&](auto K) {
            constexpr int k3 = decltype(K)::value;
            const float p = cnorm(v[brev(k3, R3)]);
            const float m = sqrtf(p);   // the reference's argmax runs over float32 |X| (carrier_detect.py:146)
            mg[k3] = m;
            sums[0] += p;
            if constexpr (MODE >= MODE_STD) sums[1] += m;
            if constexpr (MODE == MODE_DUMP) {
                if (valid && dump_fft != nullptr)
                    dump_fft[size_t(b) * GE::NB + kbase + GE::TB * k3] = v[brev(k3, R3)];
            }
            const unsigned wi = unsigned(kbase + GE::TB * k3 - cfg.win_lo) & unsigned(GE::NB - 1);
            const bool take = wi < unsigned(cfg.win_count) && (m > bestm || (m == bestm && wi < bestwi));
            bestm = take ? m : bestm;
            bestwi = take ? wi : bestwi;
        });
        unsigned long long best =
            bestm < 0.f ? 0ull : ((unsigned long long)__float_as_uint(bestm) << 32) | (0xFFFFFFFFu - bestwi);
        group_reduce<R1, MODE >= MODE_STD ? 2 : 1>(sums, best, sc_red, parity, g);
        parity ^= 1;
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        int peak_idx = int(wi) + cfg.win_lo;
        if (peak_idx > GE::NB) peak_idx -= GE::NB;  // sic: '>' (carrier_detect.py:151)
        if (valid) {
            CarStats* st = stats + b;
            // neighbour d of the peak lives in this thread iff (kbase - peak + 3) mod TB == d < 7
            const unsigned u = unsigned(kbase - peak_idx + 3) & unsigned(GE::NB - 1);
            const unsigned r = u & unsigned(GE::TB - 1), k3s = (32u - (u / unsigned(GE::TB))) & 31u;
            if (r < 7u) {
                float val = 0.f;
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    val = (k3s == unsigned(k3)) ? mg[k3] : val;
                });
                st->nb[r] = val;
            }
            if (tb == 0) {
                st->sum_mag2 = sums[0];
                st->sum_mag = MODE >= MODE_STD ? sums[1] : 0.f;
                st->peak_mag = __uint_as_float(unsigned(best >> 32));
                st->peak_idx = peak_idx;
                st->pad = 0;
            }
        }
    }
}

// =========================================================================
// shift + FFT#2 + matched filter + peak
// =========================================================================
// MULTI: more than one template -- the shifted spectrum stays live (64 VGPRs) across the loop
template <int FMT, int R1, bool MULTI, int MODE>
__global__ __launch_bounds__(NT) void k_correlate_small(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ gtw, const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, CorrStats* __restrict__ corr_stats,
    cpx* __restrict__ dump_xhat, cpx* __restrict__ dump_corr, int dump_template) {
    using GE = Geo<R1>;
    constexpr int A = GE::A;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    cpx* sc_e = lds + OFF_A;   // [G][A] phasor steps exp(2 pi i s j / N), j < A (A / Bt tables are unused here)

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;
    const int n_work = *work_count;
    const int n_groups = (n_work + GE::G - 1) / GE::G;
    const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
    int parity = 0;
    for (int gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const int t = opaque_tid();
        const int g = t / GE::TB, tb = t % GE::TB;
        const int slot = gi * GE::G + g;
        const bool valid = slot < n_work;
        const int b = work_list[valid ? slot : n_work - 1];
        const ShiftParams* sp = shifts + b;
        RawSmall<FMT, R1> raw;
        raw.load(static_cast<const unsigned char*>(samples) + size_t(b) * blk_bytes, tb);
        // shift phasor exp(2 pi i s (m / N - 1/2)): integer part of s through the exact root table,
        // fractional part through a small-angle sincosf; column m0 per thread (p0), and the steps
        // to the thread's other A - 1 columns per block (e[j], shared through LDS)
        const int si = sp->si_mod;
        const float sf = sp->sf_over_n;
        auto phasor = [&](int m) {
            const cpx wq = cconj(twn[(si * m) & (GE::NB - 1)]);   // exp(+2 pi i (si m mod N) / N)
            float sn, cs;
            sincosf(6.283185307179586f * (sf * float(m)), &sn, &cs);
            return cmul(wq, cpx{cs, sn});
        };
        const cpx p0 = cmul(phasor(tb * A), cpx{sp->c0.x, sp->c0.y});
        if (tb < A) sc_e[g * A + tb] = phasor(tb);
        __syncthreads();   // e[] visible; the previous group's last LDS reads are done
        small_pass1<R1, true>(lds, raw, g, tb, sp->rpow, p0, sc_e + g * A, gtw, nullptr);
        __syncthreads();
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx xh[R3];
        fwd_pass3(lds, xh);

        const int k1 = (t >> 5) % R1;
        const int tcol = k1 * 32 + (t & 31);   // == position of this thread's bins in the template slice
        if constexpr (MODE == MODE_DUMP) {
            if (valid && dump_xhat != nullptr)
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    dump_xhat[size_t(b) * GE::NB + k1 + R1 * (t & 31) + GE::TB * k3] = xh[brev(k3, R3)];
                });
        }
        const int n_tpl = MULTI ? cfg.n_templates : 1;
        for (int tpl = 0; tpl < n_tpl; ++tpl) {
            const int t2 = opaque_tid();  // re-derive per template: keeps LICM off the loop body
            const int g2 = t2 / GE::TB, tb2 = t2 % GE::TB;
            const f4* ts = tspec + size_t(tpl) * (GE::NB / 2) + tcol;
            cpx z[R3];
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = ts[j * GE::TB];
                z[brev(2 * j, R3)] = cmul(xh[brev(2 * j, R3)], cpx{q.x, q.y});
                z[brev(2 * j + 1, R3)] = cmul(xh[brev(2 * j + 1, R3)], cpx{q.z, q.w});
            });
            // pass A overwrites exactly the chunk this thread read in pass 3 (for tpl > 0: rows
            // whose pass-C readers are behind the previous reduction barrier)
            inv_passA(lds, z);
            __builtin_amdgcn_sched_barrier(0);
            inv_passB<true>(lds, gtw, k1 * GE::TROW);
            __syncthreads();
            cpx c[32];
            small_passC<R1>(lds, g2, tb2, c);
            // |corr|^2, windowed first-max: lags visited in increasing n (n1 outer, column inner)
            float pw[32];
            float bestp = -1.0f;
            int bestn = 0;
            float sums[2] = {0.f, 0.f};   // sum |corr|, sum |corr|^2 over [0, corr_len): MODE >= 1
            const int m0 = tb2 * A;
            static_for<32>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                constexpr int n1 = q / A, j = q % A;
                pw[q] = cnorm(c[q]);
                const int n = n1 * 1024 + m0 + j;
                const bool take = unsigned(n - cfg.corr_lo) < win_w && pw[q] > bestp;
                bestp = take ? pw[q] : bestp;
                bestn = take ? n : bestn;
                if constexpr (MODE >= MODE_STD) {
                    if (n < cfg.corr_len) {
                        sums[1] += pw[q];
                        sums[0] += __builtin_amdgcn_sqrtf(pw[q]);
                    }
                }
                if constexpr (MODE == MODE_DUMP) {
                    if (valid && dump_corr != nullptr && tpl == dump_template)
                        dump_corr[size_t(b) * GE::NB + n] = c[q];
                }
            });
            unsigned long long best =
                bestp < 0.f ? 0ull
                            : ((unsigned long long)__float_as_uint(bestp) << 32) | (0xFFFFFFFFu - unsigned(bestn));
            group_reduce<R1, MODE >= MODE_STD ? 2 : 0>(sums, best, sc_red, parity, g2);
            parity ^= 1;
            const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
            if (valid) {
                CorrStats* cs = corr_stats + size_t(b) * cfg.n_templates + tpl;
                // lag pk - 1 + d = n1 * 1024 + m0 + j: this thread owns it iff its column range holds
                // (pk - 1 + d) mod 1024
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int n = pk - 1 + d;
                    const int col = (n & 1023) - m0, n1s = n >> 10;
                    if (n >= 0 && n < GE::NB && col >= 0 && col < A) {
                        const int want = n1s * A + col;
                        float val = 0.f;
                        static_for<32>([&](auto Q) {
                            constexpr int q = decltype(Q)::value;
                            val = (want == q) ? pw[q] : val;
                        });
                        cs->m2[d] = val;
                    } else if (tb2 == 0 && (n < 0 || n >= GE::NB)) {
                        cs->m2[d] = 0.f;   // (peak at an array edge: k_finish does not use it)
                    }
                }
                if (tb2 == 0) {
                    cs->pm2 = __uint_as_float(unsigned(best >> 32));
                    cs->pk = pk;
                    cs->sum_mag = MODE >= MODE_STD ? sums[0] : 0.f;
                    cs->sum_mag2 = MODE >= MODE_STD ? sums[1] : 0.f;
                }
            }
        }
    }
}

typedef void (*carrier_small_fn)(const void*, int, DevCfg, const cpx*, const cpx*, CarStats*, cpx*);
typedef void (*correlate_small_fn)(const void*, DevCfg, const cpx*, const cpx*, const cpx*, const f4*,
                                   const ShiftParams*, const int*, const int*, CorrStats*, cpx*, cpx*, int);

template <int R1, int FMT>
carrier_small_fn carrier_fn(int mode) {
    return mode == MODE_DUMP ? &k_carrier_small<FMT, R1, MODE_DUMP>
           : mode == MODE_STD ? &k_carrier_small<FMT, R1, MODE_STD>
                              : &k_carrier_small<FMT, R1, MODE_PLAIN>;
}
// (the dump mode exists in the several-template form only: it serves any template count)
template <int R1, int FMT>
correlate_small_fn correlate_fn(bool multi, int mode) {
    if (mode == MODE_DUMP) return &k_correlate_small<FMT, R1, true, MODE_DUMP>;
    if (multi)
        return mode == MODE_STD ? &k_correlate_small<FMT, R1, true, MODE_STD>
                                : &k_correlate_small<FMT, R1, true, MODE_PLAIN>;
    return mode == MODE_STD ? &k_correlate_small<FMT, R1, false, MODE_STD>
                            : &k_correlate_small<FMT, R1, false, MODE_PLAIN>;
}
template <int R1>
carrier_small_fn carrier_r1(int fmt, int mode) {
    return fmt == THR_IN_U8 ? carrier_fn<R1, THR_IN_U8>(mode) : carrier_fn<R1, THR_IN_C64>(mode);
}
template <int R1>
correlate_small_fn correlate_r1(int fmt, bool multi, int mode) {
    return fmt == THR_IN_U8 ? correlate_fn<R1, THR_IN_U8>(multi, mode) : correlate_fn<R1, THR_IN_C64>(multi, mode);
}
carrier_small_fn pick_carrier(int r1, int fmt, int mode) {
    switch (r1) {
        case 1: return carrier_r1<1>(fmt, mode);
        case 2: return carrier_r1<2>(fmt, mode);
        case 4: return carrier_r1<4>(fmt, mode);
        default: return carrier_r1<8>(fmt, mode);
    }
}
correlate_small_fn pick_correlate(int r1, int fmt, bool multi, int mode) {
    switch (r1) {
        case 1: return correlate_r1<1>(fmt, multi, mode);
        case 2: return correlate_r1<2>(fmt, multi, mode);
        case 4: return correlate_r1<4>(fmt, multi, mode);
        default: return correlate_r1<8>(fmt, multi, mode);
    }
}

}  // namespace

bool small_supported(int block_len) {
    return block_len == 1024 || block_len == 2048 || block_len == 4096 || block_len == 8192;
}

hipError_t prepare_small(int block_len) {
    const int r1 = block_len / 1024;
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int mode = 0; mode < 3; ++mode) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_carrier(r1, fmt, mode)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
            if (e != hipSuccess) return e;
            for (int multi = 0; multi < 2; ++multi) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(pick_correlate(r1, fmt, multi != 0, mode)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                if (e != hipSuccess) return e;
            }
        }
    return hipSuccess;
}

hipError_t launch_carrier_small(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                const float2* tables, const float2* gtw, CarStats* stats, float2* dump_fft,
                                int n_cu, hipStream_t stream) {
    const int r1 = cfg.block_len / 1024, groups = (n_blocks + 16 / r1 - 1) / (16 / r1);
    const int mode = dump_fft != nullptr ? MODE_DUMP : cfg.car_want_std ? MODE_STD : MODE_PLAIN;
    hipLaunchKernelGGL(pick_carrier(r1, fmt, mode), dim3(std::min(groups, n_cu)), dim3(NT), LDS_BYTES, stream,
                       samples, n_blocks, cfg, reinterpret_cast<const cpx*>(tables),
                       reinterpret_cast<const cpx*>(gtw), stats, reinterpret_cast<cpx*>(dump_fft));
    return hipGetLastError();
}

hipError_t launch_correlate_small(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                  const float2* tables, const float2* gtw, const float2* twn,
                                  const float4* tspec, const ShiftParams* shifts, const int* work_list,
                                  const int* work_count, CorrStats* corr_stats, float2* dump_xhat,
                                  float2* dump_corr, int dump_template, int n_cu, hipStream_t stream) {
    const int r1 = cfg.block_len / 1024, groups = (n_blocks + 16 / r1 - 1) / (16 / r1);
    const int mode = (dump_xhat != nullptr || dump_corr != nullptr) ? MODE_DUMP
                     : cfg.cor_want_std                              ? MODE_STD
                                                                     : MODE_PLAIN;
    hipLaunchKernelGGL(pick_correlate(r1, fmt, cfg.n_templates > 1, mode), dim3(std::min(groups, n_cu)),
                       dim3(NT), LDS_BYTES, stream, samples, cfg, reinterpret_cast<const cpx*>(tables),
                       reinterpret_cast<const cpx*>(gtw), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list, work_count, corr_stats,
                       reinterpret_cast<cpx*>(dump_xhat), reinterpret_cast<cpx*>(dump_corr), dump_template);
    return hipGetLastError();
}

}  // namespace thr
