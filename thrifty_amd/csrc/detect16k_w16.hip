// LDS-resident fused detection kernels for block_len = 16384, 16-wave geometry.
//
// Same pipeline and record semantics as detect16k.hip, different workgroup shape:
// 1024 threads (16 waves, 4 per SIMD) x 16 elements per thread, four register passes
// 16 x 16 x 4 x 16 with three LDS exchanges.  Compared with the 8-wave 16x32x32 kernels
// this trades +50 % LDS traffic and one more (cheap, table-driven) twiddle stage for
// twice the waves per SIMD: a gfx950 wave issues at most one VALU op per ~5.4 cycles, so
// two waves per SIMD cannot keep the VALU fed through LDS waits and barriers; four can.
//
// Index algebra (forward, decimation in frequency):
//   n = n1*1024 + n2*64 + n3*16 + n4         k = k1 + 16*k2 + 256*k3 + 1024*k4
//   P1: DFT16 over n1, twiddle W_N^(m*k1)        m  = n mod 1024      thread t = m
//   P2: DFT16 over n2, twiddle W_1024^(m2*k2)    m2 = n mod 64        thread (k1 = wave, m2)
//   P3: DFT4  over n3, twiddle W_64^(n4*k3)                           thread (k1, k2, g): n4 = g+4j
//   P4: DFT16 over n4 -> k4, registers                                thread (k1, k2, k3)
// Row k1 belongs to wave k1 from P2 through the inverse's third pass, so only two
// workgroup barriers per FFT pair remain.  The inverse is the exact mirror (decimation
// in time, conjugate twiddles), consuming the digit-reversed spectrum where it lies.
//
// LDS layout: element (row r, q in [0,1024)) at r*1088 + q + 4*(q>>6) + ((q>>4)&3), i.e.
// 64-element blocks pitched 68 and 16-element chunks pitched 17: every ds_read/write_b64
// pattern of the eight passes is bank-conflict-free except P2's reads (one 2-way pair
// per half wave).
#include <hip/hip_runtime.h>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"

namespace thr {

namespace w16 {
constexpr int N = 16384;
constexpr int NT = 1024, NW = 16;
constexpr int R1 = 16, R2 = 16, R3 = 4, R4 = 16;
constexpr int S1 = 1024;
constexpr int CH = 17;            // 16-element chunk + 1 pad
constexpr int BLK = 4 * CH;       // 64-element block -> 68
constexpr int ROW = 16 * BLK;     // 1088
constexpr int DATA = 16 * ROW;    // 17408 complex
// LDS carve (complex units)
constexpr int OFF_T2 = DATA;            // T2[k2][m2] = W_1024^(m2*k2)   16 x 64
constexpr int OFF_A = OFF_T2 + 1024;    // A[k1][mh]  = W_512^(mh*k1)    16 x 32
constexpr int OFF_B = OFF_A + 512;      // B[k1][ml]  = W_N^(ml*k1)      16 x 32
constexpr int OFF_T3 = OFF_B + 512;     // T3[k3][n4] = W_64^(n4*k3)      4 x 16
constexpr int OFF_S = OFF_T3 + 64;      // 2 KiB scratch (reductions, pruned bins)
constexpr int TABLE_CPX = 1024 + 512 + 512 + 64;  // 2112
constexpr int LDS_CPX = OFF_S + 256;
constexpr size_t LDS_BYTES = size_t(LDS_CPX) * sizeof(cpx);  // 158,208 B

__device__ __forceinline__ int pad_q(int q) { return q + 4 * (q >> 6) + ((q >> 4) & 3); }
__device__ __forceinline__ int pad_m2(int m2) { return m2 + (m2 >> 4); }

__device__ __forceinline__ void load_tables(cpx* lds, const cpx* __restrict__ tables) {
    // 2112 complex = 1056 float4: threads 0..1023 one each, first 32 a second one
    const f4* src = reinterpret_cast<const f4*>(tables);
    f4* dst = reinterpret_cast<f4*>(lds + OFF_T2);
    dst[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x < TABLE_CPX / 2 - NT) dst[threadIdx.x + NT] = src[threadIdx.x + NT];
}

// ---------------------------------------------------------------- sample load
// Per-thread raw samples of one block: sample m = t of each sub-sequence n1
// (n = n1*1024 + t).  u8 flavour: 16 VGPRs (one I,Q pair each), prefetchable.
template <int FMT>
struct Raw;

template <>
struct Raw<THR_IN_U8> {
    unsigned short q[R1];
    __device__ __forceinline__ void load(const void* __restrict__ blk, int t) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(blk) + t;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) q[n1] = p[n1 * S1];
    }
    __device__ __forceinline__ cpx get(int n1) const {
        const unsigned w = q[n1];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;  // == (v - 127.4f) / 128 exactly
        return cpx{fmaf(float(w & 0xffu), sc, of), fmaf(float(w >> 8), sc, of)};
    }
};

template <>
struct Raw<THR_IN_C64> {
    const cpx* p;
    __device__ __forceinline__ void load(const void* __restrict__ blk, int t) {
        p = reinterpret_cast<const cpx*>(blk) + t;
    }
    __device__ __forceinline__ cpx get(int n1) const { return p[n1 * S1]; }
};

// ------------------------------------------------------------ forward passes
// P1: thread t = m.  If PH: x[n1] *= rpow[n1], and the per-m phasor p folds into the twiddle.
template <int FMT, bool PH>
__device__ __forceinline__ void fwd_p1(cpx* lds, const Raw<FMT>& raw,
                                       const float2* __restrict__ rpow, cpx p,
                                       float* energy = nullptr) {
    const int t = opaque_tid();
    cpx v[R1];
    float e = 0.f;
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        v[n1] = raw.get(n1);
        if (energy != nullptr) e += cnorm(v[n1]);
        if constexpr (PH) v[n1] = cmul(v[n1], cpx{rpow[n1].x, rpow[n1].y});
    }
    if (energy != nullptr) *energy = e;
    dft_dif<R1, -1>(v);
    const cpx* tA = lds + OFF_A + (t >> 5);
    const cpx* tB = lds + OFF_B + (t & 31);
    cpx* out = lds + pad_q(t);
    static_for<R1>([&](auto K) {
        constexpr int k1 = decltype(K)::value;
        cpx y = v[brev(k1, R1)];
        if constexpr (k1 == 0) {
            if constexpr (PH) y = cmul(y, p);
        } else {
            cpx w = cmul(tA[k1 * 32], tB[k1 * 32]);
            if constexpr (PH) w = cmul(w, p);
            y = cmul(y, w);
        }
        out[k1 * ROW] = y;
    });
}

// P2: thread (k1 = wave, m2 = lane).  KEEP < 16: only outputs k2 < KEEP are written (pruned).
template <int KEEP = R2>
__device__ __forceinline__ void fwd_p2(cpx* lds) {
    const int t = opaque_tid();
    const int m2 = t & 63;
    cpx* base = lds + (t >> 6) * ROW + pad_m2(m2);
    const cpx* tw = lds + OFF_T2 + m2;
    cpx v[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = base[n2 * BLK];
    dft_dif<R2, -1>(v);
    static_for<KEEP>([&](auto K) {
        constexpr int k2 = decltype(K)::value;
        cpx y = v[brev(k2, R2)];
        if constexpr (k2 != 0) y = cmul(y, tw[k2 * 64]);
        base[k2 * BLK] = y;
    });
}

// P3: thread (k1, k2 = (t>>2)&15, g = t&3): four DFT4 over n3 at n4 = g + 4j, in place.
__device__ __forceinline__ void fwd_p3(cpx* lds) {
    const int t = opaque_tid();
    const int g = t & 3;
    cpx* base = lds + (t >> 6) * ROW + ((t >> 2) & 15) * BLK + g;
    const cpx* tw = lds + OFF_T3 + g;
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        cpx v[R3];
#pragma unroll
        for (int n3 = 0; n3 < R3; ++n3) v[n3] = base[n3 * CH + 4 * j];
        dft_dif<R3, -1>(v);
        static_for<R3>([&](auto K) {
            constexpr int k3 = decltype(K)::value;
            cpx y = v[brev(k3, R3)];
            if constexpr (k3 != 0) y = cmul(y, tw[k3 * 16 + 4 * j]);
            base[k3 * CH + 4 * j] = y;
        });
    });
}

// P4: thread (k1, k2, k3 = t&3): DFT16 over n4, registers only.
// On return bin k = k1 + 16*k2 + 256*k3 + 1024*k4 is in v[brev(k4, 16)].
__device__ __forceinline__ void fwd_p4(const cpx* lds, cpx* v) {
    const int t = opaque_tid();
    const cpx* base = lds + (t >> 6) * ROW + CH * (t & 63);
#pragma unroll
    for (int n4 = 0; n4 < R4; ++n4) v[n4] = base[n4];
    dft_dif<R4, -1>(v);
}

// ------------------------------------------------------------ inverse passes
// PA: thread (k1, k2, k3): DFT16 over k4 (input z[brev(k4)]), twiddle conj W_64^(n4*k3).
__device__ __forceinline__ void inv_pa(cpx* lds, cpx* z) {
    const int t = opaque_tid();
    const int k3 = t & 3;
    cpx v[R4];
    static_for<R4>([&](auto K) {
        constexpr int k4 = decltype(K)::value;
        v[k4] = z[brev(k4, R4)];
    });
    dft_dif<R4, +1>(v);
    cpx* base = lds + (t >> 6) * ROW + CH * (t & 63);
    const cpx* tw = lds + OFF_T3 + k3 * 16;  // k3 == 0: all ones (multiplication kept: uniform code)
    static_for<R4>([&](auto K) {
        constexpr int n4 = decltype(K)::value;
        cpx y = v[brev(n4, R4)];
        if constexpr (n4 != 0) y = cmulc(y, tw[n4]);
        base[n4] = y;
    });
}

// PB: thread (k1, k2, g): four DFT4 over k3 at n4 = g + 4j, twiddle conj W_1024^(m2*k2).
__device__ __forceinline__ void inv_pb(cpx* lds) {
    const int t = opaque_tid();
    const int g = t & 3, k2 = (t >> 2) & 15;
    cpx* base = lds + (t >> 6) * ROW + k2 * BLK + g;
    const cpx* tw = lds + OFF_T2 + k2 * 64 + g;
    static_for<4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        cpx v[R3];
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) v[k3] = base[k3 * CH + 4 * j];
        dft_dif<R3, +1>(v);
        static_for<R3>([&](auto K) {
            constexpr int n3 = decltype(K)::value;
            // m2 = n3*16 + n4, n4 = g + 4j
            base[n3 * CH + 4 * j] = cmulc(v[brev(n3, R3)], tw[n3 * 16 + 4 * j]);
        });
    });
}

// PC: thread (k1 = wave, m2): DFT16 over k2, twiddle conj W_N^(k1*(64*n2 + m2)), in place.
__device__ __forceinline__ void inv_pc(cpx* lds) {
    const int t = opaque_tid();
    const int k1 = t >> 6, m2 = t & 63;
    cpx* base = lds + k1 * ROW + pad_m2(m2);
    cpx v[R2];
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k2] = base[k2 * BLK];
    dft_dif<R2, +1>(v);
    // m = 64*n2 + m2 = 32*(2*n2 + (m2>>5)) + (m2&31)
    const cpx b = lds[OFF_B + k1 * 32 + (m2 & 31)];
    const cpx* tA = lds + OFF_A + k1 * 32 + (m2 >> 5);
    static_for<R2>([&](auto K) {
        constexpr int n2 = decltype(K)::value;
        const cpx w = cmul(tA[2 * n2], b);
        base[n2 * BLK] = cmulc(v[brev(n2, R2)], w);
    });
}

// PD: thread t = m: DFT16 over k1; c[brev(n1)] = corr[n1*1024 + t].
__device__ __forceinline__ void inv_pd(const cpx* lds, cpx* c) {
    const int t = opaque_tid();
    const cpx* base = lds + pad_q(t);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) c[k1] = base[k1 * ROW];
    dft_dif<R1, +1>(c);
}

// =========================================================================
// K_A (full spectrum)
// =========================================================================
template <int FMT, bool WANT_STD, bool DUMP>
__global__ __launch_bounds__(NT) void k_carrier_w16(const void* __restrict__ samples, int n_blocks,
                                                    DevCfg cfg, const cpx* __restrict__ tables,
                                                    CarStats* __restrict__ stats,
                                                    cpx* __restrict__ dump_fft) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    int parity = 0;

    Raw<FMT> cur;
    if (int(blockIdx.x) < n_blocks)
        cur.load(static_cast<const unsigned char*>(samples) + size_t(blockIdx.x) * blk_bytes,
                 opaque_tid());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        Raw<FMT> nxt = cur;
        if (b + int(gridDim.x) < n_blocks)
            nxt.load(static_cast<const unsigned char*>(samples) + size_t(b + gridDim.x) * blk_bytes,
                     opaque_tid());
        fwd_p1<FMT, false>(lds, cur, nullptr, cpx{});
        cur = nxt;
        __syncthreads();
        fwd_p2(lds);
        __builtin_amdgcn_sched_barrier(0);
        fwd_p3(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx v[R4];
        fwd_p4(lds, v);

        const int t = opaque_tid();
        const int kb = (t >> 6) + 16 * ((t >> 2) & 15) + 256 * (t & 3);  // bin = kb + 1024*k4
        float sums[2] = {0.f, 0.f};
        float pw[R4];
        float bestp = -1.0f;
        unsigned bestwi = 0;
        static_for<R4>([&](auto K) {
            constexpr int k4 = decltype(K)::value;
            const float p = cnorm(v[brev(k4, R4)]);
            pw[k4] = p;
            sums[0] += p;
            if constexpr (WANT_STD) sums[1] += __builtin_amdgcn_sqrtf(p);
            const unsigned wi = unsigned(kb + 1024 * k4 - cfg.win_lo) & unsigned(N - 1);
            const bool take = wi < unsigned(cfg.win_count) &&
                              (p > bestp || (p == bestp && wi < bestwi));
            bestp = take ? p : bestp;
            bestwi = take ? wi : bestwi;
        });
        unsigned long long best =
            bestp < 0.f ? 0ull
                        : ((unsigned long long)__float_as_uint(bestp) << 32) | (0xFFFFFFFFu - bestwi);
        double tot[2];
        block_reduce<WANT_STD ? 2 : 1, NW>(reinterpret_cast<float(&)[WANT_STD ? 2 : 1]>(sums),
                                           reinterpret_cast<double(&)[WANT_STD ? 2 : 1]>(tot), best,
                                           sc_red, parity);
        parity ^= 1;
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        int peak_idx = int(wi) + cfg.win_lo;
        if (peak_idx > N) peak_idx -= N;  // sic: '>' (carrier_detect.py:151)
        // neighbour d of the fit window is bin kb + 1024*k4 iff (kb - peak + 3) mod 1024 == d
        CarStats* st = stats + b;
        {
            const unsigned u = unsigned(kb - peak_idx + 3) & unsigned(N - 1);
            const unsigned r = u & 1023u, k4s = (16u - (u >> 10)) & 15u;
            float val = 0.f;
            static_for<R4>([&](auto K) {
                constexpr int k4 = decltype(K)::value;
                val = (k4s == unsigned(k4)) ? pw[k4] : val;
            });
            if (r < 7u) st->nb[r] = sqrtf(val);
        }
        if constexpr (DUMP) {
            cpx* out = dump_fft + size_t(b) * N;
            static_for<R4>([&](auto K) {
                constexpr int k4 = decltype(K)::value;
                out[kb + 1024 * k4] = v[brev(k4, R4)];
            });
        }
        if (t == 0) {
            st->sum_mag2 = (float)tot[0];
            st->sum_mag = WANT_STD ? (float)tot[1] : 0.f;
            st->peak_mag = sqrtf(__uint_as_float(unsigned(best >> 32)));
            st->peak_idx = peak_idx;
            st->pad = 0;
        }
    }
}

// =========================================================================
// K_A, pruned (window + fit margin inside bins [0,128): k2 < 8, k3 = k4 = 0)
// =========================================================================
template <int FMT>
__global__ __launch_bounds__(NT) void k_carrier_pruned_w16(const void* __restrict__ samples,
                                                           int n_blocks, DevCfg cfg,
                                                           const cpx* __restrict__ tables,
                                                           CarStats* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float* sc_bins = reinterpret_cast<float*>(sc_red + 2 * red_slot_bytes<NW>());  // [128]

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    int parity = 0;

    Raw<FMT> cur;
    if (int(blockIdx.x) < n_blocks)
        cur.load(static_cast<const unsigned char*>(samples) + size_t(blockIdx.x) * blk_bytes,
                 opaque_tid());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        Raw<FMT> nxt = cur;
        if (b + int(gridDim.x) < n_blocks)
            nxt.load(static_cast<const unsigned char*>(samples) + size_t(b + gridDim.x) * blk_bytes,
                     opaque_tid());
        float sums[1];
        fwd_p1<FMT, false>(lds, cur, nullptr, cpx{}, &sums[0]);
        cur = nxt;
        __syncthreads();
        fwd_p2<8>(lds);
        __builtin_amdgcn_sched_barrier(0);
        // bins k = k1 + 16*k2 (k2 < 8): X = sum of the 64 elements of block k2; thread
        // (k1, k2, q = t&3) adds chunk q, the quad finishes with two DPP adds
        const int t = opaque_tid();
        const int k2 = (t >> 2) & 15;
        const int k = (t >> 6) + 16 * k2;
        unsigned long long best = 0;
        {
            const cpx* src = lds + (t >> 6) * ROW + CH * (t & 63);
            cpx acc = src[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) acc += src[i];
            acc.x += __uint_as_float(dpp_u32<0xB1, 0xf>(0u, __float_as_uint(acc.x)));  // quad_perm [1,0,3,2]
            acc.y += __uint_as_float(dpp_u32<0xB1, 0xf>(0u, __float_as_uint(acc.y)));
            acc.x += __uint_as_float(dpp_u32<0x4E, 0xf>(0u, __float_as_uint(acc.x)));  // quad_perm [2,3,0,1]
            acc.y += __uint_as_float(dpp_u32<0x4E, 0xf>(0u, __float_as_uint(acc.y)));
            if (k2 < 8 && (t & 3) == 0) {
                const float p = cnorm(acc);
                sc_bins[k] = p;
                const unsigned wi = unsigned(k - cfg.win_lo) & unsigned(N - 1);
                if (wi < unsigned(cfg.win_count))
                    best = ((unsigned long long)__float_as_uint(p) << 32) | (0xFFFFFFFFu - wi);
            }
        }
        double tot[1];
        block_reduce<1, NW>(sums, tot, best, sc_red, parity);
        parity ^= 1;
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        const int peak_idx = int(wi) + cfg.win_lo;  // < 128: the '> N' wrap cannot trigger
        CarStats* st = stats + b;
        if (t < 7) st->nb[t] = sqrtf(sc_bins[peak_idx - 3 + t]);
        if (t == 0) {
            st->sum_mag2 = (float)(tot[0] * double(N));  // Parseval
            st->sum_mag = 0.f;
            st->peak_mag = sqrtf(__uint_as_float(unsigned(best >> 32)));
            st->peak_idx = peak_idx;
            st->pad = 0;
        }
    }
}

// =========================================================================
// K_B: shift + FFT#2 + matched filter + peak statistics
// =========================================================================
__device__ __forceinline__ cpx shift_phasor1(const ShiftParams* __restrict__ sp,
                                             const cpx* __restrict__ twn, int m) {
    const int q = (sp->si_mod * m) & (N - 1);
    const cpx wq = cconj(twn[q]);  // exp(+2 pi i q / N)
    float sn, cs;
    sincosf(6.283185307179586f * (sp->sf_over_n * float(m)), &sn, &cs);
    return cmul(cmul(wq, cpx{cs, sn}), cpx{sp->c0.x, sp->c0.y});
}

template <int FMT, bool WANT_STD, bool MULTI, bool DUMP>
__global__ __launch_bounds__(NT) void k_correlate_w16(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, CorrStats* __restrict__ corr_stats,
    thr_record* __restrict__ records, f4* __restrict__ xhat_scratch,
    cpx* __restrict__ dump_xhat, cpx* __restrict__ dump_corr, int dump_template) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    const int n_work = *work_count;
    int parity = 0;

    Raw<FMT> cur;
    cpx p = cpx{0.f, 0.f};
    int b_next = int(blockIdx.x) < n_work ? work_list[blockIdx.x] : 0;
    int b_next2 = int(blockIdx.x + gridDim.x) < n_work ? work_list[blockIdx.x + gridDim.x] : 0;
    if (int(blockIdx.x) < n_work) {
        cur.load(static_cast<const unsigned char*>(samples) + size_t(b_next) * blk_bytes,
                 opaque_tid());
        p = shift_phasor1(shifts + b_next, twn, opaque_tid());
    }
    for (int wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const int b = b_next;
        const int t = opaque_tid();
        const ShiftParams* sp = shifts + b;
        Raw<FMT> nxt = cur;
        const bool more = wi + int(gridDim.x) < n_work;
        if (more) {
            b_next = b_next2;
            nxt.load(static_cast<const unsigned char*>(samples) + size_t(b_next) * blk_bytes, t);
            if (wi + 2 * int(gridDim.x) < n_work) b_next2 = work_list[wi + 2 * gridDim.x];
        }
        fwd_p1<FMT, true>(lds, cur, sp->rpow, p);
        cur = nxt;
        if (more) p = shift_phasor1(shifts + b_next, twn, t);
        __syncthreads();
        fwd_p2(lds);
        __builtin_amdgcn_sched_barrier(0);
        fwd_p3(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx xh[R4];
        fwd_p4(lds, xh);

        const int kb = (t >> 6) + 16 * ((t >> 2) & 15) + 256 * (t & 3);
        float e2 = 0.f;
#pragma unroll
        for (int i = 0; i < R4; ++i) e2 += cnorm(xh[i]);
        asm volatile("" : "+v"(e2));  // pin here: else LLVM sinks the sum (and the live xh) to its use
        if constexpr (DUMP) {
            if (dump_xhat != nullptr) {
                cpx* out = dump_xhat + size_t(b) * N;
                static_for<R4>([&](auto K) {
                    constexpr int k4 = decltype(K)::value;
                    out[kb + 1024 * k4] = xh[brev(k4, R4)];
                });
            }
        }
        f4* park = nullptr;
        if constexpr (MULTI) {
            park = xhat_scratch + size_t(blockIdx.x) * (N / 2) + t;
            static_for<R4 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                park[j * NT] = f4{xh[brev(2 * j, R4)].x, xh[brev(2 * j, R4)].y,
                                  xh[brev(2 * j + 1, R4)].x, xh[brev(2 * j + 1, R4)].y};
            });
        }

        const int n_tpl = MULTI ? cfg.n_templates : 1;
        for (int tpl = 0; tpl < n_tpl; ++tpl) {
            const int t = opaque_tid();  // re-derive per template: keeps LICM off the loop body
            if constexpr (MULTI) park = xhat_scratch + size_t(blockIdx.x) * (N / 2) + t;
            const f4* ts = tspec + size_t(tpl) * (N / 2) + t;
            cpx z[R4];
            static_for<R4 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = ts[j * NT];
                cpx x0, x1;
                if constexpr (MULTI) {
                    const f4 xx = park[j * NT];  // own writes: program order suffices
                    x0 = cpx{xx.x, xx.y};
                    x1 = cpx{xx.z, xx.w};
                } else {
                    x0 = xh[brev(2 * j, R4)];
                    x1 = xh[brev(2 * j + 1, R4)];
                }
                z[brev(2 * j, R4)] = cmul(x0, cpx{q.x, q.y});
                z[brev(2 * j + 1, R4)] = cmul(x1, cpx{q.z, q.w});
            });
            inv_pa(lds, z);
            __builtin_amdgcn_sched_barrier(0);
            inv_pb(lds);
            __builtin_amdgcn_sched_barrier(0);
            inv_pc(lds);
            __syncthreads();
            cpx c[R1];
            inv_pd(lds, c);

            // ---- |corr|^2 over lags n = n1*1024 + t: windowed first-max (+ std sums)
            float sums[3] = {tpl == 0 ? e2 : 0.f, 0.f, 0.f};
            float pw[R1];
            float bestp = -1.0f;
            int bestn = 0;
            const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
            static_for<R1>([&](auto K) {
                constexpr int n1 = decltype(K)::value;
                const int n = n1 * S1 + t;
                const float v = cnorm(c[brev(n1, R1)]);
                pw[n1] = v;
                const bool take = unsigned(n - cfg.corr_lo) < win_w && v > bestp;
                bestp = take ? v : bestp;
                bestn = take ? n : bestn;
                if constexpr (WANT_STD) {
                    if (n < cfg.corr_len) {
                        sums[2] += v;
                        sums[1] += __builtin_amdgcn_sqrtf(v);
                    }
                }
            });
            unsigned long long best =
                bestp < 0.f ? 0ull
                            : ((unsigned long long)__float_as_uint(bestp) << 32) |
                                  (0xFFFFFFFFu - unsigned(bestn));
            constexpr int NS = WANT_STD ? 3 : 1;
            double tot[3] = {0, 0, 0};
            block_reduce<NS, NW>(reinterpret_cast<float(&)[NS]>(sums),
                                 reinterpret_cast<double(&)[NS]>(tot), best, sc_red, parity);
            parity ^= 1;
            const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
            // |corr[pk-1+d]|^2, d = 0..2, is held by thread t iff (pk - 1 + d - t) mod 1024 == 0
            CorrStats* cs = corr_stats + size_t(b) * cfg.n_templates + tpl;
            {
                const int delta = pk - 1 - t;
                const unsigned d = unsigned(-delta) & 1023u;
                const int n1s = (delta + int(d)) >> 10;
                float val = 0.f;
                static_for<R1>([&](auto K) {
                    constexpr int n1 = decltype(K)::value;
                    val = (n1s == n1) ? pw[n1] : val;
                });
                if (d < 3u && n1s >= 0 && n1s < R1) cs->m2[d] = val;
            }
            if constexpr (DUMP) {
                if (dump_corr != nullptr && tpl == dump_template) {
                    cpx* out = dump_corr + size_t(b) * N;
                    static_for<R1>([&](auto K) {
                        constexpr int n1 = decltype(K)::value;
                        out[n1 * S1 + t] = c[brev(n1, R1)];
                    });
                }
            }
            if (t == 0) {
                cs->pm2 = __uint_as_float(unsigned(best >> 32));
                cs->pk = pk;
                if (tpl == 0) cs->sum_x2 = (float)tot[0];
                cs->sum_mag = WANT_STD ? (float)tot[1] : 0.f;
                cs->sum_mag2 = WANT_STD ? (float)tot[2] : 0.f;
            }
        }
    }
}

}  // namespace w16

// ------------------------------------------------------------------ launchers
using namespace w16;

size_t lds_bytes_16k_w16() { return LDS_BYTES; }
int table_cpx_16k_w16() { return TABLE_CPX; }

namespace {
typedef void (*carrier_fn)(const void*, int, DevCfg, const cpx*, CarStats*, cpx*);
typedef void (*correlate_fn)(const void*, DevCfg, const cpx*, const cpx*, const f4*,
                             const ShiftParams*, const int*, const int*, CorrStats*, thr_record*,
                             f4*, cpx*, cpx*, int);
#ifdef THR_DEV_MINIMAL
carrier_fn carrier_variant(int, bool, bool) { return &k_carrier_w16<THR_IN_U8, false, false>; }
correlate_fn correlate_variant(int, bool, bool, bool) {
    return &k_correlate_w16<THR_IN_U8, false, false, false>;
}
#else
template <int FMT, bool STD>
carrier_fn pick_carrier(bool dump) {
    return dump ? &k_carrier_w16<FMT, STD, true> : &k_carrier_w16<FMT, STD, false>;
}
carrier_fn carrier_variant(int fmt, bool want_std, bool dump) {
    if (fmt == THR_IN_U8)
        return want_std ? pick_carrier<THR_IN_U8, true>(dump) : pick_carrier<THR_IN_U8, false>(dump);
    return want_std ? pick_carrier<THR_IN_C64, true>(dump) : pick_carrier<THR_IN_C64, false>(dump);
}
template <int FMT, bool STD, bool MULTI>
correlate_fn pick_correlate(bool dump) {
    return dump ? &k_correlate_w16<FMT, STD, MULTI, true> : &k_correlate_w16<FMT, STD, MULTI, false>;
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
}  // namespace

hipError_t prepare_16k_w16() {
    for (const void* f : {reinterpret_cast<const void*>(&k_carrier_pruned_w16<THR_IN_U8>),
                          reinterpret_cast<const void*>(&k_carrier_pruned_w16<THR_IN_C64>)}) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int st = 0; st < 2; ++st)
            for (int d = 0; d < 2; ++d) {
                hipError_t e = hipFuncSetAttribute(
                    reinterpret_cast<const void*>(carrier_variant(fmt, st, d)),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                if (e != hipSuccess) return e;
                for (int m = 0; m < 2; ++m) {
                    e = hipFuncSetAttribute(
                        reinterpret_cast<const void*>(correlate_variant(fmt, st, m, d)),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                    if (e != hipSuccess) return e;
                }
            }
    return hipSuccess;
}

hipError_t launch_carrier_16k_w16(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                                  const float2* tables, const float2* /*twn*/, CarStats* stats,
                                  float2* dump_fft, int grid, hipStream_t stream) {
    if (cfg.car_prune == 1 && dump_fft == nullptr) {
        if (fmt == THR_IN_U8)
            hipLaunchKernelGGL(k_carrier_pruned_w16<THR_IN_U8>, dim3(grid), dim3(NT), LDS_BYTES, stream,
                               samples, n_blocks, cfg, reinterpret_cast<const cpx*>(tables), stats);
        else
            hipLaunchKernelGGL(k_carrier_pruned_w16<THR_IN_C64>, dim3(grid), dim3(NT), LDS_BYTES, stream,
                               samples, n_blocks, cfg, reinterpret_cast<const cpx*>(tables), stats);
        return hipGetLastError();
    }
    carrier_fn fn = carrier_variant(fmt, cfg.car_want_std != 0, dump_fft != nullptr);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                       reinterpret_cast<const cpx*>(tables), stats, reinterpret_cast<cpx*>(dump_fft));
    return hipGetLastError();
}

hipError_t launch_correlate_16k_w16(int fmt, const void* samples, const DevCfg& cfg,
                                    const float2* tables, const float2* twn, const float4* tspec,
                                    const ShiftParams* shifts, const int* work_list,
                                    const int* work_count, CorrStats* corr_stats,
                                    thr_record* records, float4* xhat_scratch, float2* dump_xhat,
                                    float2* dump_corr, int dump_template, int grid,
                                    hipStream_t stream) {
    const bool dump = dump_xhat != nullptr || dump_corr != nullptr;
    correlate_fn fn = correlate_variant(fmt, cfg.cor_want_std != 0, cfg.n_templates > 1, dump);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list, work_count,
                       corr_stats, records, reinterpret_cast<f4*>(xhat_scratch),
                       reinterpret_cast<cpx*>(dump_xhat), reinterpret_cast<cpx*>(dump_corr),
                       dump_template);
    return hipGetLastError();
}

}  // namespace thr
