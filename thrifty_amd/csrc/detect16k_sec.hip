// block_len 16384 with a SHORT template: the correlate stage as overlap-save sections of 4096 samples.
//
// What the reference keeps of the correlation, ifft(X^ conj(T^))[:corr_len] with the template
// zero-padded to the block (soa_estimator.py:97-102), is a LINEAR correlation -- corr[l] =
// sum_{n < W} y[l + n] t[n], no kept lag wraps -- and of those lags the peak search reads only the
// unique window [corr_lo, corr_hi) (soa_estimator.py:20-39, 137-143) and the two lags beside the
// peak (159-170).  A 4096-sample section against a W-sample template yields 4096 - W + 1 exact
// lags of the block; BASELINE's geometry (history 4096, 1023-sample template: window [1537, 13825),
// 12288 lags) is FOUR sections of 3072 owned lags each -- 4 x (2 x 4096-point transforms) =
// 4 x 2 x 12 butterfly stages of 4096 points against 2 x 14 stages of 16384 for k_correlate: one
// seventh fewer butterflies ("fewer transforms per block, not faster ones").
//
// A section is 4096 = 4 x 32 x 32 points: FOUR rows of the 16 x 32 x 32 LDS image the 16384 kernels
// use, 34 KiB, and its two row-local radix-32 passes and their inverses (passes_w8.hpp) are the
// 16384 code unchanged; only the first forward and the last inverse pass are radix 4 instead of
// radix 16.  So a section is the work of TWO waves, and a workgroup here is 128 threads with a
// 35 KiB image: four independent workgroups per CU instead of one of eight waves in lockstep.
// Measured (MI355X, 32768 dense blocks per launch, same box, alternating): k_correlate 1.62 ms,
// this kernel 1.53 ms (-5.5 %) at 1572 instead of 2037 VALU instructions per wave and 4096 points.
// The time does not follow the instruction count alone: a CU's time per section is its VALU issue
// (about 4 k cycles) PLUS its LDS time (about 2.4 k cycles, three quarters of it the stores, which
// the LDS takes at 80 B/clk), and the LDS bytes per point are those of k_correlate; nor does it
// follow the waves' phase relation (3 resident workgroups per CU: 1.59 ms, 2: 2.11 ms) -- what
// the independent workgroups buy is that no wave ever waits for seven others.
//
// Twiddles: 35 KiB per workgroup leaves no room for the C[32][32] = W_1024^(a b) table of passes 2
// and A in LDS (8 KiB x 4).  A thread's column of it -- C[k][t & 31], k = 1 .. 31 -- is the same for
// pass 2 (thread (row, m')) and pass A (thread (row, k2)) and for every work item, so it lives in 62
// VGPRs for the life of the persistent kernel (2 waves per SIMD: 256 registers each).  Pass 1 needs
// W_4096^(k1 m) x the shift phasor of column m: three constant twiddles per thread (its first
// column) times 32 numbers per ITEM that every thread shares (the phasor steps with the twiddle
// steps folded in, formed by 32 threads one item ahead, through LDS) -- no global load in the pass.
// Pass B takes W_4096^(k1 q) = W_16384^(4 k1 q) from the L2-resident table the 16384 kernels use,
// the template spectrum is requested two passes ahead of its use, the work cursor's atomic and the
// next item's root-table entries are requested a pass before they are touched: no wave waits for a
// round trip to L2 (twelve pass-1 twiddle loads waited for one by one cost 4.6 % of the kernel).
//
// A work item is (carrier-positive block, section); it writes the windowed first-max of the lags it
// OWNS (cfg.seg_lo / seg_hi, section coordinates) to seg_stats[block][section] and k_finish keeps
// the first section holding the block's largest power (np.argmax: lowest lag) -- the machinery of
// the long blocks' sections (detect_seg.hip).  No stddev term, no stage dumps:
// every other launch of a 16384-sample block keeps k_correlate (handle.hip chooses).
//
// SEVERAL TEMPLATES (BASELINE configs[4]; one SoaEstimator per template, soa_estimator.py:78-102):
// the section is transformed ONCE and its spectrum multiplied with each template's in turn --
// 1 + T transforms of 4096 points per item.  The spectrum (64 VGPRs) and the twiddle column (62)
// do not both fit beside a pass's working set at two waves per SIMD, and a third 32 KiB array in LDS
// costs a resident workgroup.  Two forms, a compile-time choice (Q_MULTI_FORM, A/B in profiles/README.md):
//   1  the SPECTRUM stays in registers and the twiddle column is re-read for every pass that uses it
//      (passes 2 and A) from an 8 KiB pair table in global memory -- every workgroup of a CU reads the
//      same 8 KiB all the time: it stays in the CU's vector L1;
//   2  the twiddle COLUMN stays in registers and the spectrum is parked in a 32 KiB scratch slot of the
//      workgroup in global memory (written once per item, read back T - 1 times while it is still in L2).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;

namespace {

constexpr int QN = 4096;          // samples per section
constexpr int QR = 4;             // radix of the first forward / last inverse pass: 4096 = 4 x 32 x 32
constexpr int QA = 8;             // adjacent columns per thread in those passes
constexpr int QT = 128;           // threads per workgroup: two waves
constexpr int QDATA = QR * ROW;   // LDS image, complex
// scratch behind the image (bytes): [0, 128) reductions (2 parities x 2 waves x 32 B), [128, 132) the
// dynamic work cursor, [192, 192 + 98 x 8) the next item's shift-phasor factors and pass-1 twiddle steps
constexpr int Q_DYN = 128, Q_PH = 192, Q_NPH = 98;
constexpr size_t QLDS = size_t(QDATA) * sizeof(cpx) + 1024;   // 35 840 B: four workgroups per CU
#ifndef THR_Q_MULTI_FORM
#define THR_Q_MULTI_FORM 1
#endif
constexpr int Q_MULTI_FORM = THR_Q_MULTI_FORM;   // several templates: 1 spectrum in registers, 2 spectrum in scratch

// ---------------------------------------------------------------- sample source
// thread t holds, of each of the section's four sub-sequences n1, the eight adjacent samples
// m = 8 t .. 8 t + 7 (n = n1 * 1024 + m)
template <int FMT>
struct RawQ;

template <>
struct RawQ<THR_IN_U8> {
    unsigned q[QR * QA / 2];   // [n1][4] dwords: samples m0 + 2i, m0 + 2i + 1
    __device__ __forceinline__ void load(const void* __restrict__ sec, int t) {
        const unsigned char* p = static_cast<const unsigned char*>(sec) + unsigned(t) * (QA * 2);
#pragma unroll
        for (int n1 = 0; n1 < QR; ++n1)   // 16 contiguous bytes (16-byte aligned for aligned section starts)
            __builtin_memcpy(&q[n1 * (QA / 2)], p + n1 * 2048, QA * 2);
    }
    __device__ __forceinline__ void pair(int n1, int i, cpx& a, cpx& b) const {
        const unsigned w = q[n1 * (QA / 2) + i];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;  // == (v - 127.4f) / 128 exactly
        // (one packed fma per sample: same roundings as fmaf per component)
        a = __builtin_elementwise_fma(cpx{float(w & 0xffu), float((w >> 8) & 0xffu)}, cpx{sc, sc}, cpx{of, of});
        b = __builtin_elementwise_fma(cpx{float((w >> 16) & 0xffu), float(w >> 24)}, cpx{sc, sc}, cpx{of, of});
    }
};

template <>
struct RawQ<THR_IN_C64> {
    const f4* p;
    __device__ __forceinline__ void load(const void* __restrict__ sec, int t) {
        p = reinterpret_cast<const f4*>(sec) + size_t(t) * (QA / 2);
    }
    __device__ __forceinline__ void pair(int n1, int i, cpx& a, cpx& b) const {
        const f4 w = p[n1 * 512 + i];
        a = cpx{w.x, w.y};
        b = cpx{w.z, w.w};
    }
};

// ---------------------------------------------------------------- passes
// Pass 1: the frequency shift, radix 4 over n1 for the thread's 8 columns, twiddle W_4096^(k1 m)
// -> rows k1 of the image.  x[n1 * 1024 + m] is rotated by rpow[n1] (wave-uniform) before the
// butterfly; after it (the butterfly is linear in a column factor) output k1 of column m = m0 + j
// takes  p(m) W_4096^(k1 m) = [p0 W_4096^(k1 m0)] * [e[j] W_4096^(k1 j)] = qk[k1] * u[k1][j]:
// qk from the thread's three constant twiddles W_4096^(k1 m0), u -- 32 numbers per ITEM, the same
// for every thread -- from the item's table in LDS (q_phasor_table): no global load in this pass.
template <class RAW>
__device__ __forceinline__ void q_pass1(cpx* lds, const RAW& raw, int t, const float2* __restrict__ rpow,
                                        const cpx (&qk)[QR], const cpx* u) {
    const int m0 = t * QA;
    cpx* out = lds + (m0 >> 5) * CHUNK + (m0 & 31);
    static_for<QA / 2>([&](auto I) {
        constexpr int i = decltype(I)::value;
        cpx v0[QR], v1[QR];
#pragma unroll
        for (int n1 = 0; n1 < QR; ++n1) {
            raw.pair(n1, i, v0[n1], v1[n1]);
            if (n1 != 0) {   // rpow[0] == 1
                const cpx r = cpx{rpow[n1].x, rpow[n1].y};
                v0[n1] = cmul_uniform(v0[n1], r);
                v1[n1] = cmul_uniform(v1[n1], r);
            }
        }
        dft_reg<QR, -1>(v0);
        dft_reg<QR, -1>(v1);
        static_for<QR>([&](auto K) {
            constexpr int k1 = decltype(K)::value;
            const f4 uu = *reinterpret_cast<const f4*>(u + k1 * QA + 2 * i);   // u[k1][2i], u[k1][2i + 1]
            cpx w0, w1, y0, y1;
            cmul2(qk[k1], cpx{uu.x, uu.y}, qk[k1], cpx{uu.z, uu.w}, w0, w1);
            cmul2(v0[brev(k1, QR)], w0, v1[brev(k1, QR)], w1, y0, y1);
            *reinterpret_cast<f4*>(out + k1 * ROW + 2 * i) = f4{y0.x, y0.y, y1.x, y1.y};
        });
    });
}

// Pass 2 (radix 32 over n2, in place) -- thread (row = t >> 5, m' = t & 31) -- with the thread's
// column of the C table in registers: cw[k] = W_1024^(k (t & 31))
__device__ __forceinline__ void q_pass2(cpx* lds, const cpx (&cw)[R2]) {
    const int t = opaque_tid();
    cpx* base = lds + (t >> 5) * ROW + (t & 31);
    cpx v[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = lds_b64(base + n2 * CHUNK);
    dft_reg<R2, -1>(v);
    static_for<R2 / 2>([&](auto K) {
        constexpr int k2 = 2 * decltype(K)::value;
        cpx y0 = v[brev(k2, R2)], y1 = v[brev(k2 + 1, R2)];
        if constexpr (k2 != 0)
            cmul2(y0, cw[k2], y1, cw[k2 + 1], y0, y1);
        else
            y1 = cmul(y1, cw[1]);
        base[k2 * CHUNK] = y0;
        base[(k2 + 1) * CHUNK] = y1;
    });
}

// Pass A (radix 32 over k3, registers) then conj(W_1024^(n3 k2)) -> LDS; thread (row, k2 = t & 31):
// the same register column, conjugated
__device__ __forceinline__ void q_passA(cpx* lds, cpx* z, const cpx (&cw)[R2]) {
    const int t = opaque_tid();
    cpx v[R3];
    static_for<R3>([&](auto K) {
        constexpr int k3 = decltype(K)::value;
        v[k3] = z[brev(k3, R3)];
    });
    dft_reg<R3, +1>(v);
    f4* dst = reinterpret_cast<f4*>(lds + (t >> 5) * ROW + (t & 31) * CHUNK);
    static_for<R3 / 2>([&](auto J) {
        constexpr int j = decltype(J)::value;
        cpx y0 = v[brev(2 * j, R3)], y1 = v[brev(2 * j + 1, R3)];
        if constexpr (j != 0)
            cmulc2(y0, cw[2 * j], y1, cw[2 * j + 1], y0, y1);
        else
            y1 = cmulc(y1, cw[1]);
        dst[j] = f4{y0.x, y0.y, y1.x, y1.y};
    });
}

// Several templates, form 1: the same two passes with the thread's twiddle column READ for the pass
// instead of held.  ctp[j][c] = (C[2 j][c], C[2 j + 1][c]), c = t & 31: 8 KiB that every workgroup of the
// CU reads in every pass 2 and pass A (both half-waves of a wave the same 512 bytes per load).
template <int H>
__device__ __forceinline__ void q_cw_half(const f4* __restrict__ ctp, f4 (&w)[R2 / 4]) {
    const char* p = reinterpret_cast<const char*>(ctp);
    const unsigned off = unsigned(opaque_tid() & 31) * 16u;   // (uniform base + 32-bit lane offset: saddr form)
    static_for<R2 / 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        w[j] = *reinterpret_cast<const f4*>(p + (off + unsigned((H * (R2 / 4) + j) * 512)));
    });
}

__device__ __forceinline__ void q_pass2_ld(cpx* lds, const f4* __restrict__ ctp) {
    const int t = opaque_tid();
    cpx* base = lds + (t >> 5) * ROW + (t & 31);
    f4 w0[R2 / 4], w1[R2 / 4];   // requested in front of the butterfly, used behind it
    q_cw_half<0>(ctp, w0);
    q_cw_half<1>(ctp, w1);
    cpx v[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = lds_b64(base + n2 * CHUNK);
    dft_reg<R2, -1>(v);
    static_for<R2 / 2>([&](auto K) {
        constexpr int j = decltype(K)::value, k2 = 2 * j;
        const f4 w = j < R2 / 4 ? w0[j % (R2 / 4)] : w1[j % (R2 / 4)];
        cpx y0 = v[brev(k2, R2)], y1 = v[brev(k2 + 1, R2)];
        if constexpr (k2 != 0)
            cmul2(y0, cpx{w.x, w.y}, y1, cpx{w.z, w.w}, y0, y1);
        else
            y1 = cmul(y1, cpx{w.z, w.w});
        base[k2 * CHUNK] = y0;
        base[(k2 + 1) * CHUNK] = y1;
    });
}

// (the spectrum stays live beside this pass: the column comes in two halves, the first requested in
// front of the butterfly, the second behind it -- when the butterfly's temporaries are free again --
// and used after the first)
#ifndef THR_Q_HALF2_LATE
#define THR_Q_HALF2_LATE 0   // dev A/B: 1 = request the second half only when the first has been used
#endif
__device__ __forceinline__ void q_passA_ld(cpx* lds, cpx* z, const f4* __restrict__ ctp) {
    const int t = opaque_tid();
    cpx v[R3];
    static_for<R3>([&](auto K) {
        constexpr int k3 = decltype(K)::value;
        v[k3] = z[brev(k3, R3)];
    });
    f4 w[R2 / 4], w2[R2 / 4];
    q_cw_half<0>(ctp, w);
    dft_reg<R3, +1>(v);
    if constexpr (!THR_Q_HALF2_LATE) {
        __builtin_amdgcn_sched_barrier(0);
        q_cw_half<1>(ctp, w2);
    }
    f4* dst = reinterpret_cast<f4*>(lds + (t >> 5) * ROW + (t & 31) * CHUNK);
    static_for<2>([&](auto H) {
        constexpr int h = decltype(H)::value;
        static_for<R3 / 4>([&](auto J) {
            constexpr int jj = decltype(J)::value, j = h * (R3 / 4) + jj;
            const f4 ww = h == 0 ? w[jj] : w2[jj];
            cpx y0 = v[brev(2 * j, R3)], y1 = v[brev(2 * j + 1, R3)];
            if constexpr (j != 0)
                cmulc2(y0, cpx{ww.x, ww.y}, y1, cpx{ww.z, ww.w}, y0, y1);
            else
                y1 = cmulc(y1, cpx{ww.z, ww.w});
            dst[j] = f4{y0.x, y0.y, y1.x, y1.y};
        });
        if constexpr (h == 0 && THR_Q_HALF2_LATE) {
            __builtin_amdgcn_sched_barrier(0);
            q_cw_half<1>(ctp, w2);
        }
    });
}

// Pass B (radix 32 over k2, in place; thread (row, n3)) with its table twiddles W_4096^(row q) =
// W_16384^(4 row q) from the PAIR table behind the shared one (handle.hip, build_constants):
// gtwp[(4 row * 16 + j) * 32 + n3] = (W[32 (2 j) + n3], W[32 (2 j + 1) + n3]) -- sixteen 16-byte loads
// per thread and pass instead of thirty-two 8-byte ones (8-byte accesses run at 0.54-0.70 x the
// 16-byte rate, MI355X_MICROARCH.md: one template -1.9 %, four templates -3.7 %, three interleaved
// rounds each).  HALVES: the spectrum stays live beside this pass (several templates): the twiddles
// come in two halves, the second requested behind the butterfly.
#ifndef THR_Q_PAIRED_B
#define THR_Q_PAIRED_B 1   // dev A/B: 0 = the shared table, 8-byte loads
#endif
template <bool HALVES>
__device__ __forceinline__ void q_passB(cpx* lds, const f4* __restrict__ gtwp) {
    const int t = opaque_tid();
    const int k1 = t >> 5, n3 = t & 31;
    cpx* base = lds + k1 * ROW + n3;
    cpx v[R2];
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k2] = lds_b64(base + k2 * CHUNK);
    // (uniform base + 32-bit lane offset: the saddr form of global_load, no 64-bit address held per lane)
    const char* tw = reinterpret_cast<const char*>(gtwp);
    const unsigned off = unsigned(k1 * (16 / QR) * 16 * 32 + n3) * 16u;
    f4 w[R2 / 2];
    constexpr int FIRST = HALVES ? R2 / 4 : R2 / 2;
    static_for<FIRST>([&](auto J) {
        constexpr int j = decltype(J)::value;
        w[j] = *reinterpret_cast<const f4*>(tw + (off + unsigned(j * 512)));
    });
    dft_reg<R2, +1>(v);
    if constexpr (HALVES) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<R2 / 4>([&](auto J) {
            constexpr int j = R2 / 4 + decltype(J)::value;
            w[j] = *reinterpret_cast<const f4*>(tw + (off + unsigned(j * 512)));
        });
    }
    static_for<R2 / 2>([&](auto J) {
        constexpr int j = decltype(J)::value, n2 = 2 * j;
        cpx y0, y1;
        cmulc2(v[brev(n2, R2)], cpx{w[j].x, w[j].y}, v[brev(n2 + 1, R2)], cpx{w[j].z, w[j].w}, y0, y1);
        base[n2 * CHUNK] = y0;
        base[(n2 + 1) * CHUNK] = y1;
    });
}

// (dev A/B, THR_Q_PAIRED_B = 0) the same beside a live spectrum on the shared table: inv_passB<true, true>
// of passes_w8.hpp with the second half requested behind the butterfly
__device__ __forceinline__ void q_passB_ld(cpx* lds, const cpx* __restrict__ gtw, int tw_row) {
    const int t = opaque_tid();
    const int k1 = t >> 5, n3 = t & 31;
    cpx* base = lds + k1 * ROW + n3;
    cpx v[R2];
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k2] = lds_b64(base + k2 * CHUNK);
    const cpx* tw = gtw + tw_row * 1024 + n3;
    cpx w[R2 / 2], w2[R2 / 2];
#pragma unroll
    for (int n2 = 0; n2 < R2 / 2; ++n2) w[n2] = tw[n2 * 32];
    dft_reg<R2, +1>(v);
    if constexpr (!THR_Q_HALF2_LATE) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n2 = 0; n2 < R2 / 2; ++n2) w2[n2] = tw[(n2 + R2 / 2) * 32];
    }
    static_for<2>([&](auto H) {
        constexpr int h = decltype(H)::value;
        static_for<R2 / 4>([&](auto K) {
            constexpr int n2 = h * (R2 / 2) + 2 * decltype(K)::value;
            constexpr int i = n2 - h * (R2 / 2);
            cpx y0, y1;
            cmulc2(v[brev(n2, R2)], h == 0 ? w[i] : w2[i], v[brev(n2 + 1, R2)], h == 0 ? w[i + 1] : w2[i + 1], y0, y1);
            base[n2 * CHUNK] = y0;
            base[(n2 + 1) * CHUNK] = y1;
        });
        if constexpr (h == 0 && THR_Q_HALF2_LATE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n2 = 0; n2 < R2 / 2; ++n2) w2[n2] = tw[(n2 + R2 / 2) * 32];
        }
    });
}

// Pass C (radix 4 over k1, the thread's 8 columns), registers out: c[n1 * 8 + j] = corr[n1 * 1024 + 8 t + j]
__device__ __forceinline__ void q_passC(const cpx* lds, int t, cpx* c) {
    const int m0 = t * QA;
    const cpx* in = lds + (m0 >> 5) * CHUNK + (m0 & 31);
    static_for<QA / 2>([&](auto I) {
        constexpr int i = decltype(I)::value;
        cpx v0[QR], v1[QR];
#pragma unroll
        for (int k1 = 0; k1 < QR; ++k1) {
            const f4 q = *reinterpret_cast<const f4*>(in + k1 * ROW + 2 * i);
            v0[k1] = cpx{q.x, q.y};
            v1[k1] = cpx{q.z, q.w};
        }
        dft_reg<QR, +1>(v0);
        dft_reg<QR, +1>(v1);
        static_for<QR>([&](auto K) {
            constexpr int n1 = decltype(K)::value;
            c[n1 * QA + 2 * i] = v0[brev(n1, QR)];
            c[n1 * QA + 2 * i + 1] = v1[brev(n1, QR)];
        });
    });
}

// Shift phasor exp(2 pi i s (n / N - 1/2)) of block sample n = seg_start + n1 * 1024 + m, m = 8 t + j,
// t = 64 w + l:   segc0 * rpow[n1] * [exp(2 pi i s 512 w / N)] * [exp(2 pi i s 8 l / N)] * exp(2 pi i s j / N)
// -- 64 lane factors, 2 wave factors (segc0 folded in) and, with the pass-1 twiddle step folded in,
// u[k1][j] = exp(2 pi i s j / N) W_4096^(k1 j) = exp(2 pi i (s - 4 k1) j / N): 98 exactly formed
// numbers per item (integer part of the exponent through the root table, fractional part of s
// through a small-angle polynomial, |2 pi sf m| <= 2 pi (0.5 / N) 512 = 0.098 rad), formed by 98
// threads for the NEXT item.
// In two halves, so that no wave waits for a round trip to L2: the root and the item's constants are
// REQUESTED at the top of the previous item (q_phasor_fetch) and used behind its pass 1
// (q_phasor_store).
struct QPhasor {
    cpx wq;     // exp(-2 pi i q / N), q = the integer part of the exponent: the table entry, untouched (any
                // arithmetic on it here would wait for the load)
    float ang;  // 2 pi sf m
    cpx seg;    // segc0 of the item's section (wave factors only)
};
__device__ __forceinline__ QPhasor q_phasor_fetch(const ShiftParams* __restrict__ sp, int seg,
                                                  const cpx* __restrict__ twn, int t) {
    QPhasor r;
    const int tt = t < Q_NPH ? t : 0;   // (threads beyond the table fetch entry 0 and store nothing)
    const int u = tt - 66;   // >= 0: u[k1 = u >> 3][j = u & 7]
    const int m = tt < 64 ? 8 * tt : tt < 66 ? 512 * (tt - 64) : (u & 7);
    const int si = sp->si_mod - (tt < 66 ? 0 : 4 * (u >> 3));
    r.wq = twn[(si * m) & (N - 1)];
    r.ang = 6.283185307179586f * (sp->sf_over_n * float(m));
    r.seg = cpx{sp->segc0[seg].x, sp->segc0[seg].y};
    return r;
}
__device__ __forceinline__ void q_phasor_store(const QPhasor& r, int t, cpx* sc_ph) {
    if (t < Q_NPH) {
        float sn, cs;
        sincos_small(r.ang, &sn, &cs);
        cpx v = cmul(cconj(r.wq), cpx{cs, sn});
        if (t >= 64 && t < 66) v = cmul(v, r.seg);
        sc_ph[t] = v;
    }
}

// =========================================================================
// GENERIC_ROWS: every lag takes the window test (THR_PATH_GENERIC_ROWS, the cross-check form);
// otherwise an item whose section owns rows 0 .. 2 but perhaps lag 0, and of row 3 at most lag 3072
// (owned lags [0 or 1, 3072 or 3073): every section of BASELINE's geometry, every section but the
// last of any 1023-sample template) takes a peak search without the tests -- one uniform branch per item.
// =========================================================================
// MULTI: 0 one template; 1 / 2 several, the spectrum in registers / in the workgroup's scratch slot
// (Q_MULTI_FORM above).  ctp: the pair table of form 1; park: the scratch slots of form 2, 32 KiB each.
template <int FMT, bool GENERIC_ROWS, int MULTI>
__global__ __launch_bounds__(QT) __attribute__((amdgpu_waves_per_eu(2))) void k_correlate_4k(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ ctab,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec, const ShiftParams* __restrict__ shifts,
    const int* __restrict__ work_list, const int* __restrict__ work_count,
    CorrStats* __restrict__ seg_stats, const f4* __restrict__ ctp, f4* __restrict__ park) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + QDATA);
    cpx* sc_ph = reinterpret_cast<cpx*>(sc_red + Q_PH);
    int* sc_dyn = reinterpret_cast<int*>(sc_red + Q_DYN);

    const size_t blk_bytes = cfg.blk_stride;
    const int n_seg = cfg.n_seg;
    const int n_work = *work_count * n_seg;
    constexpr int kSampleBytes = FMT == THR_IN_U8 ? 2 : 8;
    const cpx* gtw = static_cast<const cpx*>(cfg.gtw);
    int parity = 0;

    // the thread's column of C = W_1024^(a b): the twiddles of passes 2 and A of every item
    cpx cw[R2];
    if constexpr (MULTI != 1) {
        const int c = opaque_tid() & 31;
        cw[0] = cpx{1.f, 0.f};
#pragma unroll
        for (int k = 1; k < R2; ++k) cw[k] = ctab[k * 32 + c];
    }
    f4* const slot = MULTI == 2 ? park + size_t(blockIdx.x) * (QN / 2) + opaque_tid() : nullptr;
    const int n_tpl = MULTI ? cfg.n_templates : 1;

    // W_4096^(k1 m0) of the thread's first column m0 = 8 t: row 4 k1 of the W_16384 table
    cpx bk[QR];
    {
        const int m0 = opaque_tid() * QA;
        bk[0] = cpx{1.f, 0.f};
#pragma unroll
        for (int k1 = 1; k1 < QR; ++k1) bk[k1] = gtw[k1 * (16 / QR) * 1024 + m0];
    }

    // work item -> (work-list entry, section); the entry's block index is LOADED where the item is
    // chosen and first touched an iteration later (no wait on the load's round trip)
    auto item_entry = [&](int wi) -> int { return n_seg == 4 ? wi >> 2 : wi / n_seg; };
    auto item_at = [&](int wi) -> int {   // block << 3 | section
        const int e = item_entry(wi);
        return (work_list[e] << 3) | (wi - e * n_seg);
    };
    auto item_samples = [&](int item) -> const unsigned char* {
        return static_cast<const unsigned char*>(samples) + size_t(item >> 3) * blk_bytes +
               size_t(cfg.seg_start[item & 7]) * kSampleBytes;
    };

    // Work distribution: a TICKET is one carrier-positive block, its n_seg sections are consecutive
    // items of one workgroup (items = ticket * n_seg + section).  A workgroup's first two tickets are
    // static (g and g + G), every later one comes from a global cursor -- one atomic per BLOCK: with
    // one per section (131072 same-address atomics per 1.5 ms launch) the kernel ran at the pace of
    // the cursor, whatever else was taken out of it.  The cursor is read by thread 0 at the top of
    // the iteration that needs it and handed round through LDS across the pass-1 barrier, two items
    // ahead of its use.  (A launch with fewer items than two per workgroup hands out single sections
    // instead -- tk = 1 -- so that a small batch still spreads over the chip.)
    const int G = int(gridDim.x);
    const int tk = n_work <= 2 * G ? 1 : n_seg;   // items per ticket
    auto last_of_ticket = [&](int wi) -> bool {
        return tk == 1 ? true : tk == 4 ? ((wi + 1) & 3) == 0 : (wi + 1) % tk == 0;
    };
    bool static_ticket = true;   // ticket g + G not handed out yet
    int wi0 = int(blockIdx.x) * tk, wi_nxt;
    if (!last_of_ticket(wi0)) {
        wi_nxt = wi0 + 1;
    } else {
        wi_nxt = (int(blockIdx.x) + G) * tk;
        static_ticket = false;
    }
    RawQ<FMT> cur;
    int it_next = wi0 < n_work ? item_at(wi0) : 0;
    int it_next2 = wi_nxt < n_work ? item_at(wi_nxt) : 0;
    int blk_next2 = it_next2 >> 3, seg_next2 = it_next2 & 7;
    if (wi0 < n_work) {
        cur.load(item_samples(it_next), opaque_tid());
        q_phasor_store(q_phasor_fetch(shifts + (it_next >> 3), it_next & 7, twn, opaque_tid()), opaque_tid(), sc_ph);
    }
    __syncthreads();
    int* dyn_ctr = const_cast<int*>(work_count) + 1;
    // the template spectrum of the thread's 32 bins k = row + 4 k2 + 128 k3 (16 x float4 from L2).  One
    // template: requested in every item two passes ahead of the product.  Several: template tpl + 1 is
    // requested in template tpl's turn, behind its peak search -- and the FIRST template's slice for the
    // next item in the last template's turn, so every request is unconditional (a conditional one would
    // keep the old slice's 64 registers alive through the whole turn).
    f4 tq[R3 / 2];
    auto request_tq = [&](int tpl) {
        const char* ts = reinterpret_cast<const char*>(tspec + size_t(tpl) * (QN / 2));
        const unsigned off = unsigned(opaque_tid()) * 16u;
        static_for<R3 / 2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            tq[j] = *reinterpret_cast<const f4*>(ts + (off + unsigned(j * (QT * 16))));
        });
    };
    if constexpr (MULTI != 0) request_tq(0);
    for (int wi = wi0, iter = 0; wi < n_work; ++iter) {
#ifdef THR_DEV
        const bool tl_on = blockIdx.x == 0 && iter == 40 && cfg.timeline != nullptr;
#endif
        THR_STAMP(0);
        const int item = it_next;
        const int b = item >> 3, seg = item & 7;
        const int t = opaque_tid();
        const ShiftParams* sp = shifts + b;
        // (the cursor's value is first touched behind pass 1: nothing waits for the atomic's round trip)
        const bool need_ticket = last_of_ticket(wi_nxt);
        int wi_dyn = 0;
#ifdef THR_Q_STATIC   // dev A/B: tickets g, g + G, g + 2 G, ... (no cursor)
        (void)dyn_ctr;
        wi_dyn = (wi_nxt + 1) / tk - 1 - G;   // the ticket after T is T + G = 2 G + wi_dyn
#else
        if (t == 0 && need_ticket && !static_ticket) wi_dyn = atomicAdd(dyn_ctr, 1);
#endif
        RawQ<FMT> nxt = cur;
        const bool more = wi_nxt < n_work;
        if (more) {
            it_next = (blk_next2 << 3) | seg_next2;
            nxt.load(item_samples(it_next), t);
        }
        // (unconditional: past the last item it re-fetches the current one's and nothing is stored)
        const QPhasor ph = q_phasor_fetch(shifts + (it_next >> 3), it_next & 7, twn, t);
        THR_STAMP(1);
        {
            cpx qk[QR];
            qk[0] = cmul(sc_ph[64 + (t >> 6)], sc_ph[t & 63]);
#pragma unroll
            for (int k1 = 1; k1 < QR; ++k1) qk[k1] = cmul(qk[0], bk[k1]);
            q_pass1(lds, cur, t, sp->rpow, qk, sc_ph + 66);
        }
        cur = nxt;
        THR_STAMP(2);
        if (t == 0)
            *sc_dyn = !need_ticket ? wi_nxt + 1
                                   : (static_ticket ? int(blockIdx.x) + G : 2 * G + wi_dyn) * tk;
        if (need_ticket) static_ticket = false;
        THR_STAMP(3);
        __syncthreads();
        THR_STAMP(4);
        const int wi_nxt2 = __builtin_amdgcn_readfirstlane(*sc_dyn);
        if (wi_nxt2 < n_work) {
            const int e = item_entry(wi_nxt2);
            blk_next2 = work_list[e];
            seg_next2 = wi_nxt2 - e * n_seg;
        }
        wi = wi_nxt;
        wi_nxt = wi_nxt2;
        // the next item's phasor factors: this item's were read before the barrier above, the new
        // ones are read after the two barriers that follow
        if (more) q_phasor_store(ph, t, sc_ph);
        if constexpr (MULTI == 0) request_tq(0);   // here: two passes ahead of the product
        // rows 2w, 2w + 1 belong to wave w through passes 2, 3, A and B: no barriers
        if constexpr (MULTI == 1)
            q_pass2_ld(lds, ctp);
        else
            q_pass2(lds, cw);
        __builtin_amdgcn_sched_barrier(0);
        THR_STAMP(5);
        // X^ conj(T^) / 4096 of the first template; with several, the product for template tpl + 1 is
        // formed at the bottom of template tpl's turn (z is what the loop carries)
        cpx z[R3];
        cpx xh[R3];
        fwd_pass3(lds, xh);
        THR_STAMP(6);
        if constexpr (MULTI == 2) {   // park the spectrum: the pairs the template's float4 j multiplies
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                slot[j * QT] = f4{xh[brev(2 * j, R3)].x, xh[brev(2 * j, R3)].y, xh[brev(2 * j + 1, R3)].x,
                                  xh[brev(2 * j + 1, R3)].y};
            });
        }
        static_for<R3 / 2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const f4 q = tq[j];
            cmul2(xh[brev(2 * j, R3)], cpx{q.x, q.y}, xh[brev(2 * j + 1, R3)], cpx{q.z, q.w},
                  z[brev(2 * j, R3)], z[brev(2 * j + 1, R3)]);
        });
      for (int tpl = 0;;) {
        if constexpr (MULTI == 1)
            q_passA_ld(lds, z, ctp);
        else
            q_passA(lds, z, cw);
        __builtin_amdgcn_sched_barrier(0);
        THR_STAMP(7);
        if constexpr (THR_Q_PAIRED_B)
            q_passB<MULTI == 1>(lds, reinterpret_cast<const f4*>(gtw + 16 * 1024));   // (the pair table follows the shared one)
        else if constexpr (MULTI == 1)
            q_passB_ld(lds, gtw, (opaque_tid() >> 5) * (16 / QR));
        else
            inv_passB<true>(lds, gtw, (opaque_tid() >> 5) * (16 / QR));
        THR_STAMP(8);
        __syncthreads();
        THR_STAMP(9);
        cpx c[QR * QA];
        q_passC(lds, opaque_tid(), c);
        THR_STAMP(10);

        // ---- |corr|^2, windowed first-max over the lags this section owns (maximum first, lag after:
        // correlate16k.hpp).  Lag n = n1 * 1024 + 8 t + j is q = n1 * 8 + j of the thread, ascending.
        const int w_lo = cfg.seg_lo[seg], w_hi = cfg.seg_hi[seg];
        const int m0 = t * QA;
        float pw[QR * QA];   // (the whole-rows search leaves row 3 beyond lag 3073 unset: never read)
        float wmax;
        unsigned lag;
        const bool whole_rows = !GENERIC_ROWS && w_lo <= 1 && w_hi >= 3 * 1024 && w_hi <= 3 * 1024 + 1;
        if (whole_rows) {
            float tmax = -1.0f;
            static_for<3 * QA / 2 + 1>([&](auto Q) {
                constexpr int q = 2 * decltype(Q)::value;
                pw[q] = cnorm(c[q]);
                pw[q + 1] = cnorm(c[q + 1]);
            });
            // lag 0 (q = 0 of thread 0) and lag 3072 (q = 24 of thread 0) are the two that may not be owned
            const float e0 = m0 >= w_lo ? pw[0] : -1.f;
            const float e24 = (m0 == 0 && w_hi > 3 * 1024) ? pw[24] : -1.f;
            tmax = __builtin_fmaxf(e0, e24);
            static_for<3 * QA / 2>([&](auto Q) {
                constexpr int q = 2 * decltype(Q)::value;
                if constexpr (q == 0)
                    tmax = __builtin_fmaxf(tmax, pw[1]);
                else
                    tmax = __builtin_fmaxf(tmax, __builtin_fmaxf(pw[q], pw[q + 1]));
            });
#pragma unroll
            for (int q = 3 * QA + 2; q < QR * QA; ++q) pw[q] = 0.f;
            wmax = wave_max_f32(tmax);
            int first = 63;   // the thread's first lag (n1 * 8 + j) that holds the wave's maximum
            first = e24 == wmax ? 24 : first;
            static_for<3 * QA - 1>([&](auto Q) {
                constexpr int q = 3 * QA - 1 - decltype(Q)::value;
                first = pw[q] == wmax ? q : first;
            });
            first = e0 == wmax ? 0 : first;
            lag = first == 63 ? 0xFFFFFFFFu : unsigned((first >> 3) * 1024 + m0 + (first & 7));
        } else {
            const unsigned win_w = unsigned(w_hi - w_lo);
            float ew[QR * QA];
            float tmax = -1.0f;
            static_for<QR * QA / 2>([&](auto Q) {
                constexpr int q = 2 * decltype(Q)::value;
                constexpr int n1 = q / QA, j = q % QA;
                pw[q] = cnorm(c[q]);
                pw[q + 1] = cnorm(c[q + 1]);
                const int n = n1 * 1024 + m0 + j;
                ew[q] = unsigned(n - w_lo) < win_w ? pw[q] : -1.f;
                ew[q + 1] = unsigned(n + 1 - w_lo) < win_w ? pw[q + 1] : -1.f;
                tmax = __builtin_fmaxf(tmax, __builtin_fmaxf(ew[q], ew[q + 1]));
            });
            wmax = wave_max_f32(tmax);
            int first = 63;
            static_for<QR * QA>([&](auto Q) {
                constexpr int q = QR * QA - 1 - decltype(Q)::value;
                first = ew[q] == wmax ? q : first;
            });
            lag = first == 63 ? 0xFFFFFFFFu : unsigned((first >> 3) * 1024 + m0 + (first & 7));
        }
        const unsigned wlag = wave_min_u32(lag);
        unsigned long long best =
            wmax < 0.f ? 0ull : ((unsigned long long)__float_as_uint(wmax) << 32) | (0xFFFFFFFFu - wlag);
        THR_STAMP(11);
        // form 2: the parked spectrum comes back in two halves -- the first requested here, under the
        // reduction, the second when the powers are dead
        f4 xs0[R3 / 4], xs1[R3 / 4];
        if constexpr (MULTI != 0) {
            // the next spectrum slice (the correlation's 64 registers are free again), used behind the
            // reduction and the neighbours
            __builtin_amdgcn_sched_barrier(0);   // (not above the correlation's arithmetic)
            request_tq(tpl + 1 < n_tpl ? tpl + 1 : 0);
#ifndef THR_Q_XS_LATE
            if constexpr (MULTI == 2)
                static_for<R3 / 4>([&](auto J) { xs0[decltype(J)::value] = slot[decltype(J)::value * QT]; });
#endif
        }
        block_reduce_wave_keys<QT / 64>(best, sc_red, parity);
        THR_STAMP(12);
        parity ^= 1;
        const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        CorrStats* cs = seg_stats + (size_t(b) * cfg.n_templates + tpl) * n_seg + seg;
        // |corr[pk - 1 .. pk + 1]|^2 for the log-parabola.  pk is uniform in the workgroup, so WHICH of
        // its 32 powers a thread would contribute, q = n1 * 8 + j of lag n = n1 * 1024 + 8 t + j, is
        // uniform too: one scalar jump picks the register, the one thread that holds the lag stores it
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int n = pk - 1 + d;
            if (n < 0 || n >= QN) {
                if (t == 0) cs->m2[d] = 0.f;   // (a peak on lag 0 of the block: k_finish does not use it)
                continue;
            }
            const int q = __builtin_amdgcn_readfirstlane((n >> 10) * QA + (n & 7));
            float val = 0.f;
            switch (q) {
#define THR_Q_CASE(Q) case Q: val = pw[Q]; break;
                THR_Q_CASE(0) THR_Q_CASE(1) THR_Q_CASE(2) THR_Q_CASE(3) THR_Q_CASE(4) THR_Q_CASE(5) THR_Q_CASE(6) THR_Q_CASE(7)
                THR_Q_CASE(8) THR_Q_CASE(9) THR_Q_CASE(10) THR_Q_CASE(11) THR_Q_CASE(12) THR_Q_CASE(13) THR_Q_CASE(14) THR_Q_CASE(15)
                THR_Q_CASE(16) THR_Q_CASE(17) THR_Q_CASE(18) THR_Q_CASE(19) THR_Q_CASE(20) THR_Q_CASE(21) THR_Q_CASE(22) THR_Q_CASE(23)
                THR_Q_CASE(24) THR_Q_CASE(25) THR_Q_CASE(26) THR_Q_CASE(27) THR_Q_CASE(28) THR_Q_CASE(29) THR_Q_CASE(30) THR_Q_CASE(31)
#undef THR_Q_CASE
            }
            if (((n & 1023) >> 3) == t) cs->m2[d] = val;
        }
        if (t == 0) {
            cs->pm2 = __uint_as_float(unsigned(best >> 32));
            cs->pk = pk;
            cs->sum_mag = 0.f;
            cs->sum_mag2 = 0.f;
        }
        THR_STAMP(13);
        if (++tpl >= n_tpl) break;
        if constexpr (MULTI == 2) {   // X^ conj(T^) / 4096 of the next template
#ifdef THR_Q_XS_LATE
            static_for<R3 / 4>([&](auto J) { xs0[decltype(J)::value] = slot[decltype(J)::value * QT]; });
#endif
            static_for<R3 / 4>([&](auto J) { xs1[decltype(J)::value] = slot[(R3 / 4 + decltype(J)::value) * QT]; });
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = tq[j];
                const f4 x = j < R3 / 4 ? xs0[j % (R3 / 4)] : xs1[j % (R3 / 4)];
                cmul2(cpx{x.x, x.y}, cpx{q.x, q.y}, cpx{x.z, x.w}, cpx{q.z, q.w},
                      z[brev(2 * j, R3)], z[brev(2 * j + 1, R3)]);
            });
        } else {
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = tq[j];
                cmul2(xh[brev(2 * j, R3)], cpx{q.x, q.y}, xh[brev(2 * j + 1, R3)], cpx{q.z, q.w},
                      z[brev(2 * j, R3)], z[brev(2 * j + 1, R3)]);
            });
        }
      }   // templates
    }
}

typedef void (*correlate4k_fn)(const void*, DevCfg, const cpx*, const cpx*, const f4*, const ShiftParams*,
                               const int*, const int*, CorrStats*, const f4*, f4*);

template <int FMT>
correlate4k_fn pick_4k(const DevCfg& cfg) {
    if (cfg.n_templates > 1)
        return cfg.no_row_geom ? &k_correlate_4k<FMT, true, Q_MULTI_FORM> : &k_correlate_4k<FMT, false, Q_MULTI_FORM>;
    return cfg.no_row_geom ? &k_correlate_4k<FMT, true, 0> : &k_correlate_4k<FMT, false, 0>;
}

}  // namespace

size_t lds_bytes_4k() { return QLDS; }
// several templates: bytes of the spectrum scratch for `grid` workgroups (0: this build keeps the spectrum in registers)
size_t park_bytes_4k(int grid) { return Q_MULTI_FORM == 2 ? size_t(grid) * QN * sizeof(cpx) : 0; }

// seg_stats: [block of the batch][template][section]; tspec4k: per template conj(FFT(template zero-padded
// to 4096)) / 4096 in the short-block kernels' lane-coalesced order; ctab_pair: C[32][32] as
// [j][c] = (C[2 j][c], C[2 j + 1][c]) (handle.hip, build_constants); park: park_bytes_4k(grid) of scratch
hipError_t launch_correlate_4k(int fmt, const void* samples, const DevCfg& cfg, const float2* tables,
                               const float2* twn, const float4* tspec4k, const ShiftParams* shifts,
                               const int* work_list, const int* work_count, CorrStats* seg_stats,
                               const float4* ctab_pair, float4* park, int grid, hipStream_t stream) {
    correlate4k_fn fn = fmt == THR_IN_U8 ? pick_4k<THR_IN_U8>(cfg) : pick_4k<THR_IN_C64>(cfg);
#ifdef THR_Q_GRIDCAP   // dev A/B: resident workgroups
    grid = std::min(grid, THR_Q_GRIDCAP);
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(QT), QLDS, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec4k), shifts, work_list, work_count, seg_stats,
                       reinterpret_cast<const f4*>(ctab_pair), reinterpret_cast<f4*>(park));
    return hipGetLastError();
}

}  // namespace thr
