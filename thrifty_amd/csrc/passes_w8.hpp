// Register passes of the 16 x 32 x 32 LDS-resident FFT (512 threads, 8 waves) and their
// mirrored inverse -- shared by detect16k.hip (block_len 16384) and detect_long.hip
// (block_len R0 * 16384, which runs R0 of these sub-transforms per block).
// Index algebra, LDS layout and bank-conflict reasoning: DESIGN.md section 2.
#pragma once
#include <hip/hip_runtime.h>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"

namespace thr {

namespace k16 {
constexpr int N = 16384;
constexpr int R1 = 16, R2 = 32, R3 = 32;
constexpr int S1 = R2 * R3;       // 1024
constexpr int CHUNK = 34;         // 32 complex + 2 pad (16 B) -> conflict-free strided b128
constexpr int ROW = R2 * CHUNK;   // 1088
constexpr int NT = 512;           // threads per workgroup
constexpr int DATA = R1 * ROW;    // 17408 complex
// LDS carve (complex units)
constexpr int OFF_C = DATA;             // C[32][32]  = W_1024^(a*b)
constexpr int OFF_A = OFF_C + 1024;     // A[16][32]  = W_512^(n2*k1)   [k1][n2]
constexpr int OFF_B = OFF_A + 512;      // Bt[16][32] = W_N^(m'*k1)     [k1][m']
// Scratch (1.5 KiB): [0, 512) the block reductions (2 parities x 8 waves x 32 B); the rest is
// per-kernel: k_carrier_pruned 128 window powers + 16 phasors ([512, 1152)), k_preshift its
// two neighbour powers ([512, 528)), k_correlate the dynamic work cursor ([768, 772)), the
// long-block kernels 16 + R0 item factors ([512, 672)).
constexpr int OFF_S = OFF_B + 512;
constexpr int LDS_CPX = OFF_S + 192;
constexpr size_t LDS_BYTES = size_t(LDS_CPX) * sizeof(cpx);  // 157,184 B of the CU's 163,840
}  // namespace k16

using namespace k16;

// One 8-byte LDS read that stays ONE ds_read_b64.  The backend pairs neighbouring 8-byte reads of a
// strided column into ds_read2_b64, which the LDS serves 16 lanes per cycle (8 cycles per
// wave-instruction, 128 B/clk) where two ds_read_b64 take 2 cycles each (32 lanes per cycle,
// 256 B/clk -- MI355X_MICROARCH.md, LDS table): the strided reads of passes 2 and B and their
// twiddle-table reads were 44 % of k_correlate's LDS-array cycles in the paired form.  A volatile
// access is never paired.  -DTHR_LDS_PAIRED restores the compiler's choice (A/B).
__device__ __forceinline__ cpx lds_b64(const cpx* p) {
#ifdef THR_LDS_PAIRED
    return *p;
#else
    // (explicitly in the LDS address space: address-space inference leaves volatile accesses alone,
    // and a generic volatile load is a flat_load)
    typedef const volatile __attribute__((address_space(3))) cpx* lds_cptr;
    return *(lds_cptr)(p);
#endif
}

// ---------------------------------------------------------------- LDS tables
__device__ __forceinline__ void load_tables(cpx* lds, const cpx* __restrict__ tables) {
    // 2048 complex = 16 KiB: 512 threads x 2 x float4
    const f4* src = reinterpret_cast<const f4*>(tables);
    f4* dst = reinterpret_cast<f4*>(lds + OFF_C);
    dst[threadIdx.x] = src[threadIdx.x];
    dst[threadIdx.x + NT] = src[threadIdx.x + NT];
}

// ---------------------------------------------------------------- sample load
// Per-thread raw samples of one block: for each sub-sequence n1 the two adjacent
// samples m = 2t, 2t+1 (n = n1*1024 + m).  The u8 flavour holds them in 16 VGPRs so
// the NEXT block's HBM loads can be in flight while the current block is computed;
// the c64 flavour (64 VGPRs if held) just remembers the address and loads at use.
template <int FMT>
struct RawSamples;

template <>
struct RawSamples<THR_IN_U8> {
    unsigned q[R1];
    __device__ __forceinline__ void load(const void* __restrict__ blk, int t) {
        // (block base uniform + 32-bit lane offset: the saddr form of global_load -- no 64-bit
        // VALU address arithmetic; rows 2j and 2j + 1 share an offset register, 2048 apart in the
        // instruction's immediate)
        const char* p = reinterpret_cast<const char*>(blk);
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1)
            q[n1] = *reinterpret_cast<const unsigned*>(
                p + (unsigned(t) * 4u + unsigned((n1 >> 1) * (S1 * 4))) + (n1 & 1) * (S1 * 2));
    }
    __device__ __forceinline__ void get(int n1, cpx& a, cpx& b) const {
        const unsigned w = q[n1];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;  // == (v - 127.4f) / 128 exactly
        a = cpx{fmaf(float(w & 0xffu), sc, of), fmaf(float((w >> 8) & 0xffu), sc, of)};
        b = cpx{fmaf(float((w >> 16) & 0xffu), sc, of), fmaf(float(w >> 24), sc, of)};
    }
    // the raw bytes (fwd_pass1_pre applies the quantiser's affine map after the radix-16 butterfly)
    static constexpr bool kBytes = true;
    __device__ __forceinline__ unsigned word(int n1) const { return q[n1]; }
    __device__ __forceinline__ void get_bytes(int n1, cpx& a, cpx& b) const {
        const unsigned w = q[n1];
        a = cpx{float(w & 0xffu), float((w >> 8) & 0xffu)};
        b = cpx{float((w >> 16) & 0xffu), float(w >> 24)};
    }
};

template <>
struct RawSamples<THR_IN_C64> {
    const f4* p;
    __device__ __forceinline__ void load(const void* __restrict__ blk, int t) {
        p = reinterpret_cast<const f4*>(blk) + t;
    }
    __device__ __forceinline__ void get(int n1, cpx& a, cpx& b) const {
        const f4 w = p[n1 * (S1 / 2)];
        a = cpx{w.x, w.y};
        b = cpx{w.z, w.w};
    }
    static constexpr bool kBytes = false;
};

// ------------------------------------------------------------ forward passes
// Pass 1 (radix 16 over n1, two adjacent m per thread) -> LDS.
// If PH: pre-rotate x[n1] by rpow[n1] and fold the per-m phasor p0/p1 into the twiddle.
// GTW: the twiddles come from `gtw` = W_16384^(k1 * q), [16][1024], in global memory
// (L2-resident): the twiddle of output k1 for the thread's two points is ONE coalesced 16-byte
// load instead of two LDS reads and a complex product A[k1][n2] * Bt[k1][m'].  A COMPILE-TIME
// choice: with both forms compiled in behind a run-time test of the pointer, k_correlate was
// 8.5 % slower (1.845 vs 1.69 ms per 32768 blocks).
template <bool PH, bool GTW = false, class RAW = void>
__device__ __forceinline__ void fwd_pass1(cpx* lds, const RAW& raw,
                                          const float2* __restrict__ rpow, cpx p0, cpx p1,
                                          float* energy = nullptr,
                                          const cpx* __restrict__ gtw = nullptr) {
    const int t = opaque_tid();
    cpx v0[R1], v1[R1];
    float e = 0.f;
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        raw.get(n1, v0[n1], v1[n1]);
        if (energy != nullptr) e += cnorm(v0[n1]) + cnorm(v1[n1]);  // time-domain sum |x|^2
        if constexpr (PH) {
            const cpx r = cpx{rpow[n1].x, rpow[n1].y};
            v0[n1] = cmul_uniform(v0[n1], r);   // (rpow: the same for every thread, scalar loads)
            v1[n1] = cmul_uniform(v1[n1], r);
        }
    }
    if (energy != nullptr) *energy = e;
    dft_reg<R1, -1>(v0);
    dft_reg<R1, -1>(v1);
    const int n2 = t >> 4, mp = 2 * (t & 15);
    const cpx* tA = lds + OFF_A;
    const cpx* tB = lds + OFF_B;
    f4* out = reinterpret_cast<f4*>(lds + n2 * CHUNK + mp);
    const char* gbase = reinterpret_cast<const char*>(gtw);
    static_for<R1>([&](auto K) {
        constexpr int k1 = decltype(K)::value;
        constexpr int src = brev(k1, R1);
        cpx y0 = v0[src], y1 = v1[src];
        if constexpr (k1 == 0) {
            if constexpr (PH) {
                cmul2(y0, p0, y1, p1, y0, y1);
            }
        } else {
            cpx w0, w1;
            if constexpr (GTW) {
                // (uniform row base + 32-bit lane offset: the saddr form of global_load, no
                // 64-bit VALU address arithmetic per row)
                const f4 ww = *reinterpret_cast<const f4*>(
                    gbase + (unsigned(t) * 16u + unsigned(k1 * 8192)));
                w0 = cpx{ww.x, ww.y};
                w1 = cpx{ww.z, ww.w};
            } else {
                const cpx a = tA[k1 * 32 + n2];
                const f4 bb = *reinterpret_cast<const f4*>(tB + k1 * 32 + mp);
                cmul2(a, cpx{bb.x, bb.y}, a, cpx{bb.z, bb.w}, w0, w1);
            }
            if constexpr (PH) {
                cmul2(w0, p0, w1, p1, w0, w1);
            }
            cmul2(y0, w0, y1, w1, y0, y1);
        }
        out[k1 * (ROW / 2)] = f4{y0.x, y0.y, y1.x, y1.y};
    });
}

// Pass-1 twiddles W_N^(k1 m) of a thread's two columns m = 2t, 2t+1 (k1 = 1..15): they do not
// depend on the block, so a persistent kernel WITHOUT a frequency shift forms them once (60
// VGPRs) instead of two LDS reads and a complex product per output and block.  `scale`: 1/128
// for u8 input (fwd_pass1_pre transforms the raw bytes; a power of two: exact).
__device__ __forceinline__ void pass1_twiddles(const cpx* lds, cpx (&w0)[R1], cpx (&w1)[R1],
                                               float scale = 1.0f) {
    const int t = opaque_tid();
    const int n2 = t >> 4, mp = 2 * (t & 15);
    const cpx* tA = lds + OFF_A;
    const cpx* tB = lds + OFF_B;
    w0[0] = w1[0] = cpx{scale, 0.f};
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) {
        const cpx a = tA[k1 * 32 + n2] * cpx{scale, scale};
        const f4 bb = *reinterpret_cast<const f4*>(tB + k1 * 32 + mp);
        cmul2(a, cpx{bb.x, bb.y}, a, cpx{bb.z, bb.w}, w0[k1], w1[k1]);
    }
}
template <class RAW>
constexpr float pass1_scale() { return RAW::kBytes ? 1.0f / 128.0f : 1.0f; }

// Pass 1 without a frequency shift, twiddles from registers (pass1_twiddles<scale>).
// u8 input (RAW::kBytes): the radix-16 butterfly runs on the RAW BYTES and the quantiser's affine
// map x = (u - 127.4f) / 128 comes after it -- the transform is linear: the factor 1/128 sits in
// the twiddles (pass1_twiddles' scale) and the offset lands in output k1 = 0 alone, 16 (-127.4f / 128)
// per component; 64 fewer VALU ops per thread and block than mapping every byte.  It is also the
// more exact order: the adds of the butterfly are integer arithmetic in float (sums <= 16 * 255,
// exact), the offset never meets the rotations.  Energy: sum |x|^2 of the thread's 64 bytes
// from two integer dot products per word (v_dot4_u32_u8: sum u^2 and sum u, exact) and
// (S2 - 2 c S1 + 64 c^2) / 128^2 in double.
template <class RAW>
__device__ __forceinline__ void fwd_pass1_pre(cpx* lds, const RAW& raw, const cpx (&w0)[R1],
                                              const cpx (&w1)[R1], float* energy = nullptr) {
    const int t = opaque_tid();
    cpx v0[R1], v1[R1];
    if constexpr (RAW::kBytes) {
        unsigned s1 = 0, s2 = 0;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            raw.get_bytes(n1, v0[n1], v1[n1]);
            if (energy != nullptr) {
                const unsigned w = raw.word(n1);
                s2 = __builtin_amdgcn_udot4(w, w, s2, false);
                s1 = __builtin_amdgcn_udot4(w, 0x01010101u, s1, false);
            }
        }
        if (energy != nullptr) {
            constexpr double c = double(127.4f);
            *energy = float((double(s2) - 2.0 * c * double(s1) + 4.0 * R1 * c * c) * (1.0 / 16384.0));
        }
    } else {
        float e = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            raw.get(n1, v0[n1], v1[n1]);
            if (energy != nullptr) e += cnorm(v0[n1]) + cnorm(v1[n1]);  // time-domain sum |x|^2
        }
        if (energy != nullptr) *energy = e;
    }
    dft_reg<R1, -1>(v0);
    dft_reg<R1, -1>(v1);
    const int n2 = t >> 4, mp = 2 * (t & 15);
    f4* out = reinterpret_cast<f4*>(lds + n2 * CHUNK + mp);
    static_for<R1>([&](auto K) {
        constexpr int k1 = decltype(K)::value;
        constexpr int src = brev(k1, R1);
        cpx y0 = v0[src], y1 = v1[src];
        if constexpr (k1 != 0) {
            cmul2(y0, w0[k1], y1, w1[k1], y0, y1);
        } else if constexpr (RAW::kBytes) {
            constexpr float sc = 1.0f / 128.0f, of16 = R1 * (-127.4f / 128.0f);
            y0 = __builtin_elementwise_fma(y0, cpx{sc, sc}, cpx{of16, of16});
            y1 = __builtin_elementwise_fma(y1, cpx{sc, sc}, cpx{of16, of16});
        }
        out[k1 * (ROW / 2)] = f4{y0.x, y0.y, y1.x, y1.y};
    });
}

// Pass 2 (radix 32 over n2, in place) -- thread (k1 = t>>5, m' = t&31).
// KEEP < 32: only outputs k2 < KEEP are written back (pruned FFT: bins k2 >= KEEP unused).
template <int KEEP = R2>
__device__ __forceinline__ void fwd_pass2(cpx* lds) {
    const int t = opaque_tid();
    const int k1 = t >> 5, mp = t & 31;
    cpx* base = lds + k1 * ROW + mp;
    const cpx* tC = lds + OFF_C + mp;
    cpx v[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) v[n2] = lds_b64(base + n2 * CHUNK);
    dft_reg<R2, -1>(v);
    static_assert(KEEP % 2 == 0, "outputs are twiddled in pairs");
    static_for<KEEP / 2>([&](auto K) {
        constexpr int k2 = 2 * decltype(K)::value;
        cpx y0 = v[brev(k2, R2)], y1 = v[brev(k2 + 1, R2)];
        if constexpr (k2 != 0)
            cmul2(y0, lds_b64(tC + k2 * 32), y1, lds_b64(tC + (k2 + 1) * 32), y0, y1);
        else
            y1 = cmul(y1, lds_b64(tC + 32));
        base[k2 * CHUNK] = y0;
        base[(k2 + 1) * CHUNK] = y1;
    });
}

// Pass 3 (radix 32 over m', registers only) -- thread (k1 = t>>5, k2 = t&31).
// On return bin k = k1 + 16*k2 + 512*k3 is in v[brev(k3, 32)].
__device__ __forceinline__ void fwd_pass3(const cpx* lds, cpx* v) {
    const int t = opaque_tid();
    const f4* src = reinterpret_cast<const f4*>(lds + (t >> 5) * ROW + (t & 31) * CHUNK);
#pragma unroll
    for (int j = 0; j < R3 / 2; ++j) {
        const f4 q = src[j];
        v[2 * j] = cpx{q.x, q.y};
        v[2 * j + 1] = cpx{q.z, q.w};
    }
    dft_reg<R3, -1>(v);
}

// ------------------------------------------------------------ inverse passes
// Pass A (radix 32 over k3, registers) then twiddle conj(W_1024^(n3*k2)) -> LDS.
// Input: z[brev(k3)] = Z[k1,k2,k3] (same placement fwd_pass3 produces).
__device__ __forceinline__ void inv_passA(cpx* lds, cpx* z) {
    const int t = opaque_tid();
    const int k2 = t & 31;
    // z is indexed by brev(k3); a DIF butterfly wants natural order input. Re-label:
    cpx v[R3];
    static_for<R3>([&](auto K) {
        constexpr int k3 = decltype(K)::value;
        v[k3] = z[brev(k3, R3)];
    });
    dft_reg<R3, +1>(v);
    const cpx* tC = lds + OFF_C + k2;
    f4* dst = reinterpret_cast<f4*>(lds + (t >> 5) * ROW + k2 * CHUNK);
    static_for<R3 / 2>([&](auto J) {
        constexpr int j = decltype(J)::value;
        cpx y0 = v[brev(2 * j, R3)], y1 = v[brev(2 * j + 1, R3)];
        if constexpr (j != 0)
            cmulc2(y0, lds_b64(tC + (2 * j) * 32), y1, lds_b64(tC + (2 * j + 1) * 32), y0, y1);
        else
            y1 = cmulc(y1, lds_b64(tC + 32));
        dst[j] = f4{y0.x, y0.y, y1.x, y1.y};
    });
}

// Pass B (radix 32 over k2, in place) -- thread (k1 = t>>5, n3 = t&31);
// twiddle conj(W_N^(k1*(32*n2 + n3))) = conj(A[k1][n2] * Bt[k1][n3]).
// tw_row >= 0: row of the gtw table to use instead of k1 (block_len R1 * 1024 < 16384: LDS row
// r holds sub-sequence k1 = r mod R1 of one of the 16 / R1 blocks, whose twiddle
// W_N^(k1 q) = W_16384^(k1 (16 / R1) q) is table row k1 * 16 / R1 -- detect_small.hip)
// GTW_LATE (with GTW): the table twiddles are requested in two halves AFTER the butterfly -- 32
// instead of 64 more live VGPRs, for kernels that keep the spectrum live beside this pass (several
// templates); the second half's L2 latency is exposed once per call.
template <bool GTW = false, bool GTW_LATE = false>
__device__ __forceinline__ void inv_passB(cpx* lds, const cpx* __restrict__ gtw = nullptr,
                                          int tw_row = -1) {
    const int t = opaque_tid();
    const int k1 = t >> 5, n3 = t & 31;
    cpx* base = lds + k1 * ROW + n3;
    cpx v[R2];
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k2] = lds_b64(base + k2 * CHUNK);
    // GTW: twiddles W_N^(k1 (32 n2 + n3)) from the L2-resident table; one template: out of its PAIR half
    // (behind the 16 x 1024 entries: [(row * 16 + j) * 32 + n3] = (n2 = 2 j, n2 = 2 j + 1), handle.hip) --
    // sixteen 16-byte loads per thread instead of thirty-two 8-byte ones
    if constexpr (GTW && GTW_LATE) {
        // (uniform base + 32-bit lane offset: with a 64-bit address held per lane across the butterfly
        // these kernels, which sit at 256 VGPRs, spill)
        const char* tw = reinterpret_cast<const char*>(gtw + 16 * 1024);
        const unsigned off = unsigned(((tw_row >= 0 ? tw_row : k1) * 16) * 32 + n3) * 16u;
        f4 w[R2 / 4];
        static_for<R2 / 4>([&](auto J) {
            w[decltype(J)::value] = *reinterpret_cast<const f4*>(tw + (off + unsigned(decltype(J)::value * 512)));
        });
        dft_reg<R2, +1>(v);
        static_for<2>([&](auto H) {
            constexpr int h = decltype(H)::value;
            static_for<R2 / 4>([&](auto K) {
                constexpr int jj = decltype(K)::value, n2 = h * (R2 / 2) + 2 * jj;
                cpx y0, y1;
                cmulc2(v[brev(n2, R2)], cpx{w[jj].x, w[jj].y}, v[brev(n2 + 1, R2)], cpx{w[jj].z, w[jj].w}, y0, y1);
                base[n2 * CHUNK] = y0;
                base[(n2 + 1) * CHUNK] = y1;
            });
            if constexpr (h == 0) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<R2 / 4>([&](auto J) {
                    w[decltype(J)::value] =
                        *reinterpret_cast<const f4*>(tw + (off + unsigned((R2 / 4 + decltype(J)::value) * 512)));
                });
            }
        });
        return;
    }
    if constexpr (GTW) {
        // issued before the butterfly, consumed after it
        const char* tw = reinterpret_cast<const char*>(gtw + 16 * 1024);
        const unsigned off = unsigned(((tw_row >= 0 ? tw_row : k1) * 16) * 32 + n3) * 16u;
        f4 w[R2 / 2];
        static_for<R2 / 2>([&](auto J) {
            w[decltype(J)::value] = *reinterpret_cast<const f4*>(tw + (off + unsigned(decltype(J)::value * 512)));
        });
        dft_reg<R2, +1>(v);
        static_for<R2 / 2>([&](auto K) {
            constexpr int j = decltype(K)::value, n2 = 2 * j;
            cpx y0, y1;
            cmulc2(v[brev(n2, R2)], cpx{w[j].x, w[j].y}, v[brev(n2 + 1, R2)], cpx{w[j].z, w[j].w}, y0, y1);
            base[n2 * CHUNK] = y0;
            base[(n2 + 1) * CHUNK] = y1;
        });
        return;
    }
    dft_reg<R2, +1>(v);
    const cpx b = lds[OFF_B + k1 * 32 + n3];
    const cpx* tA = lds + OFF_A + k1 * 32;
    static_for<R2>([&](auto K) {
        constexpr int n2 = decltype(K)::value;
        cpx y = v[brev(n2, R2)];
        cpx w = b;
        if constexpr (n2 != 0) w = cmul(lds_b64(tA + n2), b);
        base[n2 * CHUNK] = cmulc(y, w);
    });
}

// Pass C (radix 16 over k1, two adjacent m per thread), registers out:
// c0[brev(n1)] = corr[n1*1024 + 2t], c1[...] = corr[n1*1024 + 2t + 1].
__device__ __forceinline__ void inv_passC(const cpx* lds, cpx* c0, cpx* c1) {
    const int t = opaque_tid();
    const int n2 = t >> 4, mp = 2 * (t & 15);
    const f4* src = reinterpret_cast<const f4*>(lds + n2 * CHUNK + mp);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        const f4 q = src[k1 * (ROW / 2)];
        c0[k1] = cpx{q.x, q.y};
        c1[k1] = cpx{q.z, q.w};
    }
    dft_reg<R1, +1>(c0);
    dft_reg<R1, +1>(c1);
}

}  // namespace thr
