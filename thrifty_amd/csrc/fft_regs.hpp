// In-register complex DFT building blocks for gfx950 (wave64, fp32).
//
// Complex values are clang ext-vector float2 (`cpx`), so a complex add/sub is one
// v_pk_add_f32 and a rotation by a compile-time twiddle is v_pk_mul + v_pk_fma with
// op_sel swizzles -- no register shuffling.  On CDNA4 a packed op costs the same
// VALU cycles as its two scalar halves (32 lane-ops/clk/SIMD either way), so the
// aim is simply the fewest issue slots and zero v_mov: products with run-time twiddles
// are two packed instructions (cmul below), and multiplications by -+i are folded into the
// radix-4 butterflies as a v_pk_fma with a (+-1, -+1) constant.  This translation unit must be
// compiled with -fno-slp-vectorize (the SLP vectorizer re-packs the scalar forms
// and pays for it in v_mov/v_pk_mov -- measured ~40 % extra VALU).
//
// Every array index is a compile-time constant after inlining: `cpx v[R]` lives
// entirely in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace thr {

typedef float cpx __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Run-time complex products as TWO packed instructions: the broadcast of a.x / a.y, the swap of
// b and the sign are all VOP3P modifiers (op_sel, op_sel_hi, neg_lo / neg_hi), which the compiler
// does not fold by itself (it materialises (-b.y, b.x) with v_xor + v_mov, or -- the form used
// until round 2 -- emits four scalar mul / fma).  Same roundings as the scalar form (one mul, one
// fma per component): bit-identical results.  Measured (MI355X, 2 waves per SIMD): k_correlate
// -1.3 % (-3.9 % with four templates), pruned carrier kernel -3 %.  -DTHR_SCALAR_CMUL restores
// the scalar form (A/B).
#ifndef THR_SCALAR_CMUL
__device__ __forceinline__ cpx cmul_lo(cpx a, cpx b) {   // (a.x b.x, a.x b.y)
    cpx t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ cpx cmul_hi(cpx a, cpx b, cpx t) {   // t + (-a.y b.y, a.y b.x)
    cpx r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ cpx cmulc_lo(cpx a, cpx b) {   // (a.x b.x, -a.x b.y)
    cpx t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ cpx cmulc_hi(cpx a, cpx b, cpx t) {   // t + (a.y b.y, a.y b.x)
    cpx r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return cmul_hi(a, b, cmul_lo(a, b)); }
// acc + a * b, two packed fma (the accumulation rides on the first)
__device__ __forceinline__ cpx cmul_acc(cpx acc, cpx a, cpx b) {
    cpx t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(acc));
    return cmul_hi(a, b, t);
}
// a * u with a wave-uniform u read straight from its SGPR pair (no v_mov_b64 into VGPRs first)
__device__ __forceinline__ cpx cmul_uniform(cpx a, cpx u) {
    cpx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(u));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "s"(u), "v"(t));
    return r;
}
// a * conj(b)
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) { return cmulc_hi(a, b, cmulc_lo(a, b)); }
#else
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
    cpx r;
    r.x = fmaf(-a.y, b.y, a.x * b.x);
    r.y = fmaf(a.y, b.x, a.x * b.y);
    return r;
}
// a * conj(b)
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) {
    cpx r;
    r.x = fmaf(a.y, b.y, a.x * b.x);
    r.y = fmaf(a.y, b.x, -(a.x * b.y));
    return r;
}
__device__ __forceinline__ cpx cmul_uniform(cpx a, cpx u) { return cmul(a, u); }
__device__ __forceinline__ cpx cmul_acc(cpx acc, cpx a, cpx b) { return acc + cmul(a, b); }
#endif
// two independent products.  (Interleaving them -- mul, mul, fma, fma, which saves the wait state
// a packed result costs when the very next instruction consumes it -- was measured 4 % SLOWER in
// k_correlate than the plain sequence, s_nop included.)
__device__ __forceinline__ void cmul2(cpx a0, cpx b0, cpx a1, cpx b1, cpx& r0, cpx& r1) {
    r0 = cmul(a0, b0);
    r1 = cmul(a1, b1);
}
__device__ __forceinline__ void cmulc2(cpx a0, cpx b0, cpx a1, cpx b1, cpx& r0, cpx& r1) {
    r0 = cmulc(a0, b0);
    r1 = cmulc(a1, b1);
}
__device__ __forceinline__ cpx cconj(cpx a) { return cpx{a.x, -a.y}; }
__device__ __forceinline__ float cnorm(cpx a) { return fmaf(a.x, a.x, a.y * a.y); }
// a + DIR*i*b  (DIR = +1 or -1) as ONE v_pk_fma_f32: b.yx * (-+1, +-1) + a.  The swap is an
// op_sel modifier, so the multiplication by +-i is free.  (Writing it as a scalar
// add/sub pair makes the backend re-pack it with v_mov shuffles.)
template <int DIR>
__device__ __forceinline__ cpx add_irot(cpx a, cpx b) {
    return __builtin_elementwise_fma(b.yx, DIR > 0 ? cpx{-1.0f, 1.0f} : cpx{1.0f, -1.0f}, a);
}

// cos(2*pi*j/32), j = 0..8
constexpr float kCos32[9] = {1.0f,
                             0.98078528040323044913f,
                             0.92387953251128675613f,
                             0.83146961230254523708f,
                             0.70710678118654752440f,
                             0.55557023301960222474f,
                             0.38268343236508977173f,
                             0.19509032201612826785f,
                             0.0f};

constexpr float cos32(int q) {
    q = ((q % 32) + 32) % 32;
    if (q > 16) q = 32 - q;
    return q <= 8 ? kCos32[q] : -kCos32[16 - q];
}
constexpr float sin32(int q) { return cos32(q - 8); }

// z * exp(i * DIR * 2*pi * Q / 32)
template <int Q, int DIR>
__device__ __forceinline__ cpx rot32(cpx z) {
    constexpr int q = ((Q % 32) + 32) % 32;
    if constexpr (q == 0) {
        return z;
    } else if constexpr (q == 16) {
        return -z;
    } else if constexpr (q == 8) {
        // +-i z = z.yx * (-+1, +-1): ONE v_pk_mul_f32 (the swap is an op_sel modifier) instead of
        // a v_xor and the v_movs that re-pair the halves
        return z.yx * (DIR > 0 ? cpx{-1.0f, 1.0f} : cpx{1.0f, -1.0f});
    } else if constexpr (q == 24) {
        return z.yx * (DIR > 0 ? cpx{1.0f, -1.0f} : cpx{-1.0f, 1.0f});
    } else {
        constexpr float C = cos32(q);
        constexpr float S = (DIR > 0 ? 1.0f : -1.0f) * sin32(q);
        // (x*C - y*S, x*S + y*C) = z*(C,C) + z.yx*(-S,S)
        return __builtin_elementwise_fma(z.yx, cpx{-S, S}, z * cpx{C, C});
    }
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int brev(int k, int r) {
    int out = 0;
    for (int b = 1; b < r; b <<= 1) {
        out = (out << 1) | (k & 1);
        k >>= 1;
    }
    return out;
}

// (Rounds 1-3, kept for A/B behind -DTHR_DFT_DIF; the passes use dft_reg = dft_dit below.)
// In-place decimation-in-frequency DFT of v[0..R), R in {2,4,8,16,32}, built from
// radix-4 stages (plus one radix-2 stage when log2 R is odd).
// DIR = -1: forward (exp(-i...)), +1: inverse (unnormalised).
// Output bin k ends up in v[brev(k, R)] (same placement as a radix-2 DIF cascade).
template <int R, int DIR>
__device__ __forceinline__ void dft_dif(cpx* v) {
    if constexpr (R == 2) {
        const cpx a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else if constexpr (R >= 4) {
        constexpr int H = R / 4;
        static_for<H>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int s = 32 / R;  // W_R^j == W_32^(j*s)
            const cpx a = v[j], b = v[j + H], c = v[j + 2 * H], d = v[j + 3 * H];
            const cpx t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
            v[j] = t0 + t2;                                             // k = 0 mod 4
            v[j + H] = rot32<2 * j * s, DIR>(t0 - t2);                   // k = 2 mod 4
            v[j + 2 * H] = rot32<j * s, DIR>(add_irot<DIR>(t1, t3));     // k = 1 mod 4
            v[j + 3 * H] = rot32<3 * j * s, DIR>(add_irot<-DIR>(t1, t3));  // k = 3 mod 4
        });
        if constexpr (H > 1) {
            dft_dif<H, DIR>(v);
            dft_dif<H, DIR>(v + H);
            dft_dif<H, DIR>(v + 2 * H);
            dft_dif<H, DIR>(v + 3 * H);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same DFT by decimation in TIME, multiply-add form.  A radix-4 combination of four
// sub-transform outputs S_r (r = 0..3) with twiddles w_r = W_R^(r k),
//     X[k + m H] = sum_r (W_4^m)^r w_r S_r,
// costs, with general twiddles, 12 packed instructions instead of the 14 of the
// decimation-in-frequency butterfly above (8 adds + 3 rotations of 2): the twiddle products ride
// on the adds,
//     u = S0 + w2 S2 (2 fma)      v = 2 S0 - u = S0 - w2 S2 (1 fma)
//     b = w1 S1 (2)               p = b + w3 S3 (2 fma)        q = 2 b - p (1 fma)
//     X[k] = u + p, X[k + 2H] = u - p, X[k + H] = v -+ i q, X[k + 3H] = v +- i q   (4)
// (the "2a - u" form of the second output is the Linzer-Feig / Goedecker trick; its rounding error
// is of the size of the larger output's, which is the usual FFT error model).  Same interface and
// placement as dft_dif: natural-order input, bin k in v[brev(k, R)].  In place: the
// sub-transform of inputs 4 m + r works on the stride-4 view v[r], v[r + 4], ... and leaves its
// bin k at view index brev(k, H), i.e. S_r[k] at v[4 brev(k, H) + r] -- the four slots the
// combination of k reads are the four it writes (X[k + m H] belongs at 4 brev(k, H) + brev(m, 4)).
// ---------------------------------------------------------------------------------------------
// acc + z * exp(i DIR 2 pi Q / 32)
template <int Q, int DIR>
__device__ __forceinline__ cpx rot32_acc(cpx acc, cpx z) {
    constexpr int q = ((Q % 32) + 32) % 32;
    if constexpr (q == 0) {
        return acc + z;
    } else if constexpr (q == 16) {
        return acc - z;
    } else if constexpr (q == 8) {
        return add_irot<DIR>(acc, z);
    } else if constexpr (q == 24) {
        return add_irot<-DIR>(acc, z);
    } else {
        constexpr float C = cos32(q);
        constexpr float S = (DIR > 0 ? 1.0f : -1.0f) * sin32(q);
        return __builtin_elementwise_fma(z.yx, cpx{-S, S}, __builtin_elementwise_fma(z, cpx{C, C}, acc));
    }
}
// acc - z * exp(i DIR 2 pi Q / 32)
template <int Q, int DIR>
__device__ __forceinline__ cpx rot32_sub(cpx acc, cpx z) { return rot32_acc<Q + 16, DIR>(acc, z); }

constexpr bool rot32_trivial(int Q) { return (((Q % 32) + 32) % 32) % 8 == 0; }

template <int R, int DIR, int STRIDE = 1>
__device__ __forceinline__ void dft_dit(cpx* v) {
    if constexpr (R == 2) {
        const cpx a = v[0], b = v[STRIDE];
        v[0] = a + b;
        v[STRIDE] = a - b;
    } else if constexpr (R >= 4) {
        constexpr int H = R / 4;
        if constexpr (H > 1) {
            dft_dit<H, DIR, 4 * STRIDE>(v);
            dft_dit<H, DIR, 4 * STRIDE>(v + STRIDE);
            dft_dit<H, DIR, 4 * STRIDE>(v + 2 * STRIDE);
            dft_dit<H, DIR, 4 * STRIDE>(v + 3 * STRIDE);
        }
        static_for<H>([&](auto K) {
            constexpr int k = decltype(K)::value;
            constexpr int s = 32 / R;                       // W_R^k == W_32^(k s)
            constexpr int q1 = k * s, q2 = 2 * k * s, q3 = 3 * k * s;
            cpx* base = v + 4 * STRIDE * brev(k, H);
            const cpx s0 = base[0], s1 = base[STRIDE], s2 = base[2 * STRIDE], s3 = base[3 * STRIDE];
            cpx u, w, p, q;
            if constexpr (rot32_trivial(q2)) {              // +-1, +-i: both halves are one instruction
                u = rot32_acc<q2, DIR>(s0, s2);
                w = rot32_sub<q2, DIR>(s0, s2);
            } else {
                u = rot32_acc<q2, DIR>(s0, s2);
                w = __builtin_elementwise_fma(s0, cpx{2.0f, 2.0f}, -u);
            }
            const cpx b = rot32<q1, DIR>(s1);
            if constexpr (rot32_trivial(q3)) {
                p = rot32_acc<q3, DIR>(b, s3);
                q = rot32_sub<q3, DIR>(b, s3);
            } else {
                p = rot32_acc<q3, DIR>(b, s3);
                q = __builtin_elementwise_fma(b, cpx{2.0f, 2.0f}, -p);
            }
            base[0] = u + p;                                // m = 0
            base[STRIDE] = u - p;                           // m = 2 -> slot brev(2, 4) = 1
            base[2 * STRIDE] = add_irot<DIR>(w, q);         // m = 1 -> slot 2: w + DIR i q
            base[3 * STRIDE] = add_irot<-DIR>(w, q);        // m = 3 -> slot 3
        });
    }
}

// the in-register DFT the passes use (-DTHR_DFT_DIF: the decimation-in-frequency form, A/B).
// Measured round 4 (same box, interleaved, per 32768 blocks): 1544 instead of 1656 packed
// instructions per wave and block in k_correlate; k_correlate 1.576 -> 1.559 ms, pruned carrier
// kernel 0.442 -> 0.431, full-spectrum carrier kernel 0.822 -> 0.794, four-template k_correlate
// 2.091 -> 2.042 (per 16384), N = 65536 carrier stage 0.941 -> 0.909, sectioned correlate stage
// unchanged (3.83).
template <int R, int DIR>
__device__ __forceinline__ void dft_reg(cpx* v) {
#ifdef THR_DFT_DIF
    dft_dif<R, DIR>(v);
#else
    dft_dit<R, DIR>(v);
#endif
}

}  // namespace thr
