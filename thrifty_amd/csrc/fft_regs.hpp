// In-register complex DFT building blocks for gfx950 (wave64, fp32).
//
// Every array index below is a compile-time constant after inlining, so the
// `cpx v[R]` working sets live entirely in VGPRs.  Twiddles internal to a
// butterfly are compile-time constants (multiples of 2*pi/32); multiplications
// by +-1 and +-i are eliminated at compile time.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace thr {

typedef float2 cpx;

__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return cpx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return cpx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
    return cpx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// a * conj(b)
__device__ __forceinline__ cpx cmulc(cpx a, cpx b) {
    return cpx{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
__device__ __forceinline__ cpx cconj(cpx a) { return cpx{a.x, -a.y}; }
__device__ __forceinline__ float cnorm(cpx a) { return a.x * a.x + a.y * a.y; }

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
        return cpx{-z.x, -z.y};
    } else if constexpr (q == 8) {
        return DIR > 0 ? cpx{-z.y, z.x} : cpx{z.y, -z.x};
    } else if constexpr (q == 24) {
        return DIR > 0 ? cpx{z.y, -z.x} : cpx{-z.y, z.x};
    } else {
        constexpr float C = cos32(q);
        constexpr float S = (DIR > 0 ? 1.0f : -1.0f) * sin32(q);
        return cpx{z.x * C - z.y * S, z.x * S + z.y * C};
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

// In-place radix-2 decimation-in-frequency DFT of v[0..R), R | 32.
// DIR = -1: forward (exp(-i...)), +1: inverse (unnormalised).
// Output bin k ends up in v[brev(k, R)].
template <int R, int DIR>
__device__ __forceinline__ void dft_dif(cpx* v) {
    if constexpr (R >= 2) {
        constexpr int H = R / 2;
        static_for<H>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const cpx a = v[j], b = v[j + H];
            v[j] = cadd(a, b);
            v[j + H] = rot32<j*(32 / R), DIR>(csub(a, b));
        });
        dft_dif<H, DIR>(v);
        dft_dif<H, DIR>(v + H);
    }
}

}  // namespace thr
