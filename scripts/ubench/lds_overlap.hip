// Dev: can ONE wave hide its LDS exchanges under the butterflies of a second, independent register
// set?  k_correlate's block time is ~19.4 k cycles of VALU issue + ~6 k cycles that LDS exchanges
// leave exposed; de-phasing the waves of a SIMD did not hide more.  The pattern here is
// the kernel's, reduced: 512 threads (2 waves per SIMD, 136 KiB of LDS claimed so that one workgroup
// owns the CU), every thread holds 32 complex values, one "pass" = exchange them through the
// workgroup's LDS image (wave-local rows: write 16 x b128, wait, read 16 x b128 at transposed
// addresses) + 5 radix-2 stages of twiddled butterflies on them (320 packed instructions).
//   serial     : exchange(X); butterflies(X); exchange(X); ...           -- what the kernels do
//   interleaved: two sets X, Y (128 VGPRs): read(X) is issued, butterflies(Y) run, then X is used;
//                X and Y take turns with the ONE LDS image
// Prints time per (set, pass) for both.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NV = 32;
constexpr int ROW = 34;                      // complex per LDS row (32 + 2 padding), as the kernels'
constexpr size_t LDS_BYTES = 136 * 1024;

__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    // (a.x w.x - a.y w.y, a.x w.y + a.y w.x): two packed instructions
    v2f r = v2f{a.x, a.x} * w;
    return r + v2f{-a.y, a.y} * v2f{w.y, w.x};
}

__device__ __forceinline__ void butterflies(v2f (&v)[NV], v2f w) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int h = 1 << s;
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) {
            const int i0 = (j / h) * 2 * h + (j % h), i1 = i0 + h;
            const v2f t = cmul(v[i1], w);
            const v2f a = v[i0];
            v[i0] = a + t;
            v[i1] = a - t;
        }
    }
}

// wave w owns rows 2w, 2w + 1 of 16 x 32 x 34 complex: thread writes its 32 values as a column,
// reads them back as a row (the kernels' pass 2 <-> pass 3 exchange); no workgroup barrier
__device__ __forceinline__ void lds_write(v2f* lds, const v2f (&v)[NV], int t) {
    v2f* base = lds + (t >> 5) * (32 * ROW) + (t & 31);
#pragma unroll
    for (int k = 0; k < NV; ++k) base[k * ROW] = v[k];
}
__device__ __forceinline__ void lds_read(const v2f* lds, v2f (&v)[NV], int t) {
    const v4f* base = reinterpret_cast<const v4f*>(lds + (t >> 5) * (32 * ROW) + (t & 31) * ROW);
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
        const v4f q = base[k];
        v[2 * k] = v2f{q.x, q.y};
        v[2 * k + 1] = v2f{q.z, q.w};
    }
}

template <bool INTERLEAVED>
__global__ __launch_bounds__(NT) void k(float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int t = threadIdx.x;
    v2f x[NV], y[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        x[i] = v2f{1.0f + 0.001f * float(t + i), 0.5f - 0.002f * float(i)};
        y[i] = v2f{0.7f - 0.001f * float(t + i), 0.25f + 0.001f * float(i)};
    }
    const v2f w = v2f{0.99995f, 0.01f};
    if (!INTERLEAVED) {
        for (int p = 0; p < passes; ++p) {          // two sets one after the other: same work as below
            lds_write(lds, x, t);
            lds_read(lds, x, t);
            butterflies(x, w);
            lds_write(lds, y, t);
            lds_read(lds, y, t);
            butterflies(y, w);
        }
    } else {
        lds_write(lds, x, t);
        for (int p = 0; p < passes; ++p) {
            lds_read(lds, x, t);                      // X's reads in flight ...
            butterflies(y, w);                        // ... under Y's butterflies
            __builtin_amdgcn_sched_barrier(0);
            lds_write(lds, y, t);                     // (same wave, program order: X's reads were issued first)
            lds_read(lds, y, t);                      // Y's reads in flight ...
            butterflies(x, w);                        // ... under X's butterflies
            __builtin_amdgcn_sched_barrier(0);
            lds_write(lds, x, t);
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += x[i].x + x[i].y + y[i].x + y[i].y;
    if (acc == 123.456f) out[0] = acc;
}

template <bool I>
float run(float* d, int passes) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<I>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    hipLaunchKernelGGL(k<I>, dim3(256), dim3(NT), LDS_BYTES, 0, d, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<I>, dim3(256), dim3(NT), LDS_BYTES, 0, d, passes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    const int passes = 4000;
    for (int rep = 0; rep < 3; ++rep) {
        const float a = run<false>(d, passes), b = run<true>(d, passes);
        // per (set, pass): 320 packed butterfly instructions + 16 b128 writes... per thread
        printf("serial %.3f ms = %.1f ns per (set, pass); interleaved %.3f ms = %.1f ns  (%.1f %%)\n", a,
               a * 1e6 / (2.0 * passes), b, b * 1e6 / (2.0 * passes), 100.0 * (b - a) / a);
    }
    return 0;
}
