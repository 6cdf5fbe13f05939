// Dev: lds_overlap.hip showed that WAVE-LOCAL exchanges are already hidden (the SIMD's other wave
// runs meanwhile).  The kernels' exchanges are not wave-local: every one sits between two WORKGROUP
// barriers, all eight waves of the workgroup reach them in the same phase, and nobody is left to
// run while the writes drain and the first reads come back.  Would TWO blocks per workgroup,
// interleaved inside each wave (block Y's butterflies between block X's LDS traffic and its
// barriers), hide that?  Same reduced pattern: 512 threads, 136 KiB LDS image (one workgroup per
// CU), 32 complex per thread and set, one pass = cross-wave exchange (write 32 x b64 column-wise,
// barrier, read 16 x b128 row-wise, barrier) + 320 packed butterfly instructions.
//   serial     : X: write, barrier, read, barrier, butterflies; then Y the same
//   interleaved: X's and Y's exchanges take turns with the one LDS image; every wait (write drain in
//                front of a barrier, read latency behind it) has half a butterfly set of the OTHER
//                block in front of it
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NV = 32;
constexpr int ROW = 34;
constexpr size_t LDS_BYTES = 136 * 1024;

__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f r = v2f{a.x, a.x} * w;
    return r + v2f{-a.y, a.y} * v2f{w.y, w.x};
}

template <int S0, int S1>
__device__ __forceinline__ void stages(v2f (&v)[NV], v2f w) {
#pragma unroll
    for (int s = S0; s < S1; ++s) {
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

// cross-wave: value k of thread t goes to [k][t] (conflict-free b64), comes back from row t of a
// 512 x 34 image (b128) -- every thread reads what other waves wrote
__device__ __forceinline__ void lds_write(v2f* lds, const v2f (&v)[NV], int t) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k * NT + t] = v[k];
}
__device__ __forceinline__ void lds_read(const v2f* lds, v2f (&v)[NV], int t) {
    const v4f* base = reinterpret_cast<const v4f*>(lds + t * ROW);
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
        const v4f q = base[k];
        v[2 * k] = v2f{q.x, q.y};
        v[2 * k + 1] = v2f{q.z, q.w};
    }
}

#define FENCE() __builtin_amdgcn_sched_barrier(0)

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
        for (int p = 0; p < passes; ++p) {
            lds_write(lds, x, t);
            __syncthreads();
            lds_read(lds, x, t);
            __syncthreads();
            stages<0, 5>(x, w);
            lds_write(lds, y, t);
            __syncthreads();
            lds_read(lds, y, t);
            __syncthreads();
            stages<0, 5>(y, w);
        }
    } else {
        // invariant at the top: x and y hold finished passes, the LDS image is free
        for (int p = 0; p < passes; ++p) {
            lds_write(lds, x, t);
            FENCE();
            stages<0, 2>(y, w);          // ... X's writes drain under Y's first stages
            FENCE();
            __syncthreads();
            lds_read(lds, x, t);
            FENCE();
            stages<2, 5>(y, w);          // ... X's reads come back under Y's last stages
            FENCE();
            __syncthreads();             // (waits for X's reads: the image is free)
            lds_write(lds, y, t);
            FENCE();
            stages<0, 2>(x, w);
            FENCE();
            __syncthreads();
            lds_read(lds, y, t);
            FENCE();
            stages<2, 5>(x, w);
            FENCE();
            __syncthreads();
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
        printf("serial %.3f ms = %.1f ns per (set, pass); interleaved %.3f ms = %.1f ns  (%+.1f %%)\n", a,
               a * 1e6 / (2.0 * passes), b, b * 1e6 / (2.0 * passes), 100.0 * (b - a) / a);
    }
    return 0;
}
