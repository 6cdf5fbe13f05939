// Dev: what would a ONE-exchange section transform cost?  k_correlate_4k transforms a 4096-sample
// section as 4 x 32 x 32: three register passes and TWO LDS exchanges per transform.
// The alternative on the books since round 5 is (2 x 32)^2 = 64 x 64: a 64-point column lives in a
// LANE PAIR (32 values each), its transform is a radix-32 butterfly in registers plus ONE radix-2
// stage across the pair (DPP quad_perm swap), and a transform needs ONE LDS exchange (the 64 x 64
// transpose).  This benchmark runs both skeletons in k_correlate_4k's geometry -- 128 threads, 35 KiB
// of LDS claimed, four workgroups per CU, every thread holds 32 complex values -- and prints the time
// per transform:
//   A  radix-4 pass (8 columns x 4, 24 twiddles) -> 16 x ds_write_b128, barrier, 32 x ds_read_b64,
//      radix-32 (5 stages), 31 twiddles, 32 x ds_write_b64 (wave-local), 16 x ds_read_b128, radix-32
//   B  radix-32, cross-lane radix-2 (per value: twiddle select, complex product, two DPP moves,
//      one packed fma), 32 twiddles, 32 x ds_write_b64, barrier, 16 x ds_read_b128, radix-32,
//      cross-lane radix-2
// (stand-in butterflies with the real instruction mix: a twiddled radix-2 stage is a two-instruction
// packed complex product and two packed adds per pair -- fft_regs.hpp's radix-32 is 226 packed
// instructions, five such stages 320, so both variants are charged alike.)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/ubench/xlane_stage.hip -o /tmp/xlane && /tmp/xlane
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NT = 128, NV = 32;
constexpr int CH = 34;                        // complex per LDS chunk (32 + 2 padding), as the kernels'
constexpr int ROW = 32 * CH;
constexpr size_t LDS_BYTES = 4 * ROW * sizeof(v2f) + 1024;   // 35 840 B: four workgroups per CU

__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f r = v2f{a.x, a.x} * w;
    return r + v2f{-a.y, a.y} * v2f{w.y, w.x};
}

template <int STAGES>
__device__ __forceinline__ void butterflies(v2f (&v)[NV], v2f w) {
#pragma unroll
    for (int s = 0; s < STAGES; ++s) {
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

__device__ __forceinline__ void twiddle(v2f (&v)[NV], const v2f (&tw)[4], int n) {
#pragma unroll
    for (int k = 1; k <= n; ++k) v[k % NV] = cmul(v[k % NV], tw[k & 3]);
}

// ---- exchanges (the kernels' address patterns)
__device__ __forceinline__ void store_b128_cols(v2f* lds, const v2f (&v)[NV], int t) {   // pass 1 -> rows
    const int m0 = t * 8;
    v2f* out = lds + (m0 >> 5) * CH + (m0 & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
            *reinterpret_cast<v4f*>(out + k1 * ROW + 2 * i) = v4f{v[k1 * 8 + 2 * i].x, v[k1 * 8 + 2 * i].y,
                                                                   v[k1 * 8 + 2 * i + 1].x, v[k1 * 8 + 2 * i + 1].y};
}
__device__ __forceinline__ void load_b64_strided(const v2f* lds, v2f (&v)[NV], int t) {   // pass 2 reads
    const volatile __attribute__((address_space(3))) v2f* base =
        (const volatile __attribute__((address_space(3))) v2f*)(lds + (t >> 5) * ROW + (t & 31));
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = base[k * CH];
}
__device__ __forceinline__ void store_b64_strided(v2f* lds, const v2f (&v)[NV], int t) {
    v2f* base = lds + (t >> 5) * ROW + (t & 31);
#pragma unroll
    for (int k = 0; k < NV; ++k) base[k * CH] = v[k];
}
__device__ __forceinline__ void load_b128_chunk(const v2f* lds, v2f (&v)[NV], int t) {     // pass 3 reads
    const v4f* base = reinterpret_cast<const v4f*>(lds + (t >> 5) * ROW + (t & 31) * CH);
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
        const v4f q = base[k];
        v[2 * k] = v2f{q.x, q.y};
        v[2 * k + 1] = v2f{q.z, q.w};
    }
}

// ---- the radix-2 stage across a lane pair: X[k] = E[k] + w^k O[k] (even lane), E[k] - w^k O[k] (odd)
__device__ __forceinline__ float swap_pair(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ void cross_lane_radix2(v2f (&v)[NV], bool odd, const v2f (&w64)[4]) {
    const v2f s = odd ? v2f{-1.f, -1.f} : v2f{1.f, 1.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        // (w_64^k: a compile-time constant in the real kernel, here one of four registers; the ODD lane's
        // values take it, the even lane's 1 -- the lanes run one instruction stream: a select per component)
        const v2f wk = w64[k & 3];
        const v2f tw = v2f{odd ? wk.x : 1.f, odd ? wk.y : 0.f};
        const v2f own = cmul(v[k], tw);
        const v2f other = v2f{swap_pair(own.x), swap_pair(own.y)};
        v[k] = __builtin_elementwise_fma(own, s, other);
    }
}

template <int VARIANT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void k_xform(float* out, int reps, float seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int t = threadIdx.x;
    v2f v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v2f{seed + 0.001f * float(t + k), seed - 0.002f * float(k)};
    const v2f w = v2f{0.999f, 0.04f + seed * 1e-6f};
    v2f tw[4], w64[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tw[k] = v2f{0.99f - 0.01f * k, 0.1f + seed * 1e-6f * k};
        w64[k] = v2f{0.98f, 0.19f + seed * 1e-6f * k};
    }
    const bool odd = t & 1;
    for (int r = 0; r < reps; ++r) {
        if constexpr (VARIANT == 0) {
            butterflies<2>(v, w);                  // radix 4 over n1 for the thread's eight columns
            twiddle(v, tw, 24);
            store_b128_cols(lds, v, t);
            __syncthreads();
            load_b64_strided(lds, v, t);
            butterflies<5>(v, w);
            twiddle(v, tw, 31);
            store_b64_strided(lds, v, t);          // rows 2w, 2w + 1 belong to wave w: no barrier
            load_b128_chunk(lds, v, t);
            butterflies<5>(v, w);
            __syncthreads();                       // (the next transform's pass-1 stores: pass C's barrier)
        } else {
            butterflies<5>(v, w);
            cross_lane_radix2(v, odd, w64);
            twiddle(v, tw, 32);
            store_b64_strided(lds, v, t);
            __syncthreads();
            load_b128_chunk(lds, v, t);
            butterflies<5>(v, w);
            cross_lane_radix2(v, odd, w64);
            __syncthreads();
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc += v[k].x + v[k].y;
    out[blockIdx.x * NT + t] = acc;
}

template <int VARIANT>
static float run(int grid, int reps, float* d_out) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_xform<VARIANT>, dim3(grid), dim3(NT), LDS_BYTES, 0, d_out, 8, 0.5f);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_xform<VARIANT>, dim3(grid), dim3(NT), LDS_BYTES, 0, d_out, reps, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int grid = 4 * prop.multiProcessorCount, reps = 2000;
    float* d_out;
    hipMalloc(&d_out, size_t(grid) * NT * sizeof(float));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_xform<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_xform<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int round = 0; round < 3; ++round) {
        const float a = run<0>(grid, reps, d_out), b = run<1>(grid, reps, d_out);
        printf("round %d: A (4 x 32 x 32, two exchanges) %.0f ns per transform and workgroup; "
               "B (64 x 64, one exchange + two cross-lane stages) %.0f ns  (B / A = %.3f)\n",
               round, a * 1e6 / reps, b * 1e6 / reps, b / a);
    }
    if (hipGetLastError() != hipSuccess) {
        printf("HIP error\n");
        return 1;
    }
    return 0;
}
