// Dev micro-benchmark: what the LDS exchange of one FFT pass costs a 512-thread workgroup (8 waves,
// 2 per SIMD, one workgroup per CU as in k_correlate), per store / load form (gfx950):
//   0: 32 x ds_write_b64, row stride 34 complex (pass 2 / B as they are)
//   1: 16 x ds_write_b128 (pass 1 / A as they are)
//   2: 64 x ds_write_addtid_b32 (split re / im planes, lane-linear: the candidate form)
//   3: 32 x ds_read_b64, row stride 34 complex (pass 2 / B reads as they are)
//   4: 64 x ds_read_b32, plane stride (what form 2's layout would make of those reads)
//   5: 16 x ds_read_b128 (pass 3 / C reads)
// hipcc --offload-arch=gfx950 -O3 lds_store_rate.hip -o lds_store_rate && ./lds_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OP>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float seed, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    v2f a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = v2f{seed + i, seed - t};
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
            v2f* base = reinterpret_cast<v2f*>(smem) + (t >> 5) * 1088 + (t & 31);
#pragma unroll
            for (int i = 0; i < 32; ++i) base[i * 34] = a[i];
        } else if (OP == 1) {
            v4f* base = reinterpret_cast<v4f*>(reinterpret_cast<v2f*>(smem) + (t >> 4) * 34 + 2 * (t & 15));
#pragma unroll
            for (int i = 0; i < 16; ++i) base[i * 544] = v4f{a[2 * i].x, a[2 * i].y, a[2 * i + 1].x, a[2 * i + 1].y};
        } else if (OP == 2) {
            // plane p (32 re + 32 im planes of 516 dwords) : address = M0 + offset + 4 * lane
            const unsigned m0 = __builtin_amdgcn_readfirstlane((t >> 6) * 256);
            asm volatile("s_mov_b32 m0, %0" ::"s"(m0));
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(a[i].x), "n"(i * 2064) : "memory");
                if (i * 2064 + 66048 < 65536)
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(a[i].y), "n"((i * 2064 + 66048) & 0xffff) : "memory");
                else
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(a[i].y), "n"((i * 2064 + 512) & 0xffff) : "memory");
            }
        } else if (OP == 3) {
            const v2f* base = reinterpret_cast<const v2f*>(smem) + (t >> 5) * 1088 + (t & 31);
#pragma unroll
            for (int i = 0; i < 32; ++i) { v2f v = base[i * 34]; acc += v.x + v.y; }
        } else if (OP == 4) {
            const float* base = reinterpret_cast<const float*>(smem) + t;
#pragma unroll
            for (int i = 0; i < 64; ++i) acc += base[i * 516];
        } else if (OP == 5) {
            const v4f* base = reinterpret_cast<const v4f*>(reinterpret_cast<const v2f*>(smem) + (t >> 5) * 1088 + (t & 31) * 34);
#pragma unroll
            for (int i = 0; i < 16; ++i) { v4f v = base[i]; acc += v.x + v.w; }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345.678f) out[1] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 150 * 1024, 0, d, 1.5f, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.0f cycles per pass-exchange of one workgroup (8 waves)\n", name, double(h) / iters);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    run<0>("32 x ds_write_b64  (stride 34 complex)", d);
    run<1>("16 x ds_write_b128", d);
    run<2>("64 x ds_write_addtid_b32 (planes)", d);
    run<3>("32 x ds_read_b64   (stride 34 complex)", d);
    run<4>("64 x ds_read_b32   (planes)", d);
    run<5>("16 x ds_read_b128", d);
    return 0;
}
