// Dev: calibrate s_memtime ticks against wall clock, and measure sustained v_pk_fma rate vs hipEvent time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void spin(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int PK>
__global__ void flops(float* out, int iters) {
    v2f a[8]; float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = v2f{1.0f + i, 2.0f - i}; s[i] = i; }
    v2f b = {1e-3f, 2e-3f}, c = {0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(b.x));
            }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    float* f; hipMalloc(&f, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned long long ticks : {1000000ull, 10000000ull}) {
        hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, ticks); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("spin %llu ticks: %.3f ms -> %.1f MHz\n", h, ms, h / (ms * 1e3));
    }
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 4, 8}) {
            const int iters = 20000, blocks = 256 * 4;  // plenty of WGs; wps waves per SIMD via block size
            const int threads = 256;                      // 4 waves per WG
            // occupancy: launch blocks = 256 CUs * wps so that each SIMD holds `wps` waves
            const int grid = 256 * wps;
            (void)blocks;
            if (pk) hipLaunchKernelGGL(flops<1>, dim3(grid), dim3(threads), 0, 0, f, 10); else hipLaunchKernelGGL(flops<0>, dim3(grid), dim3(threads), 0, 0, f, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(flops<1>, dim3(grid), dim3(threads), 0, 0, f, iters); else hipLaunchKernelGGL(flops<0>, dim3(grid), dim3(threads), 0, 0, f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = double(iters) * 64;           // per wave
            const double flop = instr * 64 * (pk ? 4 : 2) * (grid * 4.0);
            printf("%s waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.2f ns per wave-instr\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
                   flop / (ms * 1e-3) / 1e12, ms * 1e6 / instr);
        }
    return 0;
}
