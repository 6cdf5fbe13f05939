// Dev micro-benchmark: issue rate of the VALU ops the FFT kernels are made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP 64
template <int OP>
__global__ void k(unsigned long long* out, float seed, int iters) {
    v2f a[8], b = {seed, seed * 0.5f}, c = {1.0001f, 0.9999f};
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = v2f{seed + i, seed - i}; s[i] = seed * i; }
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(b.x));
                if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
                if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(c), "v"(b));
                if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(s[i]) : "v"(b.x));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345.678f) out[1] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d) {
    const int iters = 200;
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int threads = 256 * waves_per_simd;  // one workgroup on one CU
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, 1.5f, iters);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double per_instr = double(h) / (double(iters) * REP);
        printf("%-28s waves/SIMD=%d  cycles per wave-instr (elapsed/instr)=%.2f  -> SIMD issue interval=%.2f\n",
               name, waves_per_simd, per_instr, per_instr / waves_per_simd);
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    run<0>("v_pk_fma_f32", d);
    run<1>("v_pk_add_f32", d);
    run<2>("v_pk_mul_f32", d);
    run<6>("v_pk_fma_f32 op_sel swz", d);
    run<3>("v_fma_f32", d);
    run<4>("v_add_f32", d);
    run<5>("v_mul_f32", d);
    run<7>("v_mov_b32", d);
    return 0;
}
