// dev check: v_max_f64 on denormal-double bit patterns keeps every bit (the peak-key reduction
// of k_correlate relies on it: kernel_util.hpp max_power_key).  hipcc --offload-arch=gfx950 -o t t.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned long long* in, unsigned long long* out) {
    unsigned long long a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1], r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    out[threadIdx.x] = r;
}
int main() {
    unsigned long long h[8] = {0x00000000FFFFFFF0ull, 0x00000000FFFFFFF5ull,      // hi word 0 (power 0.0)
                               0x00000001FFFFFFF0ull, 0x0000000100000000ull,      // denormal floats as hi word
                               0x000FFFFF00000001ull, 0x000FFFFF00000002ull,
                               0x3F80000000000007ull, 0x3F7FFFFFFFFFFFFFull};
    unsigned long long *d, *o, r[4];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 4>>>(d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned long long want = h[2 * i] > h[2 * i + 1] ? h[2 * i] : h[2 * i + 1];
        printf("%016llx %016llx -> %016llx (%s)\n", h[2 * i], h[2 * i + 1], r[i], r[i] == want ? "ok" : "BAD");
        bad += r[i] != want;
    }
    return bad;
}
