// dev check: where does ds_write_addtid_b32 write?  (address = M0[15:0] + offset + 4 * lane ?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int m0v, int extra) {
    extern __shared__ int smem[];
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) smem[i] = -1;
    __syncthreads();
    const unsigned m0 = __builtin_amdgcn_readfirstlane(unsigned(m0v) + (threadIdx.x >> 6) * 256u);
    int v = 1000 + threadIdx.x;
    if (extra)
        asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:65000\n\ts_waitcnt lgkmcnt(0)" ::"v"(v), "s"(m0) : "memory");
    else
        asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:1024\n\ts_waitcnt lgkmcnt(0)" ::"v"(v), "s"(m0) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) out[i] = smem[i];
}
int main() {
    int *d, h[36864];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4);
    for (int m0v : {0, 60000, 70000, 100000, 131072, 63488}) {
        const int extra = m0v == 63488;
        k<<<1, 128, 36864 * 4>>>(d, m0v, extra);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int first = -1, cnt = 0, last = -1;
        for (int i = 0; i < 36864; ++i) if (h[i] != -1) { if (first < 0) first = i; last = i; ++cnt; }
        printf("m0=%d: %d dwords written, first dword index %d (value %d), last %d (value %d); expect first %d\n",
               m0v, cnt, first, first >= 0 ? h[first] : 0, last, last >= 0 ? h[last] : 0, (m0v + (extra ? 65000 : 1024)) / 4);
    }
    return 0;
}
