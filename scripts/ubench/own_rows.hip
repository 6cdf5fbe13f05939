// Dev micro-benchmark: what parking intermediate rows costs -- the exchange pattern of the N = 65536
// fused kernel (detect_long.hip): every workgroup (one per CU, 512 threads) writes ROWS rows of
// 128 KiB (16 x float4 per thread and row, coalesced) to ITS OWN region of global memory and reads
// them back as the thread that wrote them, again and again.  The regions together stay inside the
// 256 MiB Infinity Cache.  Prints bytes per clock and CU for write-only, read-only and write+read,
// i.e. how long a block's parked rows occupy a CU's path to the fabric if nothing else hides them.
//   hipcc --offload-arch=gfx950 -O3 own_rows.hip -o own_rows && ./own_rows
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 write, 1 read, 2 write then read back
__global__ __launch_bounds__(512) void k(v4f* buf, int rows, int iters, unsigned long long* cycles, float* sink) {
    v4f* mine = buf + size_t(blockIdx.x) * rows * 8192;   // 8192 float4 = 128 KiB per row
    const int t = threadIdx.x;
    v4f acc = v4f{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1)
            for (int r = 0; r < rows; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) mine[(r * 16 + j) * 512 + t] = v4f{float(it), float(r), float(j), float(t)};
        if (MODE != 0)
            for (int r = 0; r < rows; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += mine[(r * 16 + j) * 512 + t];
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 1.2345f) *sink = acc.y;
}

int main() {
    const int grid = 256, iters = 200;
    unsigned long long* d_cyc;
    float* d_sink;
    hipMalloc(&d_cyc, grid * sizeof(unsigned long long));
    hipMalloc(&d_sink, 4);
    for (int rows : {3, 6}) {
        v4f* buf;
        hipMalloc(&buf, size_t(grid) * rows * 8192 * sizeof(v4f));
        hipMemset(buf, 0, size_t(grid) * rows * 8192 * sizeof(v4f));
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, buf, rows, iters, d_cyc, d_sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, buf, rows, iters, d_cyc, d_sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, buf, rows, iters, d_cyc, d_sink);
            };
            launch();
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned long long cyc[256];
            hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
            double mean = 0;
            for (int i = 0; i < grid; ++i) mean += double(cyc[i]) / grid;
            const double bytes = double(iters) * rows * 131072.0 * (mode == 2 ? 2 : 1);
            (void)mean;   // (s_memtime ticks at a fixed 100 MHz: the event time below is the measure)
            printf("rows %d (%3.0f MiB in all) %-10s: %7.1f GB/s per CU (= %.1f B/clk at 2.1 GHz), %6.2f TB/s chip, "
                   "%.0f us per block of rows\n",
                   rows, grid * rows * 0.125, mode == 0 ? "write" : mode == 1 ? "read" : "write+read",
                   bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9, bytes * grid / (ms * 1e-3) / 1e12,
                   ms * 1e3 / iters);
        }
        hipFree(buf);
    }
    return 0;
}
