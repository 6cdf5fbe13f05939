// .card ingest on the device: base64 payload -> interleaved u8 IQ.
//
// A .card line is "<timestamp> <block_idx> <base64(2N bytes)>" (reference
// block_data.py:120-131, writer fastcard_cli.c:187-192, codec fastcard/lib/base64.c).
// The host only finds the lines and the payload offsets; the 4/3-expanded text crosses
// PCIe once and is decoded straight into the sample buffer the detection kernels read.
#include <hip/hip_runtime.h>

#include "detect_common.hpp"

namespace thr {

namespace {

// 6-bit value of a base64 character; 0x40 flags '=', 0x80 flags an invalid character
__device__ __forceinline__ unsigned b64_val(unsigned c) {
    if (c - 'A' < 26u) return c - 'A';
    if (c - 'a' < 26u) return c - 'a' + 26u;
    if (c - '0' < 10u) return c - '0' + 52u;
    if (c == '+') return 62u;
    if (c == '/') return 63u;
    if (c == '=') return 0x40u;
    return 0x80u;
}

// one thread = four base64 quanta: 16 characters in (one 16-byte load; payload offsets are
// arbitrary, gfx950 global loads take unaligned addresses), 12 bytes out (three aligned dword
// stores: out_bytes and 12 are multiples of 4).  The grid is flat -- (line, chunk of the line) is
// unfolded from blockIdx.x, so the batch size is not bound by the 65535 limit of grid.y.
__global__ __launch_bounds__(256) void k_b64_decode(const unsigned char* __restrict__ text,
                                                    const long long* __restrict__ payload_off,
                                                    int n_lines, int out_bytes, int blocks_per_line,
                                                    unsigned char* __restrict__ out,
                                                    int* __restrict__ bad_lines) {
    const int line = blockIdx.x / blocks_per_line;
    const int n_quanta = (out_bytes + 2) / 3;
    const int q0 = ((blockIdx.x - line * blocks_per_line) * blockDim.x + threadIdx.x) * 4;
    if (line >= n_lines || q0 >= n_quanta) return;
    const unsigned char* src = text + payload_off[line] + size_t(q0) * 4;
    unsigned char* dst = out + size_t(line) * out_bytes + size_t(q0) * 3;
    const bool whole = q0 + 4 <= n_quanta && out_bytes - 3 * q0 >= 12 && (out_bytes & 3) == 0;
    unsigned bad = 0;
    if (whole) {
        unsigned w[4];
        __builtin_memcpy(w, src, 16);
        unsigned o[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned a = b64_val(w[j] & 0xffu), b = b64_val((w[j] >> 8) & 0xffu),
                           c = b64_val((w[j] >> 16) & 0xffu), d = b64_val(w[j] >> 24);
            bad |= (a | b | c | d) & 0xC0u;   // no padding inside a full group of quanta
            const unsigned v = (a << 18) | (b << 12) | (c << 6) | d;   // 24 bits, big-endian bytes
            // bytes 3j, 3j+1, 3j+2 of the 12-byte little-endian output
            const unsigned b0 = v >> 16, b1 = (v >> 8) & 0xffu, b2 = v & 0xffu;
            const int at = 3 * j;
            o[at >> 2] |= b0 << (8 * (at & 3));
            o[(at + 1) >> 2] |= b1 << (8 * ((at + 1) & 3));
            o[(at + 2) >> 2] |= b2 << (8 * ((at + 2) & 3));
        }
        unsigned* d32 = reinterpret_cast<unsigned*>(dst);
        d32[0] = o[0];
        d32[1] = o[1];
        d32[2] = o[2];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j;
            if (q >= n_quanta) break;
            const unsigned a = b64_val(src[4 * j]), b = b64_val(src[4 * j + 1]),
                           c = b64_val(src[4 * j + 2]), d = b64_val(src[4 * j + 3]);
            const int remaining = out_bytes - 3 * q;  // bytes this quantum must produce (1..3)
            // padding is only legal in the last quantum and only where no byte is produced
            bad |= (a | b) & 0xC0u;
            bad |= (remaining >= 2 ? c & 0xC0u : c & 0x80u);
            bad |= (remaining >= 3 ? d & 0xC0u : d & 0x80u);
            const unsigned v = ((a & 63u) << 18) | ((b & 63u) << 12) | ((c & 63u) << 6) | (d & 63u);
            dst[3 * j] = (unsigned char)(v >> 16);
            if (remaining >= 2) dst[3 * j + 1] = (unsigned char)(v >> 8);
            if (remaining >= 3) dst[3 * j + 2] = (unsigned char)v;
        }
    }
    if (bad) atomicAdd(bad_lines, 1);
}

}  // namespace

hipError_t launch_b64_decode(const unsigned char* d_text, const long long* d_payload_off, int n_lines,
                             int out_bytes, unsigned char* d_out, int* d_bad, hipStream_t stream) {
    const int n_quanta = (out_bytes + 2) / 3;
    const int threads_needed = (n_quanta + 3) / 4;
    const int blocks_per_line = (threads_needed + 255) / 256;
    if (n_lines <= 0) return hipSuccess;
    if ((long long)blocks_per_line * n_lines > 0x7fffffffLL) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(k_b64_decode, dim3(unsigned(blocks_per_line) * unsigned(n_lines)), dim3(256), 0,
                       stream, d_text, d_payload_off, n_lines, out_bytes, blocks_per_line, d_out,
                       d_bad);
    return hipGetLastError();
}

}  // namespace thr
