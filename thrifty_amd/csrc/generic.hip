// Generic-length detection pipeline (any power-of-two block_len, 64 .. 2^20).
//
// Correctness-first companion of detect16k.hip for block lengths that do not have an
// LDS-resident kernel yet (BASELINE config C3: N = 65536, and the small blocks the
// reference's unit tests use).  Every stage is its own launch and round-trips complex64
// through HBM/L2 (Stockham radix-4 autosort passes), i.e. this path IS the "unfused
// pipeline" of SURVEY.md 8(d) and is HBM-bound by construction.  Same record semantics,
// same k_fit / k_finish kernels, same twiddle sources (exactly rounded root table) as the
// fast path.  Reference lines as in detect16k.hip.
#include <hip/hip_runtime.h>

#include "detect_common.hpp"

namespace thr {

typedef float cpx2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ cpx2 gmul(cpx2 a, cpx2 b) {
    return cpx2{fmaf(-a.y, b.y, a.x * b.x), fmaf(a.y, b.x, a.x * b.y)};
}

// ---- stage 0: samples -> complex64 (optionally times the shift phasor)
template <int FMT, bool SHIFT>
__global__ __launch_bounds__(256) void g_load(const void* __restrict__ samples, size_t blk_stride,
                                              int n, int log2n, int n_blocks,
                                              const cpx2* __restrict__ twn,
                                              const ShiftParams* __restrict__ shifts,
                                              const thr_record* __restrict__ records, int n_tpl,
                                              cpx2* __restrict__ out) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int b = int(gid >> log2n), i = int(gid & size_t(n - 1));
    if (b >= n_blocks) return;
    if (SHIFT && !(records[size_t(b) * n_tpl].flags & THR_FLAG_CARRIER)) return;
    cpx2 x;
    if (FMT == THR_IN_U8) {
        const uchar2 q = reinterpret_cast<const uchar2*>(
            static_cast<const unsigned char*>(samples) + size_t(b) * blk_stride)[i];
        constexpr float sc = 1.0f / 128.0f, of = -127.4f / 128.0f;
        x = cpx2{fmaf((float)q.x, sc, of), fmaf((float)q.y, sc, of)};
    } else {
        x = reinterpret_cast<const cpx2*>(static_cast<const unsigned char*>(samples) +
                                          size_t(b) * blk_stride)[i];
    }
    if (SHIFT) {
        const ShiftParams* sp = shifts + b;
        const long long q = ((long long)sp->si_mod * i) & (long long)(n - 1);
        const cpx2 wq = twn[q];
        float sn, cs;
        sincosf(6.283185307179586f * (sp->sf_over_n * float(i)), &sn, &cs);
        cpx2 p = gmul(cpx2{wq.x, -wq.y}, cpx2{cs, sn});
        p = gmul(p, cpx2{sp->c0.x, sp->c0.y});
        x = gmul(x, p);
    }
    out[gid] = x;
}

// ---- one Stockham radix-2 pass (autosort; natural order after log2 N passes)
template <bool INVERSE>
__global__ __launch_bounds__(256) void g_fft_pass(const cpx2* __restrict__ in,
                                                  cpx2* __restrict__ out, int n, int log2n,
                                                  int log2ns, int n_blocks,
                                                  const cpx2* __restrict__ twn,
                                                  const thr_record* __restrict__ records,
                                                  int n_tpl) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int half = n >> 1;
    const int b = int(gid >> (log2n - 1)), j = int(gid & size_t(half - 1));
    if (b >= n_blocks) return;
    if (records != nullptr && !(records[size_t(b) * n_tpl].flags & THR_FLAG_CARRIER)) return;
    const int ns = 1 << log2ns;
    const int k = j & (ns - 1);
    const cpx2 w0 = twn[size_t(k) << (log2n - log2ns - 1)];  // exp(-2 pi i k / (2 ns))
    const cpx2 w = INVERSE ? cpx2{w0.x, -w0.y} : w0;
    const cpx2* src = in + size_t(b) * n;
    const cpx2 a = src[j], bb = gmul(w, src[j + half]);
    cpx2* dst = out + size_t(b) * n;
    const int j0 = ((j >> log2ns) << (log2ns + 1)) + k;
    dst[j0] = a + bb;
    dst[j0 + ns] = a - bb;
}

// ---- one Stockham radix-4 pass (two radix-2 levels at once: half the HBM round trips)
template <bool INVERSE>
__global__ __launch_bounds__(256) void g_fft_pass4(const cpx2* __restrict__ in,
                                                   cpx2* __restrict__ out, int n, int log2n,
                                                   int log2ns, int n_blocks,
                                                   const cpx2* __restrict__ twn,
                                                   const thr_record* __restrict__ records,
                                                   int n_tpl) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int quarter = n >> 2;
    const int b = int(gid >> (log2n - 2)), j = int(gid & size_t(quarter - 1));
    if (b >= n_blocks) return;
    if (records != nullptr && !(records[size_t(b) * n_tpl].flags & THR_FLAG_CARRIER)) return;
    const int ns = 1 << log2ns;
    const int k = j & (ns - 1);
    // w = exp(-2 pi i k / (4 ns)) and its square / cube, each read exactly rounded from the table
    const size_t step = size_t(1) << (log2n - log2ns - 2);
    cpx2 w1 = twn[size_t(k) * step], w2 = twn[size_t(2 * k) * step], w3 = twn[size_t(3 * k) * step];
    if (INVERSE) {
        w1.y = -w1.y;
        w2.y = -w2.y;
        w3.y = -w3.y;
    }
    const cpx2* src = in + size_t(b) * n;
    const cpx2 u0 = src[j], u1 = gmul(w1, src[j + quarter]), u2 = gmul(w2, src[j + 2 * quarter]),
               u3 = gmul(w3, src[j + 3 * quarter]);
    const cpx2 t0 = u0 + u2, t1 = u0 - u2, t2 = u1 + u3, t3 = u1 - u3;
    // forward: -i * t3 = (t3.y, -t3.x); inverse: +i * t3
    const cpx2 r = INVERSE ? cpx2{-t3.y, t3.x} : cpx2{t3.y, -t3.x};
    cpx2* dst = out + size_t(b) * n;
    const int j0 = ((j >> log2ns) << (log2ns + 2)) + k;
    dst[j0] = t0 + t2;
    dst[j0 + ns] = t1 + r;
    dst[j0 + 2 * ns] = t0 - t2;
    dst[j0 + 3 * ns] = t1 - r;
}

// ---- block-wide helpers (256 threads)
__device__ __forceinline__ double blk_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ unsigned long long blk_max(unsigned long long v,
                                                       unsigned long long* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long m = sh[0];
    for (int i = 1; i < 4; ++i) m = sh[i] > m ? sh[i] : m;
    return m;
}

// ---- carrier statistics of one spectrum per workgroup (carrier_detect.py:99-154)
__global__ __launch_bounds__(256) void g_carrier_stats(const cpx2* __restrict__ spec, DevCfg cfg,
                                                       CarStats* __restrict__ stats) {
    __shared__ double shd[4];
    __shared__ unsigned long long shu[4];
    const int b = blockIdx.x, n = cfg.block_len;
    const cpx2* x = spec + size_t(b) * n;
    double s2 = 0, s1 = 0;
    unsigned long long best = 0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const float p = fmaf(x[k].x, x[k].x, x[k].y * x[k].y);
        s2 += p;
        if (cfg.car_want_std) s1 += sqrtf(p);
        const unsigned wi = unsigned(k - cfg.win_lo) & unsigned(n - 1);
        if (wi < unsigned(cfg.win_count)) {
            // |X| in the key (the reference compares float32 magnitudes; ties go to the first bin)
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(sqrtf(p)) << 32) | (0xFFFFFFFFu - wi);
            best = key > best ? key : best;
        }
    }
    s2 = blk_sum(s2, shd);
    if (cfg.car_want_std) s1 = blk_sum(s1, shd);
    best = blk_max(best, shu);
    if (threadIdx.x == 0) {
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        int peak_idx = int(wi) + cfg.win_lo;
        if (peak_idx > n) peak_idx -= n;  // sic (carrier_detect.py:151)
        CarStats st;
        st.sum_mag2 = (float)s2;
        st.sum_mag = (float)s1;
        st.peak_mag = __uint_as_float(unsigned(best >> 32));
        st.peak_idx = peak_idx;
        for (int d = 0; d < 7; ++d) {
            const cpx2 v = x[(peak_idx - 3 + d) & (n - 1)];
            st.nb[d] = sqrtf(fmaf(v.x, v.x, v.y * v.y));
        }
        st.pad = 0;
        stats[b] = st;
    }
}

// ---- X^ * conj(T)/N (natural order)
__global__ __launch_bounds__(256) void g_mult(const cpx2* __restrict__ xhat,
                                              const cpx2* __restrict__ tconj, int n, int log2n,
                                              int n_blocks, const thr_record* __restrict__ records,
                                              int n_tpl, cpx2* __restrict__ out) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int b = int(gid >> log2n), i = int(gid & size_t(n - 1));
    if (b >= n_blocks) return;
    if (!(records[size_t(b) * n_tpl].flags & THR_FLAG_CARRIER)) return;
    out[gid] = gmul(xhat[gid], tconj[i]);
}

// ---- preshift variant: roll(X, s)[k] * conj(T_bank)[k] / N, evaluated as X[k'] * Tc[(k' + s) mod N]
// (written at the ROLLED index, k = i + s: out[k] = roll(X, s)[k] * Tc_j[k], the reference's product
// term for term -- carrier_sync.py:241-245, detect_preshift.py:67-70 -- so that the correlation is
// the reference's, phase included, for the stage dumps of yield_data)
__global__ __launch_bounds__(256) void g_mult_preshift(const cpx2* __restrict__ spectrum,
                                                       const cpx2* __restrict__ bank, int n,
                                                       int log2n, int n_blocks,
                                                       const ShiftParams* __restrict__ shifts,
                                                       const thr_record* __restrict__ records,
                                                       cpx2* __restrict__ out) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int b = int(gid >> log2n), i = int(gid & size_t(n - 1));
    if (b >= n_blocks) return;
    if (!(records[b].flags & THR_FLAG_CARRIER)) return;
    const ShiftParams* sp = shifts + b;
    const int k = (i + sp->si_mod) & (n - 1);
    out[size_t(b) * n + k] = gmul(spectrum[gid], bank[size_t(sp->bank) * n + k]);
}

// ---- stage dump of the preshift variant: np.roll(FFT#1, round(shift)) (carrier_sync.py:241-245)
__global__ __launch_bounds__(256) void g_roll(const cpx2* __restrict__ spectrum, int n, int log2n,
                                              int n_blocks, const ShiftParams* __restrict__ shifts,
                                              const thr_record* __restrict__ records,
                                              cpx2* __restrict__ out) {
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int b = int(gid >> log2n), i = int(gid & size_t(n - 1));
    if (b >= n_blocks) return;
    if (!(records[b].flags & THR_FLAG_CARRIER)) return;
    out[size_t(b) * n + ((i + shifts[b].si_mod) & (n - 1))] = spectrum[gid];
}

// ---- correlation statistics of one block per workgroup (soa_estimator.py:137-143 + sums)
__global__ __launch_bounds__(256) void g_corr_stats(const cpx2* __restrict__ corr,
                                                    const cpx2* __restrict__ xhat, DevCfg cfg,
                                                    int tpl, const thr_record* __restrict__ records,
                                                    CorrStats* __restrict__ corr_stats) {
    __shared__ double shd[4];
    __shared__ unsigned long long shu[4];
    const int b = blockIdx.x, n = cfg.block_len;
    if (!(records[size_t(b) * cfg.n_templates].flags & THR_FLAG_CARRIER)) return;
    const cpx2* c = corr + size_t(b) * n;
    const cpx2* x = xhat + size_t(b) * n;
    double e2 = 0, s1 = 0, s2 = 0;
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (tpl == 0) e2 += fmaf(x[i].x, x[i].x, x[i].y * x[i].y);
        const float p = fmaf(c[i].x, c[i].x, c[i].y * c[i].y);
        if (i >= cfg.corr_lo && i < cfg.corr_hi) {
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(p) << 32) | (0xFFFFFFFFu - unsigned(i));
            best = key > best ? key : best;
        }
        if (cfg.cor_want_std && i < cfg.corr_len) {
            s2 += p;
            s1 += sqrtf(p);
        }
    }
    if (tpl == 0) e2 = blk_sum(e2, shd);
    if (cfg.cor_want_std) {
        s1 = blk_sum(s1, shd);
        s2 = blk_sum(s2, shd);
    }
    best = blk_max(best, shu);
    if (threadIdx.x == 0) {
        CorrStats* cs = corr_stats + size_t(b) * cfg.n_templates + tpl;
        const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        cs->pm2 = __uint_as_float(unsigned(best >> 32));
        cs->pk = pk;
        for (int d = 0; d < 3; ++d) {
            const int i = pk - 1 + d;
            cs->m2[d] = (i >= 0 && i < n) ? fmaf(c[i].x, c[i].x, c[i].y * c[i].y) : 0.f;
        }
        if (tpl == 0) cs->sum_x2 = (float)e2;
        cs->sum_mag = (float)s1;
        cs->sum_mag2 = (float)s2;
    }
}

inline int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

// Stockham passes (radix 4, plus one radix-2 pass when log2 N is odd), ping-ponging a <-> b;
// returns the buffer holding the natural-order result
cpx2* run_fft(cpx2* a, cpx2* b, int n, int n_blocks, bool inverse, const cpx2* twn,
              const thr_record* records, int n_tpl, hipStream_t stream, hipError_t* err) {
    const int log2n = ilog2(n);
    const dim3 blk(256);
    cpx2 *src = a, *dst = b;
    int s = 0;
    while (s < log2n) {
        if (log2n - s >= 2) {
            const size_t work = size_t(n_blocks) * (n / 4);
            const dim3 grid((unsigned)((work + 255) / 256));
            if (inverse)
                hipLaunchKernelGGL(g_fft_pass4<true>, grid, blk, 0, stream, src, dst, n, log2n, s,
                                   n_blocks, twn, records, n_tpl);
            else
                hipLaunchKernelGGL(g_fft_pass4<false>, grid, blk, 0, stream, src, dst, n, log2n, s,
                                   n_blocks, twn, records, n_tpl);
            s += 2;
        } else {
            const size_t work = size_t(n_blocks) * (n / 2);
            const dim3 grid((unsigned)((work + 255) / 256));
            if (inverse)
                hipLaunchKernelGGL(g_fft_pass<true>, grid, blk, 0, stream, src, dst, n, log2n, s,
                                   n_blocks, twn, records, n_tpl);
            else
                hipLaunchKernelGGL(g_fft_pass<false>, grid, blk, 0, stream, src, dst, n, log2n, s,
                                   n_blocks, twn, records, n_tpl);
            s += 1;
        }
        cpx2* t = src;
        src = dst;
        dst = t;
    }
    *err = hipGetLastError();
    return src;
}

}  // namespace

size_t generic_scratch_bytes(int n, int n_blocks) { return size_t(3) * n_blocks * n * sizeof(float2); }

// Carrier stage: FFT#1 + statistics.  `scratch` = 3 * n_blocks * n complex.
// On return *spectrum points at the natural-order FFT#1 inside scratch.
hipError_t generic_carrier(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                           const float2* twn, float2* scratch, CarStats* stats, float2** spectrum,
                           hipStream_t stream) {
    const int n = cfg.block_len, log2n = ilog2(n);
    cpx2* a = reinterpret_cast<cpx2*>(scratch);
    cpx2* b = a + size_t(n_blocks) * n;
    const cpx2* tw = reinterpret_cast<const cpx2*>(twn);
    const size_t total = size_t(n_blocks) * n;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    if (fmt == THR_IN_U8)
        hipLaunchKernelGGL((g_load<THR_IN_U8, false>), grid, blk, 0, stream, samples, size_t(cfg.blk_stride), n, log2n, n_blocks,
                           tw, nullptr, nullptr, 1, a);
    else
        hipLaunchKernelGGL((g_load<THR_IN_C64, false>), grid, blk, 0, stream, samples, size_t(cfg.blk_stride), n, log2n,
                           n_blocks, tw, nullptr, nullptr, 1, a);
    hipError_t e = hipSuccess;
    cpx2* res = run_fft(a, b, n, n_blocks, false, tw, nullptr, 1, stream, &e);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(g_carrier_stats, dim3(n_blocks), blk, 0, stream, res, cfg, stats);
    if (spectrum) *spectrum = reinterpret_cast<float2*>(res);
    return hipGetLastError();
}

// Correlation stage for carrier-positive blocks (flags already in `records` from k_fit).
// tspec_nat: [T][n] conj(FFT(template))/N in natural order.  If keep_xhat / keep_corr are
// non-null they receive pointers to the shifted spectrum / correlation of `dump_template`.
hipError_t generic_correlate(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                             const float2* twn, const float2* tspec_nat,
                             const ShiftParams* shifts, const thr_record* records, float2* scratch,
                             CorrStats* corr_stats, int dump_template, float2** keep_xhat,
                             float2** keep_corr, hipStream_t stream) {
    const int n = cfg.block_len, log2n = ilog2(n), T = cfg.n_templates;
    cpx2* a = reinterpret_cast<cpx2*>(scratch);
    cpx2* b = a + size_t(n_blocks) * n;
    cpx2* c = b + size_t(n_blocks) * n;
    const cpx2* tw = reinterpret_cast<const cpx2*>(twn);
    const size_t total = size_t(n_blocks) * n;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    if (fmt == THR_IN_U8)
        hipLaunchKernelGGL((g_load<THR_IN_U8, true>), grid, blk, 0, stream, samples, size_t(cfg.blk_stride), n, log2n, n_blocks,
                           tw, shifts, records, T, a);
    else
        hipLaunchKernelGGL((g_load<THR_IN_C64, true>), grid, blk, 0, stream, samples, size_t(cfg.blk_stride), n, log2n, n_blocks,
                           tw, shifts, records, T, a);
    hipError_t e = hipSuccess;
    cpx2* xhat = run_fft(a, b, n, n_blocks, false, tw, records, T, stream, &e);
    if (e != hipSuccess) return e;
    cpx2* free1 = (xhat == a) ? b : a;  // two buffers left for the inverse: free1, c
    if (keep_xhat) *keep_xhat = reinterpret_cast<float2*>(xhat);
    for (int tpl = 0; tpl < T; ++tpl) {
        // with several templates the dumped correlation must survive later iterations:
        // it is the last thing written only when tpl == dump_template is processed last
        hipLaunchKernelGGL(g_mult, grid, blk, 0, stream, xhat,
                           reinterpret_cast<const cpx2*>(tspec_nat) + size_t(tpl) * n, n, log2n,
                           n_blocks, records, T, free1);
        cpx2* corr = run_fft(free1, c, n, n_blocks, true, tw, records, T, stream, &e);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(g_corr_stats, dim3(n_blocks), blk, 0, stream, corr, xhat, cfg, tpl, records,
                           corr_stats);
        if (keep_corr && tpl == dump_template) {
            *keep_corr = reinterpret_cast<float2*>(corr);
            if (T > 1) return hipGetLastError();  // debug dump of one template: stop here
        }
    }
    return hipGetLastError();
}

hipError_t generic_preshift_correlate(int n_blocks, const DevCfg& cfg, const float2* twn,
                                      const float2* bank_nat, const ShiftParams* shifts,
                                      const thr_record* records, float2* scratch,
                                      const float2* spectrum, CorrStats* corr_stats,
                                      float2* dump_rolled, float2** keep_corr, hipStream_t stream) {
    const int n = cfg.block_len, log2n = ilog2(n);
    cpx2* a = reinterpret_cast<cpx2*>(scratch);
    cpx2* b = a + size_t(n_blocks) * n;
    cpx2* c = b + size_t(n_blocks) * n;
    const cpx2* spec = reinterpret_cast<const cpx2*>(spectrum);
    cpx2* free1 = (spec == a) ? b : a;  // generic_carrier left FFT#1 in a or b
    const cpx2* tw = reinterpret_cast<const cpx2*>(twn);
    const size_t total = size_t(n_blocks) * n;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipLaunchKernelGGL(g_mult_preshift, grid, blk, 0, stream, spec,
                       reinterpret_cast<const cpx2*>(bank_nat), n, log2n, n_blocks, shifts, records,
                       free1);
    hipError_t e = hipSuccess;
    cpx2* corr = run_fft(free1, c, n, n_blocks, true, tw, records, 1, stream, &e);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(g_corr_stats, dim3(n_blocks), blk, 0, stream, corr, spec, cfg, 0, records,
                       corr_stats);
    if (keep_corr) *keep_corr = reinterpret_cast<float2*>(corr);
    if (dump_rolled)
        hipLaunchKernelGGL(g_roll, grid, blk, 0, stream, spec, n, log2n, n_blocks, shifts, records,
                           reinterpret_cast<cpx2*>(dump_rolled));
    return hipGetLastError();
}

}  // namespace thr
