// PreshiftDetector variant for block_len = 16384 (SURVEY.md 8(f) rank 2;
// reference thrifty/experimental/detect_preshift.py:24-80).
//
// The reference variant avoids FFT#2: the carrier offset comes from a 3-point parabola
// on |FFT#1| (experimental/carrier_interpolators.py:40-45), FFT#1 is *rolled* by the
// rounded shift (carrier_sync.py:241-245) and the residual sub-bin shift selects the
// nearest of `num` pre-shifted template spectra (detect_preshift.py:42-45).  With no
// iterative fit between the transform and the matched filter the whole block becomes ONE
// kernel -- one forward FFT, one inverse -- instead of pruned FFT#1 + fit + FFT#2 + IFFT:
//
//   load u8/c64 -> FFT#1 (LDS, spectrum ends in registers)
//     -> sum |X|^2, windowed first-max          (carrier_detect.py:99-154)   [reduction 1]
//     -> |X[peak-1]|, |X[peak+1]| exchanged through LDS                      [barrier]
//     -> every thread: noise/threshold verdict, parabola, int/frac split, bank index
//     -> X[k] * conj(T_j)[(k + s) mod N] / N   (gathered; see layout note)
//     -> IFFT -> |.|^2 windowed first-max, sums (soa_estimator.py:97-143)    [reduction 2]
//   k_finish (shared with the default path) turns the per-record sums into the record.
//
// Roll as a gather.  corr_fft[k] = X[(k - s) mod N] * Tc[k]; this thread holds X[k'] and
// multiplies it by Tc[(k' + s) mod N], i.e. it produces corr_fft rotated by -s, whose
// inverse transform is corr[n] * exp(-2 pi i s n / N): same magnitudes, and only
// magnitudes are used downstream.  Nothing is moved.
//
// Bank layout.  After pass 3 thread (k1 = t >> 5, k2 = t & 31) holds bins
// k = k1 + 16 k2 + 512 k3.  The bank stores Tc at [(k3 * 16 + k1) * 32 + k2]; adding
// s = s1 + 16 s2 + 512 s3 digit-wise gives k1' = (k1 + s1) mod 16 (carry c1, one value per
// half-wave), k2' = (k2 + s2 + c1) mod 32 (a rotation of the half-wave's 32 lanes inside one
// 256-byte run, carry c2 per lane), k3' = (k3 + s3 + c2) mod 32: each half-wave reads two
// contiguous runs per register slot instead of 32 scattered lines.
#include <hip/hip_runtime.h>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;

namespace {

// float32 on purpose (the reference's carrier statistics and its parabola are float32 under
// NumPy >= 2), and nothing may be contracted into an fma: NumPy rounds every operation.
struct PreshiftVerdict {
    bool carrier, index_error;
    bool int_offset = false;   // the interpolator returned the Python int 0 (cosine, cos(omega) > 1)
    float peak_mag, noise_rms, offset;
    double offset_f64;   // what goes into the record (fastdet keeps the double parabola)
    int s_mod;   // int_shift mod N, in [0, N)
    int bank;    // pre-shifted template index
    int int_shift;
};

// fastdet-compatible verdict (fastcard/cardet.c:7-41, fastdet/corr_detector.cpp:88-101,177-197):
// float32 power-domain threshold c + s * noise_power with noise_power = (sum - 2 max)/(N - 1),
// integer roll by -argmax against the UNSHIFTED template, carrier offset = parabola on the
// square roots (double), clipped to +-0.5.
__device__ __forceinline__ PreshiftVerdict fastdet_verdict(const DevCfg& cfg, float peak_pow,
                                                           float sum_pow, int peak_idx, float a_mag,
                                                           float c_mag) {
#pragma clang fp contract(off)
    PreshiftVerdict v;
    const int n = cfg.block_len;
    float noise_power = 0.f;
    if (sum_pow != 0.f) noise_power = (sum_pow - 2.0f * peak_pow) / float(n - 1);
    const float thr = float(cfg.car_thr[0]) + float(cfg.car_thr[1]) * noise_power;
    v.carrier = peak_pow > thr;
    v.index_error = false;
    v.peak_mag = sqrtf(peak_pow);
    v.noise_rms = sqrtf(noise_power);
    const double a = double(a_mag), b = sqrt(double(peak_pow)), c = double(c_mag);
    double off = (c - a) / (4 * b - 2 * a - 2 * c);
    off = off < -0.5 ? -0.5 : off > 0.5 ? 0.5 : off;
    v.offset = v.carrier ? float(off) : 0.f;
    v.offset_f64 = v.carrier ? off : 0.0;
    v.bank = 0;
    v.int_shift = -peak_idx;
    v.s_mod = (n - peak_idx) & (n - 1);
    return v;
}

// peak_mag, a, c = |X[peak]|, |X[peak-1]|, |X[peak+1]|
// (peak_pow = |X[peak]|^2 as summed; only the fastdet verdict, which never takes roots, uses it)
// PARABOLIC_ONLY: the interpolator is known at compile time to be the reference's default.  With
// the choice a run-time one inside the fused kernel -- every thread evaluates the verdict -- the
// logf / acosf / atanf / sinf branches cost k_preshift 17 % (2.32 against 1.92 ms per 32768 blocks)
// although never taken: the same lesson as the L2-twiddle form of k_correlate, no run-time-selected
// alternative code inside a hot kernel.
template <bool PARABOLIC_ONLY = false>
__device__ __forceinline__ PreshiftVerdict preshift_verdict(const DevCfg& cfg, float peak_pow_raw,
                                                            float peak_mag, float sum_mag2,
                                                            float sum_mag, int peak_idx, float a,
                                                            float c, int num) {
#pragma clang fp contract(off)
    if (cfg.variant == 2) return fastdet_verdict(cfg, peak_pow_raw, sum_mag2, peak_idx, a, c);
    PreshiftVerdict v;
    const int n = cfg.block_len;
    const float peak_pow = peak_mag * peak_mag;
    const float noise_pow = (sum_mag2 - 2.0f * peak_pow) / float(n - 1);   // carrier_detect.py:99-107
    const float noise_rms = sqrtf(noise_pow);
    float thr = float(cfg.car_thr[0]) + float(cfg.car_thr[1]) * (noise_rms * noise_rms);
    if (cfg.car_want_std) {
        const double m1 = double(sum_mag) / n, m2 = double(sum_mag2) / n;
        thr += float(cfg.car_thr[2]) * float(m2 - m1 * m1);
    }
    thr = sqrtf(thr);
    v.carrier = peak_mag > thr;
    v.peak_mag = peak_mag;
    v.noise_rms = noise_rms;
    // (every interpolator but `none` reads fft_mag[peak + 1]: carrier_interpolators.py:43,51,85)
    v.index_error = v.carrier && (PARABOLIC_ONLY || cfg.interp != THR_INTERP_NONE) && peak_idx + 1 >= n;
    if (v.index_error) v.carrier = false;
    const float b = peak_mag;
    v.offset = 0.0f;
    if (v.carrier) {
        // float32 in, float32 out, operation by operation as NumPy evaluates the reference's lines
        if (PARABOLIC_ONLY || cfg.interp == THR_INTERP_PARABOLIC) {   // carrier_interpolators.py:40-45
            const float two_a = 2.0f * a, two_c = 2.0f * c, four_b = 4.0f * b;
            v.offset = (c - a) / ((four_b - two_a) - two_c);
        } else if (!PARABOLIC_ONLY && cfg.interp == THR_INTERP_GAUSSIAN) {   // :48-54, the same on the logarithms
            const float la = logf(a), lb = logf(b), lc = logf(c);
            const float two_a = 2.0f * la, two_c = 2.0f * lc, four_b = 4.0f * lb;
            v.offset = (lc - la) / ((four_b - two_a) - two_c);
        } else if (!PARABOLIC_ONLY && cfg.interp == THR_INTERP_COSINE) {     // :84-92
            const float cos_omega = (a + c) / (2.0f * b);
            if (!(cos_omega > 1.0f)) {
                const float omega = acosf(cos_omega);
                const float theta = atanf((a - c) / ((2.0f * b) * sinf(omega)));
                v.offset = -theta / omega;
            } else {
                v.int_offset = true;   // `return 0` (:87-88): an int, and the .toad column reads "0"
            }
        }
    }
    v.offset_f64 = double(v.offset);
    // shift = -(bin + offset), integer part rolled, rest -> nearest bank entry
    // (carrier_sync.py:71, detect_preshift.py:62-65,42-45; np.round == rint, half to even)
    const double shift = -(double(peak_idx) + double(v.offset));
    const double si = rint(shift);
    const double frac = shift - si;
    v.bank = int(rint((frac + 0.5) * double(num - 1)));
    v.bank = v.bank < 0 ? 0 : v.bank >= num ? num - 1 : v.bank;
    const long long sim = ((long long)si % n + n) % n;
    v.s_mod = int(sim);
    v.int_shift = int(si);
    return v;
}

__device__ __forceinline__ thr_record preshift_record(const PreshiftVerdict& vd, long long block_idx,
                                                      int peak_idx) {
    thr_record r;
    r.block_idx = block_idx;
    r.flags = (vd.carrier ? THR_FLAG_CARRIER : 0u) | (vd.index_error ? THR_FLAG_INDEX_ERROR : 0u) |
              (vd.carrier && vd.int_offset ? THR_FLAG_INT_OFFSET : 0u);
    r.template_id = 0;
    r.carrier_bin = peak_idx;
    r.corr_sample = -1;
    r.carrier_offset = vd.offset_f64;
    r.corr_offset = 0.0;
    r.carrier_energy = vd.peak_mag;
    r.carrier_noise = vd.noise_rms;
    r.corr_energy = 0.f;
    r.corr_noise = 0.f;
    // variant debug info: rolled shift (high word), bank index (low word)
    r.reserved = vd.carrier ? (((unsigned long long)(unsigned)vd.int_shift << 32) | unsigned(vd.bank)) : 0ull;
    return r;
}

}  // namespace

// Verdict stage of the multi-pass (generic block length) pipeline: one thread per block,
// from the CarStats of generic_carrier; the roll and the bank index travel in ShiftParams.
__global__ __launch_bounds__(64) void k_fit_preshift(int n_blocks, DevCfg cfg, int num,
                                                     const CarStats* __restrict__ stats,
                                                     const long long* __restrict__ block_idx,
                                                     ShiftParams* __restrict__ shifts,
                                                     thr_record* __restrict__ records) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const CarStats st = stats[b];
    const PreshiftVerdict vd = preshift_verdict(cfg, st.peak_mag * st.peak_mag, st.peak_mag,
                                                st.sum_mag2, st.sum_mag, st.peak_idx, st.nb[2],
                                                st.nb[4], num);
    shifts[b].si_mod = vd.s_mod;
    shifts[b].bank = vd.bank;
    records[b] = preshift_record(vd, block_idx ? block_idx[b] : (long long)b, st.peak_idx);
}

template <int FMT, bool CAR_STD, bool COR_STD, bool PARABOLIC_ONLY>
__global__ __launch_bounds__(NT) void k_preshift(const void* __restrict__ samples, int n_blocks,
                                                 DevCfg cfg, const cpx* __restrict__ tables,
                                                 const cpx* __restrict__ bank, int num,
                                                 const long long* __restrict__ block_idx,
                                                 CorrStats* __restrict__ corr_stats,
                                                 thr_record* __restrict__ records) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    // the reductions use the first 512 B of the scratch (2 parities x 8 waves x 32 B)
    float* sc_nb = reinterpret_cast<float*>(sc_red + 512);   // [parity][2]: |X[peak -+ 1]|^2

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;
    int parity = 0;
    cpx tw0[R1], tw1[R1];   // block-invariant pass-1 twiddles of this thread's two columns
    pass1_twiddles(lds, tw0, tw1, pass1_scale<RawSamples<FMT>>());

    RawSamples<FMT> cur;
    if (int(blockIdx.x) < n_blocks)
        cur.load(static_cast<const unsigned char*>(samples) + size_t(blockIdx.x) * blk_bytes,
                 opaque_tid());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        // (the previous block's last LDS reads -- pass 3 or pass C -- precede a reduction barrier)
        fwd_pass1_pre(lds, cur, tw0, tw1);
        // next block's samples, into the registers pass 1 has just consumed
        if (b + int(gridDim.x) < n_blocks)
            cur.load(static_cast<const unsigned char*>(samples) + size_t(b + gridDim.x) * blk_bytes,
                     opaque_tid());
        __syncthreads();
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);
        cpx v[R3];
        fwd_pass3(lds, v);

        // ---- carrier statistics over the spectrum held in registers
        const int t = opaque_tid();
        const int kbase = (t >> 5) + 16 * (t & 31);
        float csums[2] = {0.f, 0.f};
        float bestp = -1.0f;
        unsigned bestwi = 0;
        static_for<R3>([&](auto K) {
            constexpr int k3 = decltype(K)::value;
            const float p = cnorm(v[brev(k3, R3)]);
            csums[0] += p;
            if constexpr (CAR_STD) csums[1] += __builtin_amdgcn_sqrtf(p);
            const unsigned wi = unsigned(kbase + 512 * k3 - cfg.win_lo) & unsigned(N - 1);
            const bool take = wi < unsigned(cfg.win_count) &&
                              (p > bestp || (p == bestp && wi < bestwi));
            bestp = take ? p : bestp;
            bestwi = take ? wi : bestwi;
        });
        unsigned long long best =
            bestp < 0.f ? 0ull
                        : ((unsigned long long)__float_as_uint(bestp) << 32) | (0xFFFFFFFFu - bestwi);
        constexpr int NC = CAR_STD ? 2 : 1;
        double ctot[2] = {0, 0};
        block_reduce<NC, NT / 64, true>(reinterpret_cast<float(&)[NC]>(csums),
                                  reinterpret_cast<double(&)[NC]>(ctot), best, sc_red, parity);
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        int peak_idx = int(wi) + cfg.win_lo;
        if (peak_idx > N) peak_idx -= N;  // sic: '>' (carrier_detect.py:151)
        // the owners of bins peak-1 and peak+1 (indices wrap; peak+1 past the end is flagged
        // as the reference's IndexError by the verdict) publish their powers
        {
            const unsigned u = unsigned(kbase - peak_idx + 1) & unsigned(N - 1);   // 0 or 2 for owners
            const unsigned r = u & 511u, k3s = (32u - (u >> 9)) & 31u;
            if (r == 0u || r == 2u) {
                float val = 0.f;
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    val = (k3s == unsigned(k3)) ? cnorm(v[brev(k3, R3)]) : val;
                });
                sc_nb[parity * 2 + (r >> 1)] = val;
            }
        }
        __syncthreads();
        const float pa = sc_nb[parity * 2], pc = sc_nb[parity * 2 + 1];
        parity ^= 1;
        const PreshiftVerdict vd =
            preshift_verdict<PARABOLIC_ONLY>(cfg, __uint_as_float(unsigned(best >> 32)),
                             sqrtf(__uint_as_float(unsigned(best >> 32))), (float)ctot[0],
                             CAR_STD ? (float)ctot[1] : 0.f, peak_idx, sqrtf(pa), sqrtf(pc), num);
        if (t == 0) records[b] = preshift_record(vd, block_idx ? block_idx[b] : (long long)b, peak_idx);
        if (!vd.carrier) continue;   // block-uniform: every thread computed the same verdict

        // ---- X[k] * conj(T_bank)[(k + s) mod N] / N, gathered (layout note above)
        cpx z[R3];
        {
            const int s = vd.s_mod;
            const int k1s = (t >> 5) + (s & 15);
            const int k2s = (t & 31) + ((s >> 4) & 31) + (k1s >> 4);
            const int q3 = (s >> 9) + (k2s >> 5);
            const cpx* tb = bank + size_t(vd.bank) * N + ((k1s & 15) * 32 + (k2s & 31));
            static_for<R3>([&](auto K) {
                constexpr int k3 = decltype(K)::value;
                const cpx w = tb[((k3 + q3) & 31) * 512];
                z[brev(k3, R3)] = cmul(v[brev(k3, R3)], w);
            });
        }
        inv_passA(lds, z);
        __builtin_amdgcn_sched_barrier(0);
        inv_passB<true>(lds, static_cast<const cpx*>(cfg.gtw));   // twiddles from the L2 table (-0.6 %)
        __syncthreads();
        cpx c0[R1], c1[R1];
        inv_passC(lds, c0, c1);

        // ---- |corr|^2, windowed first-max, optional std sums (as k_correlate)
        float sums[3] = {0.f, 0.f, 0.f};
        float pw0[R1], pw1[R1];
        float cbestp = -1.0f;
        int cbestn = 0;
        const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
        static_for<R1>([&](auto K) {
            constexpr int n1 = decltype(K)::value;
            pw0[n1] = cnorm(c0[brev(n1, R1)]);
            pw1[n1] = cnorm(c1[brev(n1, R1)]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int n = n1 * S1 + 2 * t + e;
                const float pw = e ? pw1[n1] : pw0[n1];
                const bool take = unsigned(n - cfg.corr_lo) < win_w && pw > cbestp;
                cbestp = take ? pw : cbestp;
                cbestn = take ? n : cbestn;
                if constexpr (COR_STD) {
                    if (n < cfg.corr_len) {
                        sums[2] += pw;
                        sums[1] += __builtin_amdgcn_sqrtf(pw);
                    }
                }
            }
        });
        unsigned long long cbest =
            cbestp < 0.f ? 0ull
                         : ((unsigned long long)__float_as_uint(cbestp) << 32) |
                               (0xFFFFFFFFu - unsigned(cbestn));
        constexpr int NS = COR_STD ? 3 : 1;
        double tot[3] = {0, 0, 0};
        block_reduce<NS, NT / 64, true>(reinterpret_cast<float(&)[NS]>(sums),
                                  reinterpret_cast<double(&)[NS]>(tot), cbest, sc_red, parity);
        parity ^= 1;
        const int pk = int(0xFFFFFFFFu - unsigned(cbest & 0xFFFFFFFFu));
        CorrStats* cs = corr_stats + b;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int delta = pk - 1 - (2 * t + e);
            const unsigned d = unsigned(-delta) & 1023u;
            const int n1s = (delta + int(d)) >> 10;
            float val = 0.f;
            static_for<R1>([&](auto K) {
                constexpr int n1 = decltype(K)::value;
                val = (n1s == n1) ? (e ? pw1[n1] : pw0[n1]) : val;
            });
            if (d < 3u && n1s >= 0 && n1s < R1) cs->m2[d] = val;
        }
        if (t == 0) {
            cs->pm2 = __uint_as_float(unsigned(cbest >> 32));
            cs->pk = pk;
            cs->sum_x2 = (float)ctot[0];   // sum |X|^2: a roll does not change it
            cs->sum_mag = COR_STD ? (float)tot[1] : 0.f;
            cs->sum_mag2 = COR_STD ? (float)tot[2] : 0.f;
        }
    }
}

// ------------------------------------------------------------------ launchers
namespace {
typedef void (*preshift_fn)(const void*, int, DevCfg, const cpx*, const cpx*, int,
                            const long long*, CorrStats*, thr_record*);
template <int FMT, bool PAR>
preshift_fn pick_preshift(bool car_std, bool cor_std) {
    if (car_std) return cor_std ? &k_preshift<FMT, true, true, PAR> : &k_preshift<FMT, true, false, PAR>;
    return cor_std ? &k_preshift<FMT, false, true, PAR> : &k_preshift<FMT, false, false, PAR>;
}
// par: the carrier interpolator is the parabolic one (THR_INTERP_PARABOLIC; the fastdet variant's
// verdict never reaches the interpolators)
preshift_fn preshift_variant(int fmt, bool car_std, bool cor_std, bool par) {
    if (par)
        return fmt == THR_IN_U8 ? pick_preshift<THR_IN_U8, true>(car_std, cor_std)
                                : pick_preshift<THR_IN_C64, true>(car_std, cor_std);
    return fmt == THR_IN_U8 ? pick_preshift<THR_IN_U8, false>(car_std, cor_std)
                            : pick_preshift<THR_IN_C64, false>(car_std, cor_std);
}
}  // namespace

hipError_t launch_fit_preshift(int n_blocks, const DevCfg& cfg, int num, const CarStats* stats,
                               const long long* block_idx, ShiftParams* shifts,
                               thr_record* records, hipStream_t stream) {
    hipLaunchKernelGGL(k_fit_preshift, dim3((n_blocks + 63) / 64), dim3(64), 0, stream, n_blocks, cfg,
                       num, stats, block_idx, shifts, records);
    return hipGetLastError();
}

hipError_t prepare_preshift_16k() {
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 2; ++c)
                for (int par = 0; par < 2; ++par) {
                    hipError_t e = hipFuncSetAttribute(
                        reinterpret_cast<const void*>(preshift_variant(fmt, a, c, par != 0)),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                    if (e != hipSuccess) return e;
                }
    return hipSuccess;
}

hipError_t launch_preshift_16k(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                               const float2* tables, const float2* bank, int num,
                               const long long* block_idx, CorrStats* corr_stats,
                               thr_record* records, int grid, hipStream_t stream) {
    preshift_fn fn = preshift_variant(fmt, cfg.car_want_std != 0, cfg.cor_want_std != 0,
                                      cfg.interp == THR_INTERP_PARABOLIC);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(bank), num,
                       block_idx, corr_stats, records);
    return hipGetLastError();
}

}  // namespace thr
