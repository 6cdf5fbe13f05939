// Carrier stage, fit and record finishing of the block_len = 16384 path -- the kernels around
// k_correlate (detect16k.hip).  A translation unit of its own so that it can be compiled
// with `-mllvm -amdgpu-sched-strategy=max-ilp` (thrifty_amd/build.py): measured -4 % on the
// pruned carrier kernel and -5 % on k_fit, while the same flag costs k_correlate +11 %.
//
//   k_carrier / k_carrier_pruned : u8/c64 -> FFT#1 -> |X|^2 sum, windowed first-max, 7-bin
//                                  neighbourhood (signal_utils.py:21-25, carrier_detect.py:99-154)
//   k_fit       : noise/threshold verdict + Dirichlet fit (MINPACK lmdif) + shift phasors
//                 (carrier_detect.py:61-115, carrier_sync.py:150-196)
//   k_finish    : correlation noise, threshold verdict, log-parabola (soa_estimator.py:108-170)
//   k_compact_* : order-preserving compaction of detected records (count / scan / scatter)
#include <hip/hip_runtime.h>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "lmdif8.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;

// =========================================================================
// K_A: carrier stage
// =========================================================================
// For a power pmax > 0: *mag = sqrtf(pmax) (correctly rounded) and the return value is the smallest
// float p with sqrtf(p) == *mag -- every power in [that, pmax] has the maximum's float32 magnitude.
// A correctly rounded square root is monotone, so the set is an interval; its lower edge is the
// square of the midpoint between *mag and its predecessor (exact in double: 25 bits squared), which
// no float equals (an odd 25-bit integer squared is odd: 49 or 50 significant bits), so there are
// no ties to break.  pmax <= 0 (an all-zero window): 0, everything ties.
__device__ __forceinline__ float sqrt_preimage_lo(float pmax, float* mag) {
    const float m = sqrtf(pmax > 0.f ? pmax : 0.f);
    *mag = m;
    if (!(m > 0.f)) return 0.f;
    if (!(m < __builtin_inff())) return pmax;    // (overflowed power: only itself)
    const float below = __uint_as_float(__float_as_uint(m) - 1u);
    const double mid = 0.5 * (double(m) + double(below));
    const double lo = mid * mid;
    float f = float(lo);
    if (double(f) <= lo) f = __uint_as_float(__float_as_uint(f) + 1u);
    return f;
}

// ALLBINS: the window is the whole spectrum -- the reference's default carrier_window '0--1'
// (settings.py:75-80): no window test, and a thread's bins ascend with k3.
template <int FMT, bool WANT_STD, bool DUMP, bool ALLBINS>
__global__ __launch_bounds__(NT) void k_carrier(const void* __restrict__ samples, int n_blocks,
                                                DevCfg cfg, const cpx* __restrict__ tables,
                                                CarStats* __restrict__ stats,
                                                cpx* __restrict__ dump_fft) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    int parity = 0;
    cpx tw0[R1], tw1[R1];   // block-invariant pass-1 twiddles of this thread's two columns
    pass1_twiddles(lds, tw0, tw1, pass1_scale<RawSamples<FMT>>());

    RawSamples<FMT> cur;
    if (int(blockIdx.x) < n_blocks)
        cur.load(static_cast<const unsigned char*>(samples) + size_t(blockIdx.x) * blk_bytes,
                 opaque_tid());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        // (previous block's pass-3 LDS reads all precede its reduction barrier)
        float energy;
        fwd_pass1_pre(lds, cur, tw0, tw1, &energy);
        // next block's samples, into the registers pass 1 has just consumed: issued now, used
        // one iteration later (no second register set, no copies)
        if (b + int(gridDim.x) < n_blocks)
            cur.load(static_cast<const unsigned char*>(samples) + size_t(b + gridDim.x) * blk_bytes,
                     opaque_tid());
        __syncthreads();
        // passes 2 and 3 of row k1 are done by the same half-wave: no barrier between them
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);  // keep the passes' register working sets apart
        cpx v[R3];
        fwd_pass3(lds, v);

        // ---- statistics over the spectrum held in registers
        // First-max inside the (wrapping) window over float32 MAGNITUDES, as the reference takes it
        // (np.argmax of np.abs, carrier_detect.py:146): powers an ulp apart can round to the same
        // |X|, and then the lowest window index wins.  Taking sqrtf of all 32 bins and comparing
        // (|X|, index) pairs bin by bin cost more than the transform itself (1100 of the kernel's
        // 2000 instructions per wave and block); instead: the maximum POWER first (one v_max3 per
        // two bins, a DPP max per wave), ONE correctly rounded sqrtf of it per wave, the smallest
        // power that still rounds to that magnitude (sqrt_preimage_lo), and a second sweep that
        // names the lowest window index among the bins at or above it -- the same bin the
        // bin-by-bin comparison finds: sqrtf is monotone, so exactly the powers in [plo, max] share
        // the maximum's magnitude.  Across waves the key (|X| bits, -index) decides as before.
        const int t = opaque_tid();
        const int kbase = (t >> 5) + 16 * (t & 31);
        float pw[R3];          // powers (the seven neighbours of the peak are picked from them)
        float ew[R3];          // the same inside the window, -1 outside
        float tmax = -1.0f;
        float smag = 0.f;
        const unsigned wbase = unsigned(kbase - cfg.win_lo);
        static_for<R3>([&](auto K) {
            constexpr int k3 = decltype(K)::value;
            const float p = cnorm(v[brev(k3, R3)]);
            pw[k3] = p;
            if constexpr (WANT_STD) smag += __builtin_amdgcn_sqrtf(p);
            if constexpr (ALLBINS) {
                ew[k3] = p;
            } else {
                const unsigned wi = (wbase + unsigned(512 * k3)) & unsigned(N - 1);
                ew[k3] = wi < unsigned(cfg.win_count) ? p : -1.0f;
            }
            tmax = __builtin_fmaxf(tmax, ew[k3]);
        });
        const float wmax = wave_max_f32(tmax);
        float wmag;
        const float plo = sqrt_preimage_lo(wmax, &wmag);
        // second sweep: lowest window index among the bins whose power is >= plo.  Almost no bin
        // is: the test of a bin is ONE v_cmp and a wave-uniform branch over its (empty) lane mask
        unsigned cand = 0xFFFFFFFFu;
        static_for<R3>([&](auto K) {
            constexpr int k3 = R3 - 1 - decltype(K)::value;       // downwards: with ALLBINS the lowest sticks
            const bool hit = ew[k3] >= plo;                        // (NaN, outside (-1): never; plo >= 0)
            if (__builtin_amdgcn_ballot_w64(hit) != 0) {
                if constexpr (ALLBINS) {
                    cand = hit ? unsigned(kbase + 512 * k3) : cand;
                } else {
                    const unsigned wi = (wbase + unsigned(512 * k3)) & unsigned(N - 1);
                    cand = (hit && wi < cand) ? wi : cand;
                }
            }
        });
        const unsigned wwi = wave_min_u32(cand);
        unsigned long long best =
            wmax < 0.f ? 0ull : ((unsigned long long)__float_as_uint(wmag) << 32) | (0xFFFFFFFFu - wwi);
        // sum |X|^2 = N sum |x|^2 (Parseval) from the samples pass 1 held; sum |X| only for the
        // stddev term (v_sqrt_f32 to 1 ulp: it feeds a variance, not a comparison of bins)
        float sums[2] = {energy, smag};
        double tot[2];
        block_reduce<WANT_STD ? 2 : 1, NT / 64, true>(reinterpret_cast<float(&)[WANT_STD ? 2 : 1]>(sums),
                                       reinterpret_cast<double(&)[WANT_STD ? 2 : 1]>(tot), best,
                                       sc_red, parity);
        parity ^= 1;
        tot[0] *= double(N);
        const unsigned wi = 0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu);
        int peak_idx = int(wi) + cfg.win_lo;
        if (peak_idx > N) peak_idx -= N;  // sic: '>' (carrier_detect.py:151)
        // 7-bin neighbourhood |X[peak-3..peak+3]| (indices wrap here; K_fit flags the cases
        // where the reference would raise IndexError).  This thread's bins are
        // kbase + 512*k3, so it holds neighbour d iff (kbase - peak + 3) mod 512 == d < 7,
        // at k3 = -((kbase - peak + 3) >> 9) mod 32: exactly seven threads store one float.
        CarStats* st = stats + b;
        {
            const unsigned u = unsigned(kbase - peak_idx + 3) & unsigned(N - 1);
            const unsigned r = u & 511u, k3s = (32u - (u >> 9)) & 31u;
            if (r < 7u) {      // (at most two lanes of a wave, none in half of the waves)
                float val = 0.f;
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    val = (k3s == unsigned(k3)) ? pw[k3] : val;
                });
                st->nb[r] = sqrtf(val);
            }
        }
        if constexpr (DUMP) {
            cpx* out = dump_fft + size_t(b) * N;
            static_for<R3>([&](auto K) {
                constexpr int k3 = decltype(K)::value;
                out[kbase + 512 * k3] = v[brev(k3, R3)];
            });
        }
        if (t == 0) {
            st->sum_mag2 = (float)tot[0];
            st->sum_mag = WANT_STD ? (float)tot[1] : 0.f;
            st->peak_mag = __uint_as_float(unsigned(best >> 32));
            st->peak_idx = peak_idx;
            st->pad = 0;
        }
    }
}

// =========================================================================
// K_A, pruned: carrier window (plus the 3-bin fit margin) of at most 128 bins
// =========================================================================
// The carrier stage only ever looks at the bins of the window, +-3 neighbours for the
// fit, and at sum |X|^2.  With the window inside [0,128) the needed bins are
// k = k1 + 16*k2 with k2 < 8 and k3 = 0: pass 2 keeps 8 of its 32 outputs, pass 3
// degenerates to a 32-term sum in 128 threads, and sum |X|^2 = N * sum |x|^2 (Parseval)
// comes from the samples pass 1 already holds.  (Not usable with a stddev threshold
// term, which needs every |X|: the launcher then picks the full kernel.)
constexpr int PRUNE_K2 = 8;
constexpr int PRUNE_BINS = R1 * PRUNE_K2;  // 128

// SHIFTED: the window does not start near bin 0 -- multiply the samples by
// exp(-2 pi i base n / N), base = win_lo - 3 (exact: an integer shift through the root
// table), which moves spectrum bin `base + k'` to k'; the window then occupies
// k' = 3 .. 3 + count - 1 and the same pruned transform applies to ANY window of at most
// 122 bins (negative bins, wrap-around windows included).
template <int FMT, bool SHIFTED>
__global__ __launch_bounds__(NT) void k_carrier_pruned(const void* __restrict__ samples,
                                                       int n_blocks, DevCfg cfg,
                                                       const cpx* __restrict__ tables,
                                                       const cpx* __restrict__ twn,
                                                       CarStats* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);
    float* sc_bins = reinterpret_cast<float*>(sc_red + 2 * red_slot_bytes<NT / 64>());  // [128] |X[k']|^2
    float2* sc_rp = reinterpret_cast<float2*>(sc_bins + PRUNE_BINS);                     // [16]

    load_tables(lds, tables);
    // block-invariant shift factors: per sub-sequence n1 (LDS) and per thread (registers)
    const int base = SHIFTED ? ((cfg.win_lo - 3) & (N - 1)) : 0;
    const int win_off = SHIFTED ? 3 : cfg.win_lo;  // window start in the pruned bin domain
    cpx ph[2] = {cpx{1.f, 0.f}, cpx{1.f, 0.f}};
    if constexpr (SHIFTED) {
        if (threadIdx.x < 16) {
            const cpx r = twn[(threadIdx.x * 1024 * base) & (N - 1)];
            sc_rp[threadIdx.x] = float2{r.x, r.y};
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) ph[e] = twn[((2 * int(threadIdx.x) + e) * base) & (N - 1)];
    }
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    int parity = 0;
    cpx tw0[R1], tw1[R1];   // block-invariant pass-1 twiddles (unshifted variant only)
    if constexpr (!SHIFTED) pass1_twiddles(lds, tw0, tw1, pass1_scale<RawSamples<FMT>>());

    RawSamples<FMT> cur;
    if (int(blockIdx.x) < n_blocks)
        cur.load(static_cast<const unsigned char*>(samples) + size_t(blockIdx.x) * blk_bytes,
                 opaque_tid());
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        float sums[1];
        if constexpr (SHIFTED) {
            cpx p0 = ph[0], p1 = ph[1];
            asm volatile("" : "+v"(p0), "+v"(p1));  // keep the loop body free of hoisted products
            fwd_pass1<true>(lds, cur, sc_rp, p0, p1, &sums[0]);
        } else {
            fwd_pass1_pre(lds, cur, tw0, tw1, &sums[0]);
        }
        // next block's samples, into the registers pass 1 has just consumed
        if (b + int(gridDim.x) < n_blocks)
            cur.load(static_cast<const unsigned char*>(samples) + size_t(b + gridDim.x) * blk_bytes,
                     opaque_tid());
        __syncthreads();
        fwd_pass2<PRUNE_K2>(lds);
        __builtin_amdgcn_sched_barrier(0);
        // pass 3, output k3 = 0 only: the plain sum of a chunk -- 16 x PRUNE_K2 = 128 chunks of 32,
        // FOUR threads per chunk (8 terms each, then a quad sum on the DPP path): every lane of
        // every wave has a quarter of a sum to do, instead of a quarter of the lanes a whole one
        static_assert(PRUNE_K2 * R1 * 4 == NT, "four threads per kept chunk");
        const int t = opaque_tid();
        const int k2 = (t >> 2) & (PRUNE_K2 - 1), k1 = t >> 5, part = t & 3;
        const int k = k1 + 16 * k2;  // pruned-domain bin
        unsigned long long best = 0;
        {
            const f4* src = reinterpret_cast<const f4*>(lds + k1 * ROW + k2 * CHUNK) + part * (R3 / 8);
            f4 acc = src[0];
#pragma unroll
            for (int j = 1; j < R3 / 8; ++j) acc += src[j];
            const cpx x = cpx{quad_sum(acc.x + acc.z), quad_sum(acc.y + acc.w)};
            const float p = cnorm(x);
            const unsigned wi = unsigned(k - win_off) & unsigned(N - 1);
            // the key carries |X| (not |X|^2): the reference takes argmax over float32 magnitudes,
            // where powers an ulp apart can collide -- the first bin then wins, here as there
            if (part == 0) {
                sc_bins[k] = p;
                if (wi < unsigned(cfg.win_count))
                    best = ((unsigned long long)__float_as_uint(sqrtf(p)) << 32) | (0xFFFFFFFFu - wi);
            }
        }
        double tot[1];
        block_reduce<1, NT / 64, true>(sums, tot, best, sc_red, parity);   // (key: |X| bits, -bin)
        parity ^= 1;
        const int wi = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
        int peak_idx = wi + cfg.win_lo;
        if (peak_idx > N) peak_idx -= N;  // sic: '>' (carrier_detect.py:151)
        CarStats* st = stats + b;
        if (t < 7) st->nb[t] = sqrtf(sc_bins[wi + win_off - 3 + t]);
        if (t == 0) {
            st->sum_mag2 = (float)(tot[0] * double(N));  // Parseval
            st->sum_mag = 0.f;
            st->peak_mag = __uint_as_float(unsigned(best >> 32));
            st->peak_idx = peak_idx;
            st->pad = 0;
        }
    }
}

// =========================================================================
// K_fit: 8 lanes per block (one per fitted point, lane 7 idles in the sums).  The fit itself
// is MINPACK's lmdif as SciPy's curve_fit drives it (lmdif8.hpp).
// =========================================================================
__global__ __launch_bounds__(64) void k_fit(int n_blocks, DevCfg cfg,
                                            const CarStats* __restrict__ stats,
                                            const long long* __restrict__ block_idx,
                                            ShiftParams* __restrict__ shifts,
                                            int* __restrict__ work_list,
                                            int* __restrict__ work_count,
                                            thr_record* __restrict__ records,
                                            CorrStats* __restrict__ corr_stats,
                                            const double* __restrict__ forced_offset) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = gid & 7;
    int b = gid >> 3;
    const bool valid = b < n_blocks;
    if (!valid) b = n_blocks - 1;  // keep the whole group in the shuffles
    const CarStats* st = stats + b;
    const int n = cfg.block_len;
    const float peak_mag = st->peak_mag, sum_mag2 = st->sum_mag2;
    const int peak_idx = st->peak_idx;
    // float32 arithmetic on purpose: the reference's carrier statistics are
    // float32 under NumPy >= 2 (carrier_detect.py:99-115)
    const float peak_pow = peak_mag * peak_mag;
    const float noise_pow = (sum_mag2 - 2.0f * peak_pow) / float(n - 1);
    const float noise_rms = sqrtf(noise_pow);
    // (Python-float coefficient x np.float32 statistic: NEP 50 rounds the coefficient to float32)
    float thr = float(cfg.car_thr[0]) + float(cfg.car_thr[1]) * (noise_rms * noise_rms);
    if (cfg.car_want_std) {
        const double m1 = double(st->sum_mag) / n, m2 = double(sum_mag2) / n;
        thr += float(cfg.car_thr[2]) * float(m2 - m1 * m1);
    }
    thr = sqrtf(thr);
    bool detected = peak_mag > thr;
    unsigned flags = 0;
    double offset = 0.0;
    // (forced_offset: thr_detect_offsets -- the caller's own interpolator has produced the sub-bin
    // offset, Synchronizer.sync with a replaced `interpolator`, carrier_sync.py:52-76; what that
    // interpolator reads of the spectrum, and where it raises, is its business)
    if (detected && forced_offset == nullptr && peak_idx + 3 >= n) {
        flags |= THR_FLAG_INDEX_ERROR;  // carrier_sync.py:187 raises here
        detected = false;
    }
    // the fit is group-uniform only if `detected` is; it is (same inputs in all 8 lanes)
    if (detected) {
        flags |= THR_FLAG_CARRIER;
        if (forced_offset != nullptr)
            offset = forced_offset[b];
        else {
            int info = 0;
            offset = lmdif_dirichlet8(st->nb[j < 7 ? j : 6], st->nb[3], j, double(n),
                                      double(cfg.carrier_len), &info);
            // curve_fit raises RuntimeError for these exit codes and the reference does not catch it
            // (carrier_sync.py:189); the record keeps lmdif's last iterate and says so
            if (info >= 5) flags |= THR_FLAG_FIT_UNCONVERGED;
        }
#ifdef THR_DEBUG_FIT
        {
            double lo = offset, hi = offset;
            for (int m = 1; m < 8; m <<= 1) {
                lo = fmin(lo, __shfl_xor(lo, m, 64));
                hi = fmax(hi, __shfl_xor(hi, m, 64));
            }
            if (hi != lo && j == 0) printf("fit disagreement blk %d: lo %.17g hi %.17g\n", b, lo, hi);
        }
#endif
        // shift = -(bin + offset)  (carrier_sync.py:71)
        const double s = -(double(peak_idx) + offset);
        const double si = rint(s);
        const int r1 = n >= 1024 ? n / 1024 : 1;  // first-pass radix of the LDS path (unused otherwise)
        ShiftParams* sp = shifts + b;
        if (valid) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int jj = 2 * j + q;
                double a = s * double(jj) / double(r1);
                a -= rint(a);
                double sn, cs;
                sincospi(2.0 * a, &sn, &cs);
                sp->rpow[jj] = float2{float(cs), float(sn)};
            }
            if (j < 4 && n > 16384) {  // long blocks: phasor step between the R0 leading sub-sequences
                double a = s * double(j) / double(n / 16384);
                a -= rint(a);
                double sn, cs;
                sincospi(2.0 * a, &sn, &cs);
                sp->r0pow[j] = float2{float(cs), float(sn)};
            }
            if (j < cfg.n_seg) {  // sectioned correlate stage: phasor of a section's first sample
                double a = s * (double(cfg.seg_start[j]) / double(n) - 0.5);
                a -= rint(a);
                double sn, cs;
                sincospi(2.0 * a, &sn, &cs);
                sp->segc0[j] = float2{float(cs), float(sn)};
            }
            if (j == 0) {
                double a = -0.5 * s;  // exp(2 pi i * s * (-1/2))
                a -= rint(a);
                double sn, cs;
                sincospi(2.0 * a, &sn, &cs);
                sp->c0 = float2{float(cs), float(sn)};
                long long sim = (long long)si % n;
                if (sim < 0) sim += n;
                sp->si_mod = int(sim);
                sp->sf_over_n = float((s - si) / double(n));
            }
        }
    }
    // work-list append, one atomic per wave (8 blocks) instead of one per block: 8192
    // same-address atomics serialise in L2 and were most of this kernel's 30 us
    {
        const bool push = detected && valid && j == 0;
        const unsigned long long m = __ballot(push);
        if (m != 0) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(work_count, __popcll(m));
            base = __shfl(base, leader, 64);
            if (push) work_list[base + __popcll(m & ((1ull << lane) - 1ull))] = b;
        }
    }
    // sum |X^|^2 of the frequency-shifted spectrum == sum |X|^2 (unit-modulus phasor, Parseval):
    // the LDS-resident correlate kernels take it from here instead of summing it again
    if (valid && j == 0 && corr_stats != nullptr)
        corr_stats[size_t(b) * cfg.n_templates].sum_x2 = sum_mag2;
    if (valid && j < cfg.n_templates) {
        thr_record r;
        r.block_idx = block_idx ? block_idx[b] : (long long)b;
        r.flags = flags;
        r.template_id = j;
        r.carrier_bin = peak_idx;
        r.corr_sample = -1;
        r.carrier_offset = offset;
        r.corr_offset = 0.0;
        r.carrier_energy = peak_mag;
        r.carrier_noise = noise_rms;
        r.corr_energy = 0.f;
        r.corr_noise = 0.f;
        r.reserved = 0;
        records[size_t(b) * cfg.n_templates + j] = r;
    }
}

// =========================================================================
// K_finish: one lane per (block, template) -- noise, threshold verdict, sub-sample
// offset (soa_estimator.py:108-170; float64 like the reference).  Kept out of
// k_correlate so that no workgroup ever waits on one thread's log()/sqrt() chain.
// =========================================================================
__global__ __launch_bounds__(256) void k_finish(int n_records, DevCfg cfg,
                                                const CorrStats* __restrict__ corr_stats,
                                                thr_record* __restrict__ records,
                                                int* __restrict__ work_count,
                                                const CorrStats* __restrict__ seg_stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {  // k_correlate is done with them: re-arm for the next batch
        work_count[0] = 0;
        work_count[1] = 0;  // dynamic-scheduling counter
    }
    if (i >= n_records) return;
    thr_record* r = records + i;
    if (!(r->flags & THR_FLAG_CARRIER)) return;
    const int tpl = i % cfg.n_templates;
    CorrStats cs = corr_stats[i];
    if (seg_stats != nullptr) {
        // sectioned correlate stage (detect_seg.hip): the block's windowed first-max is the first
        // section that holds the largest power (the sections' owned lags ascend with the section
        // index, np.argmax takes the lowest lag: soa_estimator.py:137-143); the stddev sums add up
        const CorrStats* sg = seg_stats + size_t(i) * cfg.n_seg;
        int best = -1;
        double s1 = 0, s2 = 0;
        for (int g = 0; g < cfg.n_seg; ++g) {
            s1 += double(sg[g].sum_mag);
            s2 += double(sg[g].sum_mag2);
            if (sg[g].pk >= 0 && (best < 0 || sg[g].pm2 > sg[best].pm2)) best = g;
        }
        if (best < 0) best = 0;   // (cannot happen: the owned ranges tile a non-empty window)
        cs.pm2 = sg[best].pm2;
        cs.m2[0] = sg[best].m2[0];
        cs.m2[1] = sg[best].m2[1];
        cs.m2[2] = sg[best].m2[2];
        cs.pk = sg[best].pk + cfg.seg_start[best];
        cs.sum_mag = float(s1);
        cs.sum_mag2 = float(s2);
    }
    if (cfg.variant == 2) {
        // fastdet-compatible verdict (fastdet/corr_detector.cpp:103-175): float32, power domain,
        // noise clamped at 0 and -- sic -- computed from the peak power truncated to an integer
        // (estimate_noise takes it as size_t); Gaussian offset on log sqrt(power), clipped +-0.5.
#pragma clang fp contract(off)
        const float peak_power = cs.pm2;
        const float signal_energy = cs.sum_x2 / float(cfg.block_len);
        const float signal_corr_energy = signal_energy * cfg.tmpl_energy[0];
        float noise_power =
            (signal_corr_energy - float((unsigned long long)peak_power)) / float(cfg.block_len);
        if (noise_power < 0) noise_power = 0;
        const float threshold = float(cfg.cor_thr[0]) + float(cfg.cor_thr[1]) * noise_power;  // C floats there
        const bool hit = peak_power > threshold;
        double o = 0.0;
        if (hit && cs.pk != 0 && cs.pk != cfg.corr_len - 1) {
            const double a = log(sqrt(double(cs.m2[0]))), b = log(sqrt(double(cs.m2[1]))),
                         c = log(sqrt(double(cs.m2[2])));
            o = (c - a) / (4 * b - 2 * a - 2 * c);
            o = o < -0.5 ? -0.5 : o > 0.5 ? 0.5 : o;
        }
        r->corr_sample = cs.pk;
        r->corr_offset = o;
        r->corr_energy = sqrtf(peak_power);
        r->corr_noise = sqrtf(noise_power);
        if (hit) r->flags |= THR_FLAG_CORR;
        return;
    }
    const double n = double(cfg.block_len);
    const double xenergy = double(corr_stats[i - tpl].sum_x2) / n;  // mean |X^|^2
    const double pm2 = double(cs.pm2);
    const double peak_mag = sqrt(pm2);
    const double noise_pow = (xenergy * double(cfg.tmpl_energy[tpl]) - pm2) / n;
    const double noise_rms = sqrt(noise_pow);
    double th = cfg.cor_thr[0] + cfg.cor_thr[1] * (noise_rms * noise_rms);
    if (cfg.cor_want_std) {
        const double m1 = double(cs.sum_mag) / cfg.corr_len, m2 = double(cs.sum_mag2) / cfg.corr_len;
        th += cfg.cor_thr[2] * (m2 - m1 * m1);
    }
    th = sqrt(th);
    const bool det = peak_mag > th;
    double off = 0.0;
    if (det && cs.pk != 0 && cs.pk != cfg.corr_len - 1) {
        // log-parabola on magnitudes == the same formula on log |.|^2
        const double la = log(double(cs.m2[0])), lb = log(double(cs.m2[1])),
                     lc = log(double(cs.m2[2]));
        off = 0.5 * (lc - la) / (2 * lb - la - lc);
        off = off < -0.6 ? -0.6 : off > 0.6 ? 0.6 : off;
    }
    r->corr_sample = cs.pk;
    r->corr_offset = off;
    r->corr_energy = (float)peak_mag;
    r->corr_noise = (float)noise_rms;
    if (det) r->flags |= THR_FLAG_CORR;
}

// =========================================================================
// K7: order-preserving compaction of detected records, three small launches:
//   k_compact_count   one workgroup per tile of 2048 records -> kept records per tile
//   k_compact_scan    one workgroup: exclusive scan of the tile counts (+ total)
//   k_compact_scatter one workgroup per tile: ballot prefix inside the tile, 64-byte copies
// (HBM-bound byte work: 64 B read + <= 64 B written per record; the single-workgroup scan it
// replaces took 4.25 ms per 1 Mi records.)
// =========================================================================
constexpr int CMP_T = 256, CMP_PER = 8, CMP_TILE = CMP_T * CMP_PER;

__device__ __forceinline__ bool compact_keep(const thr_record* __restrict__ in, int n, int i) {
    return i < n && (in[i].flags & THR_FLAG_CORR);
}

__global__ __launch_bounds__(CMP_T) void k_compact_count(const thr_record* __restrict__ in, int n,
                                                         int* __restrict__ tile_counts) {
    __shared__ int wcnt[CMP_T / 64];
    const int base = blockIdx.x * CMP_TILE + threadIdx.x;
    int c = 0;
#pragma unroll
    for (int j = 0; j < CMP_PER; ++j) c += __popcll(__ballot(compact_keep(in, n, base + j * CMP_T)));
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

__global__ __launch_bounds__(1024) void k_compact_scan(int* __restrict__ tile_counts, int n_tiles,
                                                       int* __restrict__ n_out) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? tile_counts[i] : 0;
        int incl = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(incl, d, 64);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < wv) woff += wsum[w];
            total += wsum[w];
        }
        const int c = carry;
        if (i < n_tiles) tile_counts[i] = c + woff + incl - v;  // exclusive offset of tile i
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = carry;
}

__global__ __launch_bounds__(CMP_T) void k_compact_scatter(const thr_record* __restrict__ in, int n,
                                                           const int* __restrict__ tile_offs,
                                                           thr_record* __restrict__ out) {
    __shared__ int wcnt[CMP_PER][CMP_T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int base = blockIdx.x * CMP_TILE + threadIdx.x;
    bool keep[CMP_PER];
    int pre[CMP_PER];
#pragma unroll
    for (int j = 0; j < CMP_PER; ++j) {
        keep[j] = compact_keep(in, n, base + j * CMP_T);
        const unsigned long long m = __ballot(keep[j]);
        pre[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[j][wv] = __popcll(m);
    }
    __syncthreads();
    int off = tile_offs[blockIdx.x];
#pragma unroll
    for (int j = 0; j < CMP_PER; ++j) {
        int before = 0, row = 0;
#pragma unroll
        for (int w = 0; w < CMP_T / 64; ++w) {
            if (w < wv) before += wcnt[j][w];
            row += wcnt[j][w];
        }
        if (keep[j]) {
            const float4* src = reinterpret_cast<const float4*>(in + base + j * CMP_T);
            float4* dst = reinterpret_cast<float4*>(out + off + before + pre[j]);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            dst[0] = a;
            dst[1] = b;
            dst[2] = c;
            dst[3] = d;
        }
        off += row;
    }
}

// ------------------------------------------------------------------ launchers
namespace {
typedef void (*carrier_fn)(const void*, int, DevCfg, const cpx*, CarStats*, cpx*);

#ifdef THR_DEV_MINIMAL  // compile-time experiments only: one variant each, fast rebuilds
carrier_fn carrier_variant(int, bool, bool, bool) { return &k_carrier<THR_IN_U8, false, false, true>; }
#else
template <int FMT, bool STD>
carrier_fn pick_carrier(bool dump, bool all) {
    // (stage dumps: the windowed form serves every window)
    if (dump) return &k_carrier<FMT, STD, true, false>;
    return all ? &k_carrier<FMT, STD, false, true> : &k_carrier<FMT, STD, false, false>;
}
carrier_fn carrier_variant(int fmt, bool want_std, bool dump, bool all) {
    if (fmt == THR_IN_U8)
        return want_std ? pick_carrier<THR_IN_U8, true>(dump, all) : pick_carrier<THR_IN_U8, false>(dump, all);
    return want_std ? pick_carrier<THR_IN_C64, true>(dump, all) : pick_carrier<THR_IN_C64, false>(dump, all);
}
#endif
}  // namespace

hipError_t prepare_16k_carrier() {
    // > 64 KiB of dynamic LDS must be opted into, per device and per kernel variant
    for (const void* f : {reinterpret_cast<const void*>(&k_carrier_pruned<THR_IN_U8, false>),
                          reinterpret_cast<const void*>(&k_carrier_pruned<THR_IN_U8, true>),
                          reinterpret_cast<const void*>(&k_carrier_pruned<THR_IN_C64, false>),
                          reinterpret_cast<const void*>(&k_carrier_pruned<THR_IN_C64, true>)}) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int st = 0; st < 2; ++st)
            for (int d = 0; d < 3; ++d) {     // dump, windowed, all bins
                hipError_t e = hipFuncSetAttribute(
                    reinterpret_cast<const void*>(carrier_variant(fmt, st, d == 0, d == 2)),
                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_carrier_16k(int fmt, const void* samples, int n_blocks, const DevCfg& cfg,
                              const float2* tables, const float2* twn, CarStats* stats,
                              float2* dump_fft, int grid, hipStream_t stream) {
    if (cfg.car_prune && dump_fft == nullptr) {
        typedef void (*pruned_fn)(const void*, int, DevCfg, const cpx*, const cpx*, CarStats*);
        const bool shifted = cfg.car_prune == 2;
        pruned_fn fn = fmt == THR_IN_U8
                           ? (shifted ? &k_carrier_pruned<THR_IN_U8, true> : &k_carrier_pruned<THR_IN_U8, false>)
                           : (shifted ? &k_carrier_pruned<THR_IN_C64, true> : &k_carrier_pruned<THR_IN_C64, false>);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                           reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn), stats);
        return hipGetLastError();
    }
    // (the whole spectrum as window: win_lo 0, every bin -- the reference's default '0--1')
    carrier_fn fn = carrier_variant(fmt, cfg.car_want_std != 0, dump_fft != nullptr,
                                    cfg.win_lo == 0 && cfg.win_count == N);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, n_blocks, cfg,
                       reinterpret_cast<const cpx*>(tables), stats, reinterpret_cast<cpx*>(dump_fft));
    return hipGetLastError();
}

hipError_t launch_fit(int n_blocks, const DevCfg& cfg, const CarStats* stats,
                      const long long* block_idx, ShiftParams* shifts, int* work_list,
                      int* work_count, thr_record* records, CorrStats* corr_stats,
                      hipStream_t stream, const double* forced_offset) {
    hipLaunchKernelGGL(k_fit, dim3((n_blocks * 8 + 63) / 64), dim3(64), 0, stream, n_blocks, cfg,
                       stats, block_idx, shifts, work_list, work_count, records, corr_stats, forced_offset);
    return hipGetLastError();
}

hipError_t launch_finish(int n_records, const DevCfg& cfg, const CorrStats* corr_stats,
                         thr_record* records, int* work_count, hipStream_t stream,
                         const CorrStats* seg_stats) {
    hipLaunchKernelGGL(k_finish, dim3((n_records + 255) / 256), dim3(256), 0, stream, n_records,
                       cfg, corr_stats, records, work_count, seg_stats);
    return hipGetLastError();
}

int compact_tiles(int n) { return (n + CMP_TILE - 1) / CMP_TILE; }

hipError_t launch_compact(const thr_record* in, int n, thr_record* out, int* n_out,
                          int* tile_scratch, hipStream_t stream) {
    const int tiles = compact_tiles(n);
    if (tiles > 0)
        hipLaunchKernelGGL(k_compact_count, dim3(tiles), dim3(CMP_T), 0, stream, in, n, tile_scratch);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, stream, tile_scratch, tiles, n_out);
    if (tiles > 0)
        hipLaunchKernelGGL(k_compact_scatter, dim3(tiles), dim3(CMP_T), 0, stream, in, n,
                           tile_scratch, out);
    return hipGetLastError();
}

}  // namespace thr
