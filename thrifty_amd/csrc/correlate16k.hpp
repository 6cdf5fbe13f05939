// k_correlate: the LDS-resident shift -> FFT#2 -> x conj(T) -> IFFT -> windowed first-max kernel of
// one 16384-sample transform pair.  A header because two translation units instantiate it:
// detect16k.hip (block_len 16384: one work item per carrier-positive block) and detect_seg.hip
// (block_len > 16384, SEG: one work item per (block, overlap-save section) -- DESIGN.md section 3).
#pragma once
#include <hip/hip_runtime.h>

#include "detect_common.hpp"
#include "fft_regs.hpp"
#include "kernel_util.hpp"
#include "passes_w8.hpp"

namespace thr {

using namespace k16;

// =========================================================================
// K_B: shift + FFT#2 + matched filter + SoA
// =========================================================================
// Shift phasor c0 * exp(2 pi i s m / N) of a thread's samples m = 2t, 2t+1, t = 64 w + l:
//     p(2t) = [c0 exp(2 pi i s 128 w / N)] * [exp(2 pi i s 2 l / N)],   p(2t + 1) = p(2t) * exp(2 pi i s / N)
// -- 8 wave factors, 64 lane factors and the one-sample step: 73 exactly formed numbers per
// block (integer part of s through the root table, fractional part through a small-angle
// polynomial).  73 threads form one each for the NEXT block and park them in LDS; every thread then
// needs two LDS reads and two complex products instead of two sincosf + gathers of its own.
constexpr int PH_OFF = 896;   // bytes into the scratch area: [896, 896 + 73 * 8)
// `root_mask` = block_len - 1 (twn holds block_len roots; si_mod < block_len, m < 1024, so the
// product stays below 2^30 for every supported block length) and `c0` = the phasor of the work
// item's first sample: exp(-pi i s) for a 16384-sample block, exp(2 pi i s (start / NL - 1/2)) for
// a section that starts at sample `start` of a long block.
__device__ __forceinline__ void phasor_table(const ShiftParams* __restrict__ sp, float2 c0,
                                             const cpx* __restrict__ twn, int root_mask, int t,
                                             cpx* sc_ph) {
    if (t < 73) {
        const int si = sp->si_mod;
        const float sf = sp->sf_over_n;
        const int m = t < 64 ? 2 * t : t < 72 ? 128 * (t - 64) : 1;
        const int q = (si * m) & root_mask;
        const cpx wq = cconj(twn[q]);  // exp(+2 pi i q / N)
        float sn, cs;
        // |2 pi sf m| <= 2 pi * (0.5 / N) * 896 = 0.172 rad
        sincos_small(6.283185307179586f * (sf * float(m)), &sn, &cs);
        cpx v = cmul(wq, cpx{cs, sn});
        if (t >= 64 && t < 72) v = cmul(v, cpx{c0.x, c0.y});
        sc_ph[t] = v;
    }
}
__device__ __forceinline__ void thread_phasor(const cpx* sc_ph, int t, cpx (&p)[2]) {
    p[0] = cmul(sc_ph[64 + (t >> 6)], sc_ph[t & 63]);
    p[1] = cmul(p[0], sc_ph[72]);
}

// A work item packs (block of the batch) << 3 | (section of the block); sections exist only with SEG.
constexpr int kItemSegBits = 3;

// MULTI: more than one template (the shifted spectrum stays live in 64 VGPRs across the template loop).
// RLO / RHI (>= 0; -1, -1 = no assumption): the peak search visits the lags in 16 rows of 1024.
// A variant with RLO, RHI is launched only when rows < RLO and rows > 15 - RHI lie entirely
// outside the unique window [corr_lo, corr_hi) (with one lag of margin for the peak's
// neighbours) and rows RLO + 1 .. 14 - RHI entirely inside it (the launchers check):
// the outside rows then cost nothing -- not even their share of pass C, whose unused outputs
// the compiler drops -- and the inside rows skip the window test.  Uniform run-time branches for
// the same purpose cost more schedule than they save (profiles/README.md), so the variants are a
// small closed table, correlate16k_geom.hpp: RLO = 0 .. 2 x RHI = 0 .. 4 (RLO = 0 for sections) --
// BASELINE's history 4096 / 1023-sample template is (1, 2), the example detector.cfg (0, 4),
// BASELINE's 65536-sample blocks (0, 3) in every section; a window outside the table runs the
// generic form.
//
// SEG (block_len NL > 16384, detect_seg.hip): the correlation the reference keeps,
// ifft(X^ conj(T^))[:corr_len] with the template zero-padded to NL (soa_estimator.py:97-102), is a
// LINEAR correlation -- corr[l] = sum_{n < W} y[l + n] t[n], no lag below corr_len wraps -- so
// overlap-save sections it exactly: section g transforms the 16384 shifted samples from
// cfg.seg_start[g] on against the template zero-padded to 16384, and its lags [0, 16384 - W] are
// lags seg_start[g] + [0, 16384 - W] of the block.  A work item is one (block, section); it writes
// the windowed first-max of the lags it OWNS (cfg.seg_lo/hi, in section coordinates; with one valid
// lag either side for the peak's neighbours) and k_finish keeps the best section of the block.
// Nothing of a block ever leaves the CU but those 32 bytes per section.
template <int FMT, bool WANT_STD, bool MULTI, bool DUMP, int RLO = -1, int RHI = -1, bool SEG = false>
__global__ __launch_bounds__(NT) void k_correlate(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, CorrStats* __restrict__ corr_stats,
    cpx* __restrict__ dump_xhat, cpx* __restrict__ dump_corr, int dump_template) {
    static_assert(!(SEG && DUMP), "stage dumps of long blocks come from the unsectioned kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    const int n_seg = SEG ? cfg.n_seg : 1;
    const int n_work = *work_count * n_seg;   // work items
    const int root_mask = SEG ? cfg.block_len - 1 : N - 1;
    constexpr int kSampleBytes = FMT == THR_IN_U8 ? 2 : 8;
    int parity = 0;

    // work item wi -> (block << 3 | section); plain blocks: section 0
    auto item_at = [&](int wi) -> int {
        if constexpr (SEG) {
            const int e = wi / n_seg;
            return (work_list[e] << kItemSegBits) | (wi - e * n_seg);
        } else {
            return work_list[wi] << kItemSegBits;
        }
    };
    auto item_samples = [&](int item) -> const unsigned char* {
        const unsigned char* p =
            static_cast<const unsigned char*>(samples) + size_t(item >> kItemSegBits) * blk_bytes;
        if constexpr (SEG) p += size_t(cfg.seg_start[item & 7]) * kSampleBytes;
        return p;
    };
    auto item_phasor = [&](int item, int t, cpx* sc_ph) {
        const ShiftParams* sp = shifts + (item >> kItemSegBits);
        phasor_table(sp, SEG ? sp->segc0[item & 7] : sp->c0, twn, root_mask, t, sc_ph);
    };

    RawSamples<FMT> cur;
    cpx p[2] = {cpx{0.f, 0.f}, cpx{0.f, 0.f}};
    int it_next = int(blockIdx.x) < n_work ? item_at(blockIdx.x) : 0;
    // work-list entry two iterations ahead, so the sample prefetch never waits on an index load
    int it_next2 = int(blockIdx.x + gridDim.x) < n_work ? item_at(blockIdx.x + gridDim.x) : 0;
    cpx* sc_ph = reinterpret_cast<cpx*>(sc_red + PH_OFF);
    if (int(blockIdx.x) < n_work) {
        cur.load(item_samples(it_next), opaque_tid());
        item_phasor(it_next, opaque_tid(), sc_ph);
    }
    __syncthreads();
    // Work distribution: the first two items of workgroup g are static (g, g + G: their prefetches
    // are already in flight), every later one comes from a global counter, fetched by thread 0 two
    // iterations ahead and handed to the workgroup through LDS across the pass-1 barrier -- a CU
    // that runs a few percent slower then simply takes fewer items instead of making the whole
    // launch wait for its last one.
    int* dyn_ctr = const_cast<int*>(work_count) + 1;
    int* sc_dyn = reinterpret_cast<int*>(sc_red + 768);
    int wi_nxt = int(blockIdx.x + gridDim.x);
    for (int wi = blockIdx.x, iter = 0; wi < n_work; ++iter) {
#ifdef THR_DEV
        const bool tl_on = blockIdx.x == 0 && iter == 3 && cfg.timeline != nullptr;
#endif
        THR_STAMP(0);
        const int item = it_next;
        const int b = item >> kItemSegBits;
        const int seg = item & 7;
        const int t = opaque_tid();
        const ShiftParams* sp = shifts + b;
        int wi_dyn = 0;
        if (t == 0) wi_dyn = 2 * int(gridDim.x) + atomicAdd(dyn_ctr, 1);
        // next item's samples: issued now, consumed one iteration later.  (Loading them after
        // pass 1 into the registers it has just consumed -- no second set, no copies -- is what the
        // carrier kernels do (-5 %); here it measured +0.5 %.)
        RawSamples<FMT> nxt = cur;
        const bool more = wi_nxt < n_work;
        if (more) {
            it_next = it_next2;
            nxt.load(item_samples(it_next), t);
        }

        THR_STAMP(1);
        // (the previous item's pass-C LDS reads all precede its reduction barrier)
        // passes 1 and B take their twiddles from the L2 table
        const cpx* gtw = static_cast<const cpx*>(cfg.gtw);
        thread_phasor(sc_ph, t, p);   // (table of THIS item: written one iteration ago, two barriers back)
        fwd_pass1<true, true>(lds, cur, sp->rpow, p[0], p[1], nullptr, gtw);
        cur = nxt;
        THR_STAMP(2);
        THR_STAMP(3);
        if (t == 0) *sc_dyn = wi_dyn;
        __syncthreads();
        // (sc_dyn is rewritten only after two more barriers: every thread has read it by then)
        const int wi_nxt2 = *sc_dyn;
        if (wi_nxt2 < n_work) it_next2 = item_at(wi_nxt2);
        wi = wi_nxt;
        wi_nxt = wi_nxt2;
        // the next item's phasor table: every thread has read this item's table before the
        // barrier above, and reads the new one only after the two barriers that follow
        if (more) item_phasor(it_next, t, sc_ph);
        THR_STAMP(4);
        // rows k1 = 2w, 2w+1 belong to wave w through passes 2, 3, A and B: no barriers
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);  // keep the passes' register working sets apart
        THR_STAMP(5);
        cpx xh[R3];
        fwd_pass3(lds, xh);
        THR_STAMP(6);

        const int kbase = (t >> 5) + 16 * (t & 31);
        // (sum |X^|^2, which the correlation noise estimate needs (soa_estimator.py:108-120), is
        // N sum |x|^2 whatever the shift -- the phasor has unit modulus -- and the carrier stage
        // has that sum already: k_fit hands it to k_finish, nothing is summed here)
        if constexpr (DUMP) {
            if (dump_xhat != nullptr) {
                cpx* out = dump_xhat + size_t(b) * N;
                static_for<R3>([&](auto K) {
                    constexpr int k3 = decltype(K)::value;
                    out[kbase + 512 * k3] = xh[brev(k3, R3)];
                });
            }
        }

        // the lags this item searches [w_lo, w_hi) and sums [s_lo, s_hi) (stddev term), item-local
        const int w_lo = SEG ? cfg.seg_lo[seg] : cfg.corr_lo;
        const unsigned win_w = unsigned((SEG ? cfg.seg_hi[seg] : cfg.corr_hi) - w_lo);
        const int s_lo = SEG ? cfg.seg_sum_lo[seg] : 0;
        const unsigned sum_w = unsigned((SEG ? cfg.seg_sum_hi[seg] : cfg.corr_len) - s_lo);

        const int n_tpl = MULTI ? cfg.n_templates : 1;
        // template spectrum of this thread's 32 bins (16 x float4, L2-resident).  With several
        // templates the NEXT template's slice is requested before pass C of the current one, so
        // that its L2 latency hides under pass C, the statistics and the reduction.
        f4 tq[R3 / 2];
        {
            const char* ts = reinterpret_cast<const char*>(tspec);
            const unsigned off = unsigned(opaque_tid()) * 16u;
            static_for<R3 / 2>([&](auto J) {
                tq[decltype(J)::value] =
                    *reinterpret_cast<const f4*>(ts + (off + unsigned(decltype(J)::value * (NT * 16))));
            });
        }
        for (int tpl = 0; tpl < n_tpl; ++tpl) {
            // ---- X * conj(T)/N in digit-reversed register order
            const int t = opaque_tid();  // re-derive per template: keeps LICM off the loop body
            cpx z[R3];
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = tq[j];
                cmul2(xh[brev(2 * j, R3)], cpx{q.x, q.y}, xh[brev(2 * j + 1, R3)], cpx{q.z, q.w},
                      z[brev(2 * j, R3)], z[brev(2 * j + 1, R3)]);
            });
            // pass A overwrites exactly the chunk this thread read in pass 3 (or, for
            // tpl > 0, rows whose pass-C readers are behind the previous reduction barrier)
            inv_passA(lds, z);
            __builtin_amdgcn_sched_barrier(0);
            THR_STAMP(7);
            // (several templates: the table twiddles in two halves after the butterfly -- the spectrum
            // stays live beside this pass, and all 32 requested ahead of it spill; -3.3 % against the
            // LDS-product form this kernel used before)
            inv_passB<true, MULTI>(lds, gtw);
            THR_STAMP(8);
            __syncthreads();
            THR_STAMP(9);
            if constexpr (MULTI) {
                if (tpl + 1 < n_tpl) {
                    const f4* ts = tspec + size_t(tpl + 1) * (N / 2) + opaque_tid();
                    static_for<R3 / 2>([&](auto J) { tq[decltype(J)::value] = ts[decltype(J)::value * NT]; });
                }
            }
            cpx c0[R1], c1[R1];
            inv_passC(lds, c0, c1);
            THR_STAMP(10);

            // ---- |corr|^2, windowed first-max, optional std sums
            // The maximum first, the lag afterwards: per thread one v_max3 per two lags, per wave
            // a DPP max; then the lanes that hold the wave's maximum name their first lag with
            // it (lags scanned downwards, so the lowest one sticks), a DPP min picks the wave's
            // first, and one 64-bit key per wave -- (power, -lag) -- goes through LDS.  (Tracking
            // (power, lag) per lag costs a compare and two selects each, 132 VALU slots per block
            // with the reduction; this form 85.)  NaN powers are never candidates (v_max and
            // v_cmp_eq ignore them), like the strict '>' of a running maximum.
            float sums[2] = {0.f, 0.f};     // sum |corr|, sum |corr|^2 over the summed lags: WANT_STD only
            float pw0[R1], pw1[R1];         // powers (the peak's neighbours are picked from them)
            float ew0[R1], ew1[R1];         // the same inside the unique window, -1 outside
            float tmax = -1.0f;
            constexpr bool GEOM = RLO >= 0 && RHI >= 0 && !WANT_STD;
            static_for<R1>([&](auto K) {
                constexpr int n1 = decltype(K)::value;
                if constexpr (GEOM && (n1 < RLO || n1 > 15 - RHI)) {   // row outside the window
                    pw0[n1] = pw1[n1] = 0.f;
                    ew0[n1] = ew1[n1] = -1.f;
                    return;
                }
                pw0[n1] = cnorm(c0[brev(n1, R1)]);
                pw1[n1] = cnorm(c1[brev(n1, R1)]);
                if constexpr (GEOM && n1 > RLO && n1 < 15 - RHI) {     // row inside the window
                    ew0[n1] = pw0[n1];
                    ew1[n1] = pw1[n1];
                } else {
                    const int n = n1 * S1 + 2 * t;
                    ew0[n1] = unsigned(n - w_lo) < win_w ? pw0[n1] : -1.f;
                    ew1[n1] = unsigned(n + 1 - w_lo) < win_w ? pw1[n1] : -1.f;
                    if constexpr (WANT_STD) {
                        if (unsigned(n - s_lo) < sum_w) {
                            sums[1] += pw0[n1];
                            sums[0] += __builtin_amdgcn_sqrtf(pw0[n1]);
                        }
                        if (unsigned(n + 1 - s_lo) < sum_w) {
                            sums[1] += pw1[n1];
                            sums[0] += __builtin_amdgcn_sqrtf(pw1[n1]);
                        }
                    }
                }
                tmax = __builtin_fmaxf(tmax, __builtin_fmaxf(ew0[n1], ew1[n1]));
            });
            const float wmax = wave_max_f32(tmax);
            int first = 63;    // 2 n1 + e of the thread's first lag with the wave's maximum
            static_for<R1>([&](auto K) {
                constexpr int n1 = R1 - 1 - decltype(K)::value;
                if constexpr (GEOM && (n1 < RLO || n1 > 15 - RHI)) return;
                first = ew1[n1] == wmax ? 2 * n1 + 1 : first;
                first = ew0[n1] == wmax ? 2 * n1 : first;
            });
            const unsigned lag = first == 63 ? 0xFFFFFFFFu
                                             : unsigned((first >> 1) * S1 + 2 * t + (first & 1));
            const unsigned wlag = wave_min_u32(lag);
            // (no lag of this wave inside the window: wmax = -1, and key 0 loses to every other)
            unsigned long long best =
                wmax < 0.f ? 0ull
                           : ((unsigned long long)__float_as_uint(wmax) << 32) | (0xFFFFFFFFu - wlag);
            double tot[2] = {0, 0};
            THR_STAMP(11);
            if constexpr (WANT_STD)
                block_reduce<2, NT / 64, true>(sums, tot, best, sc_red, parity);
            else
                block_reduce_wave_keys<NT / 64>(best, sc_red, parity);
            THR_STAMP(12);
            parity ^= 1;
            const int pk = int(0xFFFFFFFFu - unsigned(best & 0xFFFFFFFFu));
            // |corr[pk-1..pk+1]|^2 for the log-parabola: lag n = n1*1024 + 2t + e, so this
            // thread holds pk-1+d iff (2t + e - pk + 1 - d) mod 1024 == 0; the three owners
            // store straight into the per-record stats (finalised by k_finish).
            CorrStats* cs = corr_stats + (size_t(b) * cfg.n_templates + tpl) * n_seg + seg;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int delta = pk - 1 - (2 * t + e);        // want n1*1024 == delta + d
                const unsigned d = unsigned(-delta) & 1023u;    // d in [0,1024)
                const int n1s = (delta + int(d)) >> 10;
                // at most three threads of the workgroup are owners: with one template the 16-way
                // select runs under a branch that seven of the eight waves skip (-2 % kernel time);
                // inside the template loop the same branch costs +7 % (measured), so there the
                // select stays branch-free
                const bool owner = d < 3u && n1s >= 0 && n1s < R1;
                if (MULTI || owner) {
                    float val = 0.f;
                    static_for<R1>([&](auto K) {
                        constexpr int n1 = decltype(K)::value;
                        val = (n1s == n1) ? (e ? pw1[n1] : pw0[n1]) : val;
                    });
                    if (owner) cs->m2[d] = val;
                }
            }
            if constexpr (DUMP) {
                if (dump_corr != nullptr && tpl == dump_template) {
                    cpx* out = dump_corr + size_t(b) * N;
                    static_for<R1>([&](auto K) {
                        constexpr int n1 = decltype(K)::value;
                        reinterpret_cast<f4*>(out + n1 * S1)[t] =
                            f4{c0[brev(n1, R1)].x, c0[brev(n1, R1)].y, c1[brev(n1, R1)].x,
                               c1[brev(n1, R1)].y};
                    });
                }
            }
            if (t == 0) {
                cs->pm2 = __uint_as_float(unsigned(best >> 32));
                cs->pk = pk;
                cs->sum_mag = WANT_STD ? (float)tot[0] : 0.f;
                cs->sum_mag2 = WANT_STD ? (float)tot[1] : 0.f;
            }
            THR_STAMP(13);
        }
    }
}

// launch signature shared by every instantiation
typedef void (*correlate_fn)(const void*, DevCfg, const cpx*, const cpx*, const f4*,
                             const ShiftParams*, const int*, const int*, CorrStats*, cpx*, cpx*, int);

// true if rows < lo and > 15 - hi lie outside [w_lo - 1, w_hi] and rows lo+1 .. 14-hi inside [w_lo, w_hi)
inline bool row_geom_applies(int lo, int hi, int w_lo, int w_hi) {
    const bool low_out = lo == 0 || lo * S1 - 1 < w_lo - 1;      // last lag of row lo - 1
    const bool high_out = hi == 0 || (16 - hi) * S1 > w_hi;      // first lag of row 16 - hi
    const bool inside = (lo + 1) * S1 >= w_lo && (15 - hi) * S1 <= w_hi;
    return low_out && high_out && inside;
}

}  // namespace thr
