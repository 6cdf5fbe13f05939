// LDS-resident fused detection kernels for block_len = 16384 on gfx950.
//
// One 512-thread workgroup owns one IQ block at a time; the block's 16384
// complex samples live in LDS (136 KiB padded) between the three register
// passes of a 16 x 32 x 32 decimation-in-frequency FFT.  The spectrum leaves
// the last pass in digit-reversed order *in registers*, which is exactly the
// order the mirrored decimation-in-time inverse consumes, so the template
// product and the first inverse pass need no LDS round trip and no reordering.
//
//   k_carrier   : u8/c64 -> FFT#1 -> |X|^2 sum, windowed first-max, 7-bin
//                 neighbourhood                (signal_utils.py:21-25,
//                 carrier_detect.py:99-154)
//   k_fit       : noise/threshold verdict + Dirichlet LSQ fit + shift phasors
//                 (carrier_detect.py:61-115, carrier_sync.py:150-196)
//   k_correlate : shift -> FFT#2 -> x conj(T) -> IFFT -> |.|^2 windowed
//                 argmax -> noise/threshold -> log-parabola -> record
//                 (carrier_sync.py:222-238, soa_estimator.py:78-170)
#include <hip/hip_runtime.h>

#include <cstdlib>

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
__device__ __forceinline__ void phasor_table(const ShiftParams* __restrict__ sp,
                                             const cpx* __restrict__ twn, int t, cpx* sc_ph) {
    if (t < 73) {
        const int si = sp->si_mod;
        const float sf = sp->sf_over_n;
        const int m = t < 64 ? 2 * t : t < 72 ? 128 * (t - 64) : 1;
        const int q = (si * m) & (N - 1);
        const cpx wq = cconj(twn[q]);  // exp(+2 pi i q / N)
        float sn, cs;
        // |2 pi sf m| <= 2 pi * (0.5 / N) * 896 = 0.172 rad
        sincos_small(6.283185307179586f * (sf * float(m)), &sn, &cs);
        cpx v = cmul(wq, cpx{cs, sn});
        if (t >= 64 && t < 72) v = cmul(v, cpx{sp->c0.x, sp->c0.y});
        sc_ph[t] = v;
    }
}
__device__ __forceinline__ void thread_phasor(const cpx* sc_ph, int t, cpx (&p)[2]) {
    p[0] = cmul(sc_ph[64 + (t >> 6)], sc_ph[t & 63]);
    p[1] = cmul(p[0], sc_ph[72]);
}

// MULTI: more than one template -- the shifted spectrum is parked in a
// per-workgroup global scratch row (L2-resident) instead of 64 live VGPRs.
// RLO / RHI (>= 0; -1, -1 = no assumption): the peak search visits the lags in 16 rows of 1024.
// A variant with RLO, RHI is launched only when rows < RLO and rows > 15 - RHI lie entirely
// outside the unique window [corr_lo, corr_hi) (with one lag of margin for the peak's
// neighbours) and rows RLO + 1 .. 14 - RHI entirely inside it (launch_correlate_16k checks):
// the outside rows then cost nothing -- not even their share of pass C, whose unused outputs
// the compiler drops -- and the inside rows skip the window test.  Uniform run-time branches for
// the same purpose cost more schedule than they save (profiles/README.md); the two geometries
// instantiated are BASELINE's (history 4096, 1023-sample template: 1, 2) and the example
// detector.cfg's (history 4920, 4914-sample template: 0, 4); any other runs the generic form.
template <int FMT, bool WANT_STD, bool MULTI, bool DUMP, int RLO = -1, int RHI = -1>
__global__ __launch_bounds__(NT) void k_correlate(
    const void* __restrict__ samples, DevCfg cfg, const cpx* __restrict__ tables,
    const cpx* __restrict__ twn, const f4* __restrict__ tspec,
    const ShiftParams* __restrict__ shifts, const int* __restrict__ work_list,
    const int* __restrict__ work_count, CorrStats* __restrict__ corr_stats,
    thr_record* __restrict__ records, f4* __restrict__ xhat_scratch,
    cpx* __restrict__ dump_xhat, cpx* __restrict__ dump_corr, int dump_template) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cpx* lds = reinterpret_cast<cpx*>(smem_raw);
    unsigned char* sc_red = reinterpret_cast<unsigned char*>(lds + OFF_S);

    load_tables(lds, tables);
    __syncthreads();
    // dev knob (THR_PRIO): waves w and w+4 share a SIMD and the older one finishes every phase
    // ~35 % earlier; raising either half's priority was measured to change nothing here
    if (cfg.prio_mode == 1 && threadIdx.x >= NT / 2) __builtin_amdgcn_s_setprio(1);
    if (cfg.prio_mode == 2 && threadIdx.x < NT / 2) __builtin_amdgcn_s_setprio(1);
    const size_t blk_bytes = cfg.blk_stride;  // dense: N * sample size; raw streams: 2 (N - H)
    const int n_work = *work_count;
    int parity = 0;
#ifdef THR_DEV_NOBAR
    // dev: with the loop barriers gone, offset the waves once so that they stay out of phase:
    // THR_STAGGER = 10 + k: wave w starts w * k * 512 cycles late; 20 + k: waves 4-7 start
    // k * 512 cycles late (the partner of each SIMD's older wave)
    {
        const int w = threadIdx.x >> 6;
        int n = 0;
        if (cfg.stagger >= 20) n = w >= 4 ? cfg.stagger - 20 : 0;
        else if (cfg.stagger >= 10) n = w * (cfg.stagger - 10);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif

    RawSamples<FMT> cur;
    cpx p[2] = {cpx{0.f, 0.f}, cpx{0.f, 0.f}};
    int b_next = int(blockIdx.x) < n_work ? work_list[blockIdx.x] : 0;
    // work-list entry two iterations ahead, so the sample prefetch never waits on an index load
    int b_next2 = int(blockIdx.x + gridDim.x) < n_work ? work_list[blockIdx.x + gridDim.x] : 0;
    cpx* sc_ph = reinterpret_cast<cpx*>(sc_red + PH_OFF);
    if (int(blockIdx.x) < n_work) {
        cur.load(static_cast<const unsigned char*>(samples) + size_t(b_next) * blk_bytes,
                 opaque_tid());
        phasor_table(shifts + b_next, twn, opaque_tid(), sc_ph);
    }
    __syncthreads();
    // Work distribution.  Static: workgroup g takes entries g, g + G, g + 2G, ...  Dynamic
    // (cfg.dyn_sched): the first two entries are static (their prefetches are already in
    // flight), every later one comes from a global counter, fetched by thread 0 two
    // iterations ahead and handed to the workgroup through LDS across the pass-1 barrier --
    // a CU that runs a few percent slower then simply takes fewer blocks instead of making
    // the whole launch wait for its last one.
    const bool dyn = cfg.dyn_sched != 0;
    int* dyn_ctr = const_cast<int*>(work_count) + 1;
    int* sc_dyn = reinterpret_cast<int*>(sc_red + 768);
    int wi_nxt = int(blockIdx.x + gridDim.x);
    for (int wi = blockIdx.x, iter = 0; wi < n_work; ++iter) {
#ifdef THR_TIMELINE
        const bool tl_on = blockIdx.x == 0 && iter == 3 && cfg.timeline != nullptr;
#endif
        THR_STAMP(0);
        const int b = b_next;
        const int t = opaque_tid();
        const ShiftParams* sp = shifts + b;
        int wi_dyn = 0;
        if (dyn && t == 0) wi_dyn = 2 * int(gridDim.x) + atomicAdd(dyn_ctr, 1);
        // next block's samples: issued now, consumed one iteration later.  (Loading them after
        // pass 1 into the registers it has just consumed -- no second set, no copies -- is what the
        // carrier kernels do (-5 %); here it measured +0.5 %.)
        RawSamples<FMT> nxt = cur;
        const bool more = wi_nxt < n_work;
        if (more) {
            b_next = b_next2;
            nxt.load(static_cast<const unsigned char*>(samples) + size_t(b_next) * blk_bytes, t);
            if (!dyn && wi_nxt + int(gridDim.x) < n_work) b_next2 = work_list[wi_nxt + gridDim.x];
        }

        THR_STAMP(1);
        // (the previous block's pass-C LDS reads all precede its reduction barrier)
        // (multi-template: 64 more live VGPRs for the spectrum -- the L2-table path would spill)
        // passes 1 and B take their twiddles from the L2 table
        const cpx* gtw = static_cast<const cpx*>(cfg.gtw);
        thread_phasor(sc_ph, t, p);   // (table of THIS block: written one iteration ago, two barriers back)
        fwd_pass1<true, true>(lds, cur, sp->rpow, p[0], p[1], nullptr, gtw);
        cur = nxt;
        THR_STAMP(2);
        THR_STAMP(3);
        if (dyn && t == 0) *sc_dyn = wi_dyn;
        THR_LOOP_BARRIER();
        // (sc_dyn is rewritten only after two more barriers: every thread has read it by then)
        const int wi_nxt2 = dyn ? *sc_dyn : wi_nxt + int(gridDim.x);
        if (dyn && wi_nxt2 < n_work) b_next2 = work_list[wi_nxt2];
        wi = wi_nxt;
        wi_nxt = wi_nxt2;
        // the next block's phasor table: every thread has read this block's table before the
        // barrier above, and reads the new one only after the two barriers that follow
        if (more) phasor_table(shifts + b_next, twn, t, sc_ph);
#ifdef THR_DEV_ABLATE
        if (cfg.stagger >= 2 && threadIdx.x >= NT / 2) {
            if (cfg.stagger == 2) __builtin_amdgcn_s_sleep(16);
            if (cfg.stagger == 3) __builtin_amdgcn_s_sleep(32);
            if (cfg.stagger == 4) __builtin_amdgcn_s_sleep(64);
        }
#endif
        THR_STAMP(4);
        THR_ABLATE_AT(11, continue);
        // rows k1 = 2w, 2w+1 belong to wave w through passes 2, 3, A and B: no barriers
        fwd_pass2(lds);
        __builtin_amdgcn_sched_barrier(0);  // keep the passes' register working sets apart
        THR_STAMP(5);
        THR_ABLATE_AT(12, { __syncthreads(); continue; });
        cpx xh[R3];
        fwd_pass3(lds, xh);
        THR_STAMP(6);
#ifdef THR_DEV_ABLATE
        if (cfg.ablate == 13) {
            float acc = 0;
#pragma unroll
            for (int i = 0; i < R3; ++i) acc += xh[i].x + xh[i].y;
            if (acc == 1.2345f) records[b].reserved = 1;
            __syncthreads();
            continue;
        }
#endif

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
        // THR_MULTI_PARK (dev A/B): park the shifted spectrum in a per-workgroup global scratch
        // row between templates instead of keeping its 64 VGPRs live (measured slower)
#ifdef THR_MULTI_PARK
        constexpr bool PARK = MULTI;
#else
        constexpr bool PARK = false;
#endif
        f4* park = nullptr;
        if constexpr (PARK) {
            park = xhat_scratch + size_t(blockIdx.x) * (N / 2) + t;
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                park[j * NT] = f4{xh[brev(2 * j, R3)].x, xh[brev(2 * j, R3)].y,
                                      xh[brev(2 * j + 1, R3)].x, xh[brev(2 * j + 1, R3)].y};
            });
        }

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
            if constexpr (PARK) park = xhat_scratch + size_t(blockIdx.x) * (N / 2) + t;
            cpx z[R3];
            static_for<R3 / 2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const f4 q = tq[j];
                cpx x0, x1;
                if constexpr (PARK) {
                    const f4 xx = park[j * NT];  // own writes: program order suffices
                    x0 = cpx{xx.x, xx.y};
                    x1 = cpx{xx.z, xx.w};
                } else {
                    x0 = xh[brev(2 * j, R3)];
                    x1 = xh[brev(2 * j + 1, R3)];
                }
                cmul2(x0, cpx{q.x, q.y}, x1, cpx{q.z, q.w}, z[brev(2 * j, R3)],
                      z[brev(2 * j + 1, R3)]);
            });
            // pass A overwrites exactly the chunk this thread read in pass 3 (or, for
            // tpl > 0, rows whose pass-C readers are behind the previous reduction barrier)
            inv_passA(lds, z);
            __builtin_amdgcn_sched_barrier(0);
            THR_STAMP(7);
            THR_ABLATE_AT(14, { __syncthreads(); continue; });
            // (several templates: the table twiddles in two halves after the butterfly -- the spectrum
            // stays live beside this pass, and all 32 requested ahead of it spill; -3.3 % against the
            // LDS-product form this kernel used before)
            inv_passB<true, MULTI>(lds, gtw);
            THR_STAMP(8);
            THR_LOOP_BARRIER();
            THR_STAMP(9);
            THR_ABLATE_AT(15, continue);
            if constexpr (MULTI) {
                if (tpl + 1 < n_tpl) {
                    const f4* ts = tspec + size_t(tpl + 1) * (N / 2) + opaque_tid();
                    static_for<R3 / 2>([&](auto J) { tq[decltype(J)::value] = ts[decltype(J)::value * NT]; });
                }
            }
            cpx c0[R1], c1[R1];
            inv_passC(lds, c0, c1);
            THR_STAMP(10);
#ifdef THR_DEV_ABLATE
            if (cfg.ablate == 16) {
                float acc = 0;
#pragma unroll
                for (int i = 0; i < R1; ++i) acc += c0[i].x + c0[i].y + c1[i].x + c1[i].y;
                if (acc == 1.2345f) records[b].reserved = 1;
                __syncthreads();
                continue;
            }
#endif

            // ---- |corr|^2, windowed first-max, optional std sums
            // The maximum first, the lag afterwards: per thread one v_max3 per two lags, per wave
            // a DPP max; then the lanes that hold the wave's maximum name their first lag with
            // it (lags scanned downwards, so the lowest one sticks), a DPP min picks the wave's
            // first, and one 64-bit key per wave -- (power, -lag) -- goes through LDS.  (Tracking
            // (power, lag) per lag costs a compare and two selects each, 132 VALU slots per block
            // with the reduction; this form 85.)  NaN powers are never candidates (v_max and
            // v_cmp_eq ignore them), like the strict '>' of a running maximum.
            float sums[2] = {0.f, 0.f};     // sum |corr|, sum |corr|^2 over [0, corr_len): WANT_STD only
            float pw0[R1], pw1[R1];         // powers (the peak's neighbours are picked from them)
            float ew0[R1], ew1[R1];         // the same inside the unique window, -1 outside
            float tmax = -1.0f;
            const unsigned win_w = unsigned(cfg.corr_hi - cfg.corr_lo);
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
                    ew0[n1] = unsigned(n - cfg.corr_lo) < win_w ? pw0[n1] : -1.f;
                    ew1[n1] = unsigned(n + 1 - cfg.corr_lo) < win_w ? pw1[n1] : -1.f;
                    if constexpr (WANT_STD) {
                        if (n < cfg.corr_len) {
                            sums[1] += pw0[n1];
                            sums[0] += __builtin_amdgcn_sqrtf(pw0[n1]);
                        }
                        if (n + 1 < cfg.corr_len) {
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
            CorrStats* cs = corr_stats + size_t(b) * cfg.n_templates + tpl;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int delta = pk - 1 - (2 * t + e);        // want n1*1024 == delta + d
                const unsigned d = unsigned(-delta) & 1023u;    // d in [0,1024)
                const int n1s = (delta + int(d)) >> 10;
                // at most three threads of the workgroup are owners: with one template the 16-way
                // select runs under a branch that seven of the eight waves skip (-2 % kernel time);
                // inside the template loop the same branch costs +7 % (measured), so there the
                // select stays branch-free
#ifndef THR_NB_BRANCH
#define THR_NB_BRANCH (!MULTI)
#endif
                const bool owner = d < 3u && n1s >= 0 && n1s < R1;
                if (!THR_NB_BRANCH || owner) {
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

// ------------------------------------------------------------------ launchers
size_t lds_bytes_16k() { return LDS_BYTES; }

namespace {
typedef void (*correlate_fn)(const void*, DevCfg, const cpx*, const cpx*, const f4*,
                             const ShiftParams*, const int*, const int*, CorrStats*, thr_record*,
                             f4*, cpx*, cpx*, int);

#ifdef THR_DEV_MINIMAL  // compile-time experiments only: one variant each, fast rebuilds
correlate_fn correlate_variant(int, bool, bool, bool) {
    return &k_correlate<THR_IN_U8, false, false, false>;
}
#else
template <int FMT, bool STD, bool MULTI>
correlate_fn pick_correlate(bool dump) {
    return dump ? &k_correlate<FMT, STD, MULTI, true> : &k_correlate<FMT, STD, MULTI, false>;
}
template <int FMT>
correlate_fn pick_correlate2(bool want_std, bool multi, bool dump) {
    if (want_std)
        return multi ? pick_correlate<FMT, true, true>(dump) : pick_correlate<FMT, true, false>(dump);
    return multi ? pick_correlate<FMT, false, true>(dump) : pick_correlate<FMT, false, false>(dump);
}
correlate_fn correlate_variant(int fmt, bool want_std, bool multi, bool dump) {
    return fmt == THR_IN_U8 ? pick_correlate2<THR_IN_U8>(want_std, multi, dump)
                            : pick_correlate2<THR_IN_C64>(want_std, multi, dump);
}
#endif

// window geometries with a specialised peak search (no stddev term, no dumps)
struct RowGeom {
    int lo, hi;
};
constexpr RowGeom kRowGeoms[] = {{1, 2}, {0, 4}};
constexpr int kNumRowGeoms = int(sizeof(kRowGeoms) / sizeof(kRowGeoms[0]));

// true if rows < lo and > 15 - hi lie outside [corr_lo - 1, corr_hi] and rows lo+1 .. 14-hi inside
// [corr_lo, corr_hi)
bool geom_applies(const RowGeom& g, const DevCfg& cfg) {
    const bool low_out = g.lo == 0 || g.lo * S1 - 1 < cfg.corr_lo - 1;   // last lag of row lo - 1
    const bool high_out = g.hi == 0 || (16 - g.hi) * S1 > cfg.corr_hi;   // first lag of row 16 - hi
    const bool inside = (g.lo + 1) * S1 >= cfg.corr_lo && (15 - g.hi) * S1 <= cfg.corr_hi;
    return low_out && high_out && inside;
}

template <int FMT, bool MULTI>
correlate_fn geom_pick(int g) {
    return g == 0 ? &k_correlate<FMT, false, MULTI, false, kRowGeoms[0].lo, kRowGeoms[0].hi>
                  : &k_correlate<FMT, false, MULTI, false, kRowGeoms[1].lo, kRowGeoms[1].hi>;
}
correlate_fn geom_variant(int fmt, int g, bool multi) {
#ifdef THR_DEV_MINIMAL
    (void)fmt;
    (void)g;
    (void)multi;
    return nullptr;
#else
    if (fmt == THR_IN_U8) return multi ? geom_pick<THR_IN_U8, true>(g) : geom_pick<THR_IN_U8, false>(g);
    return multi ? geom_pick<THR_IN_C64, true>(g) : geom_pick<THR_IN_C64, false>(g);
#endif
}
static_assert(kNumRowGeoms == 2, "geom_variant() enumerates the geometries by hand");
}  // namespace

hipError_t prepare_16k_carrier();   // detect16k_carrier.hip

hipError_t prepare_16k() {
    // > 64 KiB of dynamic LDS must be opted into, per device and per kernel variant
    hipError_t e = prepare_16k_carrier();
    if (e != hipSuccess) return e;
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int st = 0; st < 2; ++st)
            for (int d = 0; d < 2; ++d)
                for (int m = 0; m < 2; ++m) {
                    e = hipFuncSetAttribute(
                        reinterpret_cast<const void*>(correlate_variant(fmt, st, m, d)),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                    if (e != hipSuccess) return e;
                }
    for (int fmt = 0; fmt < 2; ++fmt)
        for (int g = 0; g < kNumRowGeoms; ++g)
            for (int m = 0; m < 2; ++m) {
                correlate_fn fn = geom_variant(fmt, g, m != 0);
                if (fn == nullptr) continue;
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_correlate_16k(int fmt, const void* samples, const DevCfg& cfg,
                                const float2* tables, const float2* twn, const float4* tspec,
                                const ShiftParams* shifts, const int* work_list,
                                const int* work_count, CorrStats* corr_stats,
                                thr_record* records, float4* xhat_scratch,
                                float2* dump_xhat, float2* dump_corr, int dump_template, int grid,
                                hipStream_t stream) {
    const bool dump = dump_xhat != nullptr || dump_corr != nullptr;
    correlate_fn fn = correlate_variant(fmt, cfg.cor_want_std != 0, cfg.n_templates > 1, dump);
    static const bool no_geom = getenv("THR_NO_GEOM") != nullptr;   // dev A/B
    if (!dump && cfg.cor_want_std == 0 && cfg.ablate == 0 && !no_geom)
        for (int g = 0; g < kNumRowGeoms; ++g)
            if (geom_applies(kRowGeoms[g], cfg) && geom_variant(fmt, g, cfg.n_templates > 1) != nullptr) {
                fn = geom_variant(fmt, g, cfg.n_templates > 1);
                break;
            }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NT), LDS_BYTES, stream, samples, cfg,
                       reinterpret_cast<const cpx*>(tables), reinterpret_cast<const cpx*>(twn),
                       reinterpret_cast<const f4*>(tspec), shifts, work_list, work_count,
                       corr_stats, records,
                       reinterpret_cast<f4*>(xhat_scratch), reinterpret_cast<cpx*>(dump_xhat),
                       reinterpret_cast<cpx*>(dump_corr), dump_template);
    return hipGetLastError();
}

}  // namespace thr
