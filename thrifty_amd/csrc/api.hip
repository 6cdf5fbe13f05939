// C ABI of the detection engine (see include/thrifty_hip.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include <sys/mman.h>
#include <utility>
#include <vector>

#include "detect_common.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace

namespace thr {
// error reporting for the other translation units of the library (identify.hip)
int fail_msg(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
// No exception leaves the C ABI (include/thrifty_hip.h): every entry point that allocates or starts
// a thread is a function-try-block ending here.  Host memory exhaustion and a refused thread
// (std::system_error: a pids / thread limit, eight ranks on one host) become THR_ERR_DEVICE.
int on_exception(const char* who) noexcept {
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            return fail_msg(THR_ERR_DEVICE, "%s: out of host memory", who);
        } catch (const std::exception& e) {
            return fail_msg(THR_ERR_DEVICE, "%s: %s", who, e.what());
        } catch (...) {
            return fail_msg(THR_ERR_DEVICE, "%s: unknown C++ exception", who);
        }
    } catch (...) {         // (the message itself could not be stored)
        return THR_ERR_DEVICE;
    }
}
}  // namespace thr

namespace {

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(THR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,                \
                        hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

// Plain iterative radix-2 FFT in double, host side, setup only (template
// spectrum; the reference does this once in float64 too, soa_estimator.py:68-72).
void host_fft(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    const double pi = 3.14159265358979323846;
    for (size_t len = 2; len <= n; len <<= 1) {
        // exact-ish twiddles: evaluate each directly (no recurrence drift)
        std::vector<std::complex<double>> w(len / 2);
        for (size_t k = 0; k < len / 2; ++k)
            w[k] = std::complex<double>(std::cos(2 * pi * double(k) / double(len)),
                                        -std::sin(2 * pi * double(k) / double(len)));
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w[k];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

float2 unit_root(long long num, long long den) {  // exp(-2 pi i num/den), exact reduction
    const double pi = 3.14159265358979323846;
    num %= den;
    if (num < 0) num += den;
    const double a = 2 * pi * double(num) / double(den);
    return float2{float(std::cos(a)), float(-std::sin(a))};
}

struct EventPair {
    hipEvent_t a, b;
};

}  // namespace

// thr_input_window(): a caller mapping (the input file) that the host entry points read
// sequentially.  Library threads keep a bounded stretch of it page-locked around the read position,
// one segment (128 MiB) at a time: populators map the pages of the segments ahead, a locking worker
// hipHostRegister()s them up to kAhead segments in front of the chunk copies, an unlocking worker
// hipHostUnregister()s what the copies have left behind.  The copies are then asynchronous DMA out
// of the page cache (they return at once instead of occupying the calling thread while the runtime
// stages pageable memory), the locking -- 5 ms per GiB on mapped pages, 17 ms per GiB to unlock --
// runs beside the caller instead of in front of it, and never more than 2 x kAhead segments are
// locked whatever the size of the file.  (Round 5: locking and unlocking on ONE thread filled a
// whole run -- the caller waited for locks queued behind unlocks; see profiles/README.md.)
struct InputWindow {
    static constexpr size_t kSegDefault = size_t(128) << 20;
    static constexpr size_t kAheadBytes = size_t(1) << 30;   // the worker runs at most this far ahead of `consumed`
    size_t kSeg = kSegDefault;                // bytes per segment (thr_input_window_ex: tests shrink it)
    size_t kAhead = 8;                        // segments the worker may run ahead of `consumed` (1 GiB)
    uintptr_t base = 0, end = 0;              // page-aligned span; base == 0: no window
    size_t n_seg = 0;
    size_t reg_lo = 0, reg_hi = 0;            // segments [reg_lo, reg_hi) are locked now
    size_t consumed = 0;                      // segments below this one are not needed any more
    bool stop = false, failed = false;
    bool draining = false;                    // release_all(): nothing more is locked, everything locked is let go
    int device = 0;
    std::thread worker, unlocker;
    std::mutex mu;
    std::condition_variable cv;
    // page-table population runs in front of the locking, on threads of its own: locking pages
    // that are already mapped goes at ~100 GB/s, faulting them in one by one inside
    // hipHostRegister at ~30 (measured), and the fabric copies run at 56
    static constexpr int kPopulators = 3;      // default; thr_input_window_ex sizes it (ranks share the host's CPUs)
    std::vector<std::thread> populators;
    std::vector<unsigned char> populated;      // per segment: its pages are mapped
    size_t pop_next = 0;                       // next segment a populator takes
    // where the window's threads spend their time (thr_debug_window_times; seconds, under `mu`)
    double t_populate = 0, t_register = 0, t_unregister = 0, t_acquire = 0;
    size_t n_acquire_waits = 0, n_pageable = 0;
    static double now_s() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    void populate_run() {
        std::unique_lock<std::mutex> lk(mu);
        while (!stop) {
            if (pop_next < consumed) pop_next = consumed;
            if (!draining && pop_next < n_seg && pop_next < consumed + kAhead) {
                const size_t sgm = pop_next++;
                lk.unlock();
                void* at = reinterpret_cast<void*>(seg_lo(sgm));
                const double t0 = now_s();
#ifdef MADV_POPULATE_READ
                int rc = madvise(at, seg_len(sgm), MADV_POPULATE_READ);
#else
                int rc = -1;
#endif
                if (rc != 0) {     // older kernels: touch a byte of every page
                    volatile const unsigned char* q = static_cast<const unsigned char*>(at);
                    unsigned acc = 0;
                    for (size_t i = 0; i < seg_len(sgm); i += 4096) acc += q[i];
                    (void)acc;
                }
                lk.lock();
                t_populate += now_s() - t0;
                populated[sgm] = 1;
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
    }

    uintptr_t seg_lo(size_t s) const { return base + s * kSeg; }
    size_t seg_len(size_t s) const { return size_t(std::min<uintptr_t>(end, seg_lo(s) + kSeg) - seg_lo(s)); }

    // the locking worker: page-locks segment reg_hi while it lies less than kAhead segments ahead of
    // `consumed` and its pages are mapped
    void run() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        while (!stop) {
            // (never more than 2 x kAhead segments locked, however far the unlocker lags behind)
            if (!failed && !draining && reg_hi < n_seg && reg_hi < consumed + kAhead && reg_hi < reg_lo + 2 * kAhead) {
                if (reg_hi < consumed) {      // the reader skipped ahead: nothing in between is wanted
                    if (reg_lo == reg_hi)     // (once the unlocker has let go of what was locked below)
                        reg_lo = reg_hi = consumed;
                    else
                        cv.wait(lk);
                    continue;
                }
                const size_t sgm = reg_hi;
                if (!populated[sgm]) {        // (a populator has it, or will take it next)
                    cv.wait(lk);
                    continue;
                }
                lk.unlock();
                const double t0 = now_s();
                const hipError_t rc = hipHostRegister(reinterpret_cast<void*>(seg_lo(sgm)), seg_len(sgm),
                                                      hipHostRegisterDefault);
                if (rc != hipSuccess) (void)hipGetLastError();
                lk.lock();
                t_register += now_s() - t0;
                if (rc == hipSuccess)
                    ++reg_hi;
                else
                    failed = true;            // (locked-memory limit, exotic mapping): pageable copies from here on
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
    }

    // the unlocking worker, a thread of its own: hipHostUnregister costs three times what
    // hipHostRegister costs on mapped pages (measured: 47 against 15 ms per 2.9 GB), and on ONE
    // thread the two together filled the whole run -- the caller waited for locks that were queued
    // behind unlocks of segments nobody needed any more
    void unlock_run() {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        while (!stop) {
            if (reg_lo < std::min(consumed, reg_hi)) {
                const size_t sgm = reg_lo;
                lk.unlock();
                const double t0 = now_s();
                (void)hipHostUnregister(reinterpret_cast<void*>(seg_lo(sgm)));
                lk.lock();
                t_unregister += now_s() - t0;
                ++reg_lo;
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
    }

    void open(const void* p, size_t bytes, int dev, int n_populators = kPopulators, size_t seg_bytes = 0) {
        close();
        const uintptr_t page = 4096;
        kSeg = seg_bytes ? seg_bytes : kSegDefault;
        kAhead = std::max<size_t>(2, kAheadBytes / kSeg);
        base = reinterpret_cast<uintptr_t>(p) & ~(page - 1);
        end = (reinterpret_cast<uintptr_t>(p) + bytes + page - 1) & ~(page - 1);
        n_seg = size_t((end - base + kSeg - 1) / kSeg);
        reg_lo = reg_hi = consumed = pop_next = 0;
        t_populate = t_register = t_unregister = t_acquire = 0;
        n_acquire_waits = n_pageable = 0;
        populated.assign(n_seg, 0);
        stop = failed = draining = false;
        device = dev;
        populators.clear();
        try {
            worker = std::thread([this] { run(); });
            unlocker = std::thread([this] { unlock_run(); });
            for (int i = 0; i < std::max(1, n_populators); ++i) populators.emplace_back([this] { populate_run(); });
        } catch (...) {        // a thread could not be started: stop the ones that were, no window
            close();
            throw;
        }
    }

    void close() {
        if (!worker.joinable() && !unlocker.joinable() && populators.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (worker.joinable()) worker.join();
        if (unlocker.joinable()) unlocker.join();
        for (auto& t : populators) t.join();
        populators.clear();
        for (size_t sgm = reg_lo; sgm < reg_hi; ++sgm)      // what is still locked
            (void)hipHostUnregister(reinterpret_cast<void*>(seg_lo(sgm)));
        reg_lo = reg_hi = 0;
        base = end = 0;
        n_seg = 0;
    }

    // The reader is done with the window: nothing more is locked, and the unlocking worker lets go of
    // everything that still is -- in the background; close() (or the next open()) waits for it.
    void release_all() {
        if (!worker.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            draining = true;
            consumed = n_seg;
        }
        cv.notify_all();
    }

    // [src, src + bytes) is about to be copied: wait until its segments are locked.  False: copy
    // it as pageable memory (outside the window, behind it, too far ahead, or locking failed).
    bool acquire(const void* src, size_t bytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(src);
        if (base == 0 || bytes == 0 || a < base || a + bytes > end) return false;
        const size_t s0 = size_t((a - base) / kSeg), s1 = size_t((a + bytes - 1 - base) / kSeg);
        std::unique_lock<std::mutex> lk(mu);
        if (failed || draining || s0 < reg_lo || s1 >= consumed + kAhead) {
            ++n_pageable;
            return false;
        }
        if (!(failed || reg_hi > s1)) {
            const double t0 = now_s();
            cv.wait(lk, [&] { return failed || reg_hi > s1; });
            t_acquire += now_s() - t0;
            ++n_acquire_waits;
        }
        return !failed && s0 >= reg_lo;
    }

    // every copy that ends at or before `upto` has completed
    void release_below(uintptr_t upto) {
        if (base == 0 || upto <= base) return;
        const size_t sgm = size_t((std::min(upto, end) - base) / kSeg);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (sgm <= consumed) return;
            consumed = sgm;
        }
        cv.notify_all();
    }
};

struct thr_handle {
    thr_settings cfg{};
    thr::DevCfg dev{};
    int device = 0;
    int n_cu = 0;
    bool fast = false;       // LDS-resident 16384 kernels; else the generic multi-pass path
    bool lng = false;        // block_len = 2 or 4 x 16384: R0 LDS sub-transforms per block
    bool small = false;      // block_len = 1024 ... 8192: 16 / R1 blocks per workgroup in LDS
    int long_batch = 0;      // long path: blocks per internal sub-batch
    int long_chunk = 0;      // long path: work-list slots per correlate-stage chunk (sizes d_dsub)
    float* d_win_pow = nullptr;     // long: [long_batch][win_w] |X|^2 of the window bins (+-3)
    float* d_partial = nullptr;     // long: [long_batch][R0][2] partial sums of FFT#1
    float2* d_dsub = nullptr;       // long: [long_chunk][T][R0][16384] sub-transform outputs
    bool seg = false;               // long: correlate stage as overlap-save sections (detect_seg.hip)
    float4* d_tspec16k = nullptr;   // sectioned: templates zero-padded to 16384, k_correlate's layout
    thr::CorrStats* d_seg_stats = nullptr;   // sectioned: [long_batch][T][n_seg]
    bool sec4k = false;             // block_len 16384, short template: correlate stage as 4096-sample sections (detect16k_sec.hip)
    float4* d_tspec4k = nullptr;    // sec4k: the templates zero-padded to 4096, the short-block kernels' layout
    float4* d_ctab_pair = nullptr;  // sec4k, several templates: C[32][32] as [j][c] = (C[2j][c], C[2j+1][c])
    float4* d_park = nullptr;       // sec4k, several templates: spectrum scratch (thr::park_bytes_4k; may stay null)
    int path = 0;                   // THR_PATH_* the handle was created with
    int why_unsectioned = 0;        // THR_WHY_* (thr_get_path_info)
    int gen_batch = 0;       // generic path: blocks per internal sub-batch
    float2* d_gen_scratch = nullptr;  // generic path: 3 * gen_batch * N complex
    float2* d_tspec_nat = nullptr;    // generic path: conj(FFT(template))/N, natural order
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // constants
    float2* d_tables = nullptr;
    float2* d_twn = nullptr;
    float4* d_tspec = nullptr;
    // per-batch work buffers
    thr::CarStats* d_stats = nullptr;
    thr::ShiftParams* d_shifts = nullptr;
    thr::CorrStats* d_corr_stats = nullptr;
    int* d_work_list = nullptr;
    int* d_work_count = nullptr;
    float4* d_xhat_scratch = nullptr;   // long (unsectioned), several templates: one spectrum per workgroup
    int* d_ncompact = nullptr;
    int* d_compact_tiles = nullptr;   // per-tile counts / offsets of thr_compact_device (lazy)
    int compact_tiles_cap = 0;
    // PreshiftDetector variant (thr_create_preshift): bank of pre-shifted template spectra
    int preshift_num = 0;       // 0 = default detector
    float2* d_gtw = nullptr;    // combined twiddle table W_16384^(k1 q), L2-resident
    float2* d_bank = nullptr;   // [num][N]; 16384: [k3][k1][k2] gather layout, else natural order
    // host-buffer entry points (thr_detect / _stream / _card): two sets of staging buffers so
    // that the H2D copy of chunk i + 1 (copy stream) runs under the kernels of chunk i (lazy)
    static constexpr int kPipeDepth = THR_MAX_IN_FLIGHT;
    struct HostPipe {
        bool ready = false;
        hipStream_t copy = nullptr;
        hipEvent_t ev_h2d[kPipeDepth] = {};    // chunk's inputs have landed (copy stream)
        hipEvent_t ev_done[kPipeDepth] = {};   // chunk's records are in h_rec (main stream)
        void* d_in[kPipeDepth] = {};
        size_t in_bytes[kPipeDepth] = {};
        long long* d_idx[kPipeDepth] = {};
        thr_record* d_rec[kPipeDepth] = {};
        thr_record* h_rec[kPipeDepth] = {};    // pinned: D2H never blocks the host
        unsigned char* d_text[kPipeDepth] = {};
        size_t text_bytes[kPipeDepth] = {};
        int* d_bad[kPipeDepth] = {};
        int* h_bad = nullptr;                  // pinned int[kPipeDepth]
        // a chunk's block indices and (.card) payload offsets, packed [idx[nb] | off[nb]]: pinned on the
        // host, ONE asynchronous copy into d_idx[b] (2 * max_batch entries; the offsets follow the indices)
        long long* h_meta[kPipeDepth] = {};
        // records of the chunk in buffer b still to be handed to the caller
        thr_record* pend_dst[kPipeDepth] = {};
        size_t pend_n[kPipeDepth] = {};
        size_t pend_first[kPipeDepth] = {};    // (first block of the chunk: error messages)
        bool pend_card[kPipeDepth] = {};
        // thr_submit*() / thr_collect(): the ticket a buffer's pending chunk belongs to (0: none /
        // a chunk of the synchronous entry points), tickets handed out so far, tickets not yet
        // collected
        uint64_t slot_ticket[kPipeDepth] = {};
        uint64_t next_ticket = 1;
        int async_open = 0;
        // input window: the chunk's source range [win_lo, win_end) (win_end 0: not windowed).  Chunks of a
        // raw stream OVERLAP by the history: what may be unlocked behind a finished chunk ends where the
        // earliest chunk still open begins, not where the finished one ended.
        uintptr_t win_lo[kPipeDepth] = {};
        uintptr_t win_end[kPipeDepth] = {};
    } hp;
    InputWindow win;
    // seconds the calling thread spent per phase of the host entry points' chunks
    // (thr_debug_pipe_times): grow staging, H2D calls, metadata, launches, D2H calls, chunks
    // thr_detect_offsets: the caller's sub-bin carrier offsets for the batch in flight (device
    // array; nullptr = the Dirichlet fit), and the staging behind it
    const double* forced = nullptr;
    double* d_forced = nullptr;
    bool sleepy_waits = false;   // thr_set_wait_mode: wait for a batch by query + short sleeps, not by polling
    double t_pipe[8] = {};
    double t_pipe_max[8] = {};   // the longest single occurrence of each phase

    // single-chunk staging of the test hooks (lazy)
    void* d_in = nullptr;
    size_t d_in_bytes = 0;
    long long* d_idx = nullptr;
    thr_record* d_rec = nullptr;
    // profiling
    int prof_every = 0;      // 0 = off, n = bracket the kernels of every n-th batch
    long long batch_no = 0;
    bool prof = false;       // this batch is being timed
    std::vector<EventPair> free_events;
    std::vector<EventPair> pending[THR_N_KERNEL_SLOTS];
    double ms[THR_N_KERNEL_SLOTS] = {};
    int64_t launches[THR_N_KERNEL_SLOTS] = {};
};

namespace {

struct ProfScope {
    thr_handle* h;
    int slot;
    EventPair ev{};
    bool on;
    hipStream_t stream;
    ProfScope(thr_handle* h_, int slot_, hipStream_t stream_ = nullptr)
        : h(h_), slot(slot_), on(h_->prof), stream(stream_ ? stream_ : h_->stream) {
        if (on) {
            if (!h->free_events.empty()) {
                ev = h->free_events.back();
                h->free_events.pop_back();
            } else {
                (void)hipEventCreate(&ev.a);
                (void)hipEventCreate(&ev.b);
            }
            (void)hipEventRecord(ev.a, stream);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(ev.b, stream);
            h->pending[slot].push_back(ev);
        }
    }
};

int window_indices(int start, int stop, int n, int* lo, int* count) {
    // carrier_detect.py:17-58
    if (std::abs(start) >= n || std::abs(stop) >= n)
        return fail(THR_ERR_ARG, "Frequency window out of range: %d - %d", start, stop);
    if (start < 0 && stop >= 0) {
        start += n;
        stop += n;
    }
    if (start < 0) start += n;
    if (stop < 0) stop += n;
    if (stop < start) std::swap(start, stop);
    *lo = start;
    *count = std::min(stop - start + 1, n);
    return THR_OK;
}

// Overlap-save sections of a long block's correlate stage (detect_seg.hip).  A section is 16384
// samples; against a W-sample template its lags 0 .. V - 1, V = 16384 - W + 1, are exact lags of
// the block (soa_estimator.py:97-102 keeps only lags that do not wrap).  Sections start every D
// samples, D = the largest even number <= V - 2 (even: u8 samples are fetched as 4-byte pairs; - 2:
// a section must also hold the lag below and the lag above every lag it owns, for the peak's
// neighbours, soa_estimator.py:159-170), the last one at block_len - 16384.  Section g > 0 owns the
// block's lags from its start + 1 up to the next section's start; section 0 owns lag 0 too, the
// last one everything up to corr_len.  Returns false (d.n_seg = 0) when the block needs more than
// kMaxSections -- templates longer than about half a section: the decimated kernels keep those.
bool plan_sections(thr::DevCfg& d, int template_len) {
    const int m = 16384, n = d.block_len;
    d.n_seg = 0;
    const int v = m - template_len + 1;
    if (n <= m || v < 4) return false;
    const int stride = (v - 2) & ~1;
    const int n_seg = (n - m + stride - 1) / stride + 1;
    if (n_seg > thr::kMaxSections) return false;
    int own_lo = 0;   // first lag of the block section g owns
    for (int g = 0; g < n_seg; ++g) {
        const int start = std::min(g * stride, n - m);
        const int own_hi = g + 1 < n_seg ? std::min((g + 1) * stride, n - m) + 1 : d.corr_len;
        d.seg_start[g] = start;
        d.seg_sum_lo[g] = own_lo - start;
        d.seg_sum_hi[g] = own_hi - start;
        d.seg_lo[g] = std::max(own_lo, d.corr_lo) - start;
        d.seg_hi[g] = std::max(std::min(own_hi, d.corr_hi), std::max(own_lo, d.corr_lo)) - start;
        own_lo = own_hi;
    }
    d.n_seg = n_seg;
    return true;
}

// The same idea one size down (detect16k_sec.hip): a 16384-sample block whose template is short
// enough that FOUR 4096-sample sections or fewer cover its unique window [corr_lo, corr_hi) -- then
// the sections' transforms cost less than the block's (4 x 4096 x 12 < 16384 x 14 butterfly
// stages).  Only the window is covered (no stddev term on this path: its sums run over every kept
// lag).  A section holds V = 4096 - W + 1 exact lags; it owns at most V - 2 of them (the lag below
// and the lag above every owned lag must be in it too, for the peak's neighbours,
// soa_estimator.py:159-170 -- except at the two ends of the kept lags, where the reference takes no
// neighbours either).  Sections start on multiples of 8 samples (u8 samples are fetched 16 bytes
// per thread), every D = (V - 2) & ~7 samples from the last multiple of 8 at or below corr_lo - 1,
// the last one no later than block_len - 4096.  BASELINE (history 4096, 1023 samples): starts
// 1536 + 3072 g, every section owns its lags [1, 3073).
bool plan_sections_4k(thr::DevCfg& d, int template_len) {
    const int m = 4096, n = d.block_len;
    d.n_seg = 0;
    const int v = m - template_len + 1;
    if (n != 16384 || v < 16) return false;
    const int stride = (v - 2) & ~7;
    const int s0 = std::max(d.corr_lo - 1, 0) & ~7;
    int own_lo = d.corr_lo, g = 0;
    while (own_lo < d.corr_hi) {
        if (g == 4) return false;      // a fifth section: the 16384-point kernel is cheaper
        const int start = std::min(s0 + g * stride, n - m);
        // lags [start, start + v) are in the section; it owns from the previous section's end up to its
        // last lag but one -- or up to its last lag, where that is the last kept lag of the block
        const int cap = start + v == d.corr_len ? d.corr_len : start + v - 1;
        const int own_hi = std::min(cap, d.corr_hi);
        if (own_hi <= own_lo || (own_lo > 0 && own_lo - 1 < start)) return false;
        d.seg_start[g] = start;
        d.seg_lo[g] = own_lo - start;
        d.seg_hi[g] = own_hi - start;
        d.seg_sum_lo[g] = d.seg_sum_hi[g] = 0;
        own_lo = own_hi;
        ++g;
    }
    d.n_seg = g;
    return g > 0;
}

int build_constants(thr_handle* h) {
    const int n = h->cfg.block_len;
    // --- LDS twiddle tables (forward sign): C[32][32], A[16][32], Bt[16][32]  (fast path)
    std::vector<float2> tab(2048);
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b) tab[a * 32 + b] = unit_root((long long)a * b, 1024);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 32; ++n2) tab[1024 + k1 * 32 + n2] = unit_root((long long)k1 * n2, 512);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int mp = 0; mp < 32; ++mp)
            tab[1536 + k1 * 32 + mp] = unit_root((long long)k1 * mp, h->lng ? 16384 : n);
    HIP_TRY(hipMalloc(&h->d_tables, tab.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(h->d_tables, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    // --- pass-1 / pass-B twiddles W_16384^(k1 q) of k_correlate (one template) and of the
    //     short-block kernels as one L2-resident table in global memory
    h->dev.gtw = nullptr;
    if (h->small || h->fast || h->seg || h->sec4k) {
        std::vector<float2> g(16 * 1024);
        for (int k1 = 0; k1 < 16; ++k1)
            for (int q = 0; q < 1024; ++q) g[k1 * 1024 + q] = unit_root((long long)k1 * q, 16384);
        HIP_TRY(hipMalloc(&h->d_gtw, g.size() * sizeof(float2)));
        HIP_TRY(hipMemcpy(h->d_gtw, g.data(), g.size() * sizeof(float2), hipMemcpyHostToDevice));
        h->dev.gtw = h->d_gtw;
    }
    // --- full-length root table for the shift phasor
    std::vector<float2> tw(n);
    for (int j = 0; j < n; ++j) tw[j] = unit_root(j, n);
    HIP_TRY(hipMalloc(&h->d_twn, tw.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(h->d_twn, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    // --- template spectra: conj(FFT(zero-padded template)) / N, in the
    //     digit-reversed, lane-coalesced order k_correlate consumes
    const int w = h->cfg.template_len, nt = h->cfg.n_templates;
    std::vector<float2> spec(size_t(nt) * n);
    for (int t = 0; t < nt; ++t) {
        std::vector<std::complex<double>> buf(n, 0.0);
        double energy = 0;
        for (int i = 0; i < w; ++i) {
            const double v = h->cfg.templates[size_t(t) * w + i];
            buf[i] = v;
            energy += v * v;
        }
        h->dev.tmpl_energy[t] = float(energy);
        host_fft(buf);
        float2* out = spec.data() + size_t(t) * n;
        if (h->lng) {
            // sub-transform k0 holds bins k0 + R0*q; within it the 16384 kernels' permutation
            const int r0 = n / 16384;
            for (int k0 = 0; k0 < r0; ++k0)
                for (int tid = 0; tid < 512; ++tid)
                    for (int k3 = 0; k3 < 32; ++k3) {
                        const int q = (tid >> 5) + 16 * (tid & 31) + 512 * k3;
                        const std::complex<double> c = std::conj(buf[k0 + r0 * q]) / double(n);
                        out[size_t(k0) * 16384 + ((k3 >> 1) * 512 + tid) * 2 + (k3 & 1)] =
                            float2{float(c.real()), float(c.imag())};
                    }
        } else if (h->small) {
            // thread column c = k1 * 32 + k2 (k1 < R1) holds bins k1 + R1 k2 + 32 R1 k3;
            // float4 j of the column = k3 in {2j, 2j + 1}, stored [j][c] for coalescing
            const int r1 = n / 1024, tb = 32 * r1;
            for (int c = 0; c < tb; ++c)
                for (int k3 = 0; k3 < 32; ++k3) {
                    const int k = (c >> 5) + r1 * (c & 31) + tb * k3;
                    const std::complex<double> cc = std::conj(buf[k]) / double(n);
                    out[((k3 >> 1) * tb + c) * 2 + (k3 & 1)] = float2{float(cc.real()), float(cc.imag())};
                }
        } else if (h->fast) {
            for (int tid = 0; tid < 512; ++tid)
                for (int k3 = 0; k3 < 32; ++k3) {
                    const int k = (tid >> 5) + 16 * (tid & 31) + 512 * k3;
                    const std::complex<double> c = std::conj(buf[k]) / double(n);
                    out[((k3 >> 1) * 512 + tid) * 2 + (k3 & 1)] =
                        float2{float(c.real()), float(c.imag())};
                }
        } else {
            for (int k = 0; k < n; ++k) {
                const std::complex<double> c = std::conj(buf[k]) / double(n);
                out[k] = float2{float(c.real()), float(c.imag())};
            }
        }
    }
    if (h->seg) {
        // sectioned correlate stage: conj(FFT(template zero-padded to 16384)) / 16384 in the
        // digit-reversed, lane-coalesced order k_correlate consumes (same as block_len 16384)
        const int m = 16384;
        std::vector<float2> s16(size_t(nt) * m);
        for (int t = 0; t < nt; ++t) {
            std::vector<std::complex<double>> buf(m, 0.0);
            for (int i = 0; i < w; ++i) buf[i] = h->cfg.templates[size_t(t) * w + i];
            host_fft(buf);
            float2* out = s16.data() + size_t(t) * m;
            for (int tid = 0; tid < 512; ++tid)
                for (int k3 = 0; k3 < 32; ++k3) {
                    const int k = (tid >> 5) + 16 * (tid & 31) + 512 * k3;
                    const std::complex<double> c = std::conj(buf[k]) / double(m);
                    out[((k3 >> 1) * 512 + tid) * 2 + (k3 & 1)] = float2{float(c.real()), float(c.imag())};
                }
        }
        HIP_TRY(hipMalloc(&h->d_tspec16k, s16.size() * sizeof(float2)));
        HIP_TRY(hipMemcpy(h->d_tspec16k, s16.data(), s16.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (h->sec4k) {
        // 4096-sample sections of a 16384-sample block: conj(FFT(template zero-padded to 4096)) / 4096,
        // thread column c = row * 32 + k2 holds bins row + 4 k2 + 128 k3 (the short-block layout, R1 = 4)
        const int m = 4096, r1 = 4, tb = 128;
        std::vector<float2> s4(size_t(nt) * m);
        for (int t = 0; t < nt; ++t) {
            std::vector<std::complex<double>> buf(m, 0.0);
            for (int i = 0; i < w; ++i) buf[i] = h->cfg.templates[size_t(t) * w + i];
            host_fft(buf);
            float2* out = s4.data() + size_t(t) * m;
            for (int c = 0; c < tb; ++c)
                for (int k3 = 0; k3 < 32; ++k3) {
                    const int k = (c >> 5) + r1 * (c & 31) + tb * k3;
                    const std::complex<double> cc = std::conj(buf[k]) / double(m);
                    out[((k3 >> 1) * tb + c) * 2 + (k3 & 1)] = float2{float(cc.real()), float(cc.imag())};
                }
        }
        HIP_TRY(hipMalloc(&h->d_tspec4k, s4.size() * sizeof(float2)));
        HIP_TRY(hipMemcpy(h->d_tspec4k, s4.data(), s4.size() * sizeof(float2), hipMemcpyHostToDevice));
        if (nt > 1) {
            // the C table in pairs, for the kernel form that re-reads its twiddle column every pass
            std::vector<float2> cp(1024);
            for (int j = 0; j < 16; ++j)
                for (int c = 0; c < 32; ++c) {
                    cp[(j * 32 + c) * 2] = tab[(2 * j) * 32 + c];
                    cp[(j * 32 + c) * 2 + 1] = tab[(2 * j + 1) * 32 + c];
                }
            HIP_TRY(hipMalloc(&h->d_ctab_pair, cp.size() * sizeof(float2)));
            HIP_TRY(hipMemcpy(h->d_ctab_pair, cp.data(), cp.size() * sizeof(float2), hipMemcpyHostToDevice));
            const size_t pb = thr::park_bytes_4k(4 * h->n_cu);
            if (pb) HIP_TRY(hipMalloc(&h->d_park, pb));
        }
    }
    float2* d_spec = nullptr;
    HIP_TRY(hipMalloc(&d_spec, spec.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(d_spec, spec.data(), spec.size() * sizeof(float2), hipMemcpyHostToDevice));
    if (h->fast || h->lng || h->small)
        h->d_tspec = reinterpret_cast<float4*>(d_spec);
    else
        h->d_tspec_nat = d_spec;
    return THR_OK;
}

// Pre-shifted template spectra of the PreshiftDetector variant (detect_preshift.py:24-40):
// conj(FFT(template_padded * exp(-2 pi i shift_j (n/N - 1/2)))) / N, shift_j = linspace(-.5, .5, num).
int build_preshift_bank(thr_handle* h) {
    const int n = h->cfg.block_len, w = h->cfg.template_len, num = h->preshift_num;
    std::vector<float2> bank(size_t(num) * n);
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < num; ++j) {
        // (fastdet-compatible variant: ONE unshifted template spectrum)
        const double shift = h->dev.variant == 2 ? 0.0 : num > 1 ? -0.5 + double(j) / double(num - 1) : -0.5;
        std::vector<std::complex<double>> buf(n, 0.0);
        for (int i = 0; i < w; ++i) {
            const double ph = -2.0 * pi * shift * (double(i) / double(n) - 0.5);
            buf[i] = h->cfg.templates[i] * std::complex<double>(std::cos(ph), std::sin(ph));
        }
        host_fft(buf);
        float2* out = bank.data() + size_t(j) * n;
        for (int k = 0; k < n; ++k) {
            const std::complex<double> c = std::conj(buf[k]) / double(n);
            // 16384 kernel: bin k = k1 + 16 k2 + 512 k3 lives at (k3 * 16 + k1) * 32 + k2
            const int pos = h->fast ? (((k >> 9) * 16 + (k & 15)) * 32 + ((k >> 4) & 31)) : k;
            out[pos] = float2{float(c.real()), float(c.imag())};
        }
    }
    HIP_TRY(hipMalloc(&h->d_bank, bank.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(h->d_bank, bank.data(), bank.size() * sizeof(float2), hipMemcpyHostToDevice));
    return THR_OK;
}

int ensure_pipe(thr_handle* h) {
    auto& p = h->hp;
    if (p.ready) return THR_OK;
    const size_t mb = size_t(h->cfg.max_batch), nt = size_t(h->cfg.n_templates);
    HIP_TRY(hipStreamCreateWithFlags(&p.copy, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p.h_bad), thr_handle::kPipeDepth * sizeof(int),
                          hipHostMallocDefault));
    for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
        HIP_TRY(hipEventCreateWithFlags(&p.ev_h2d[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p.ev_done[b], hipEventDisableTiming));
        HIP_TRY(hipMalloc(&p.d_idx[b], 2 * mb * sizeof(long long)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p.h_meta[b]), 2 * mb * sizeof(long long),
                              hipHostMallocDefault));
        HIP_TRY(hipMalloc(&p.d_rec[b], mb * nt * sizeof(thr_record)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p.h_rec[b]), mb * nt * sizeof(thr_record),
                              hipHostMallocDefault));
    }
    p.ready = true;
    return THR_OK;
}

// chunk input: caller memory -> device on the copy stream.  The source is pageable, so the call
// returns once the HIP runtime has staged it (~44 GB/s, whatever the chunk size); the DMA and the
// kernels of the previous chunk run meanwhile.  (Measured and not adopted: our own pinned
// staging filled by 3-6 host threads with one chunk of look-ahead -- 54 GB/s in some runs,
// 29-33 GB/s in others on the same box, and never ahead below 48 MiB per chunk.)
// With an input window (thr_input_window) around `src` the range is page-locked by then: one
// asynchronous copy per locked segment (a copy never straddles two registrations), the call
// returns at once and buffer b remembers how far the window has been read.
int pipe_h2d(thr_handle* h, int b, void* d_dst, const void* src, size_t bytes) {
    h->hp.win_end[b] = 0;
    if (h->win.acquire(src, bytes)) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(src);
        size_t done = 0;

        while (done < bytes) {
            const uintptr_t at = a + done;
            const uintptr_t seg_end = h->win.base + (size_t((at - h->win.base) / h->win.kSeg) + 1) * h->win.kSeg;
            const size_t n = std::min<size_t>(bytes - done, size_t(seg_end - at));
            HIP_TRY(hipMemcpyAsync(static_cast<char*>(d_dst) + done, reinterpret_cast<const void*>(at), n,
                                   hipMemcpyHostToDevice, h->hp.copy));
            done += n;
        }
        h->hp.win_lo[b] = a;
        h->hp.win_end[b] = a + bytes;
        return THR_OK;
    }
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->hp.copy));
    return THR_OK;
}

// buffer b's chunk has left host memory (its H2D event or its done event has been waited for)
void pipe_inputs_done(thr_handle* h, int b) {
    auto& p = h->hp;
    if (!p.win_end[b]) return;
    uintptr_t upto = p.win_end[b];
    p.win_end[b] = p.win_lo[b] = 0;
    // nothing an open chunk still reads may be unlocked: a later chunk of a raw stream starts
    // 2 * history bytes BEFORE the end of this one, possibly in the segment below
    for (int o = 0; o < thr_handle::kPipeDepth; ++o)
        if (p.win_end[o]) upto = std::min(upto, p.win_lo[o]);
    h->win.release_below(upto);
}

// blocks per chunk of the host entry points: the staging buffers stay near 64 MiB each
size_t pipe_chunk_blocks(const thr_handle* h, size_t bytes_per_block) {
    const size_t cap = std::max<size_t>(1, (size_t(64) << 20) / std::max<size_t>(1, bytes_per_block));
    return std::min(size_t(h->cfg.max_batch), cap);
}

int pipe_grow(void** buf, size_t* have, size_t need) {
    if (*have >= need) return THR_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(buf, need + (need >> 3)));
    *have = need + (need >> 3);
    return THR_OK;
}

// Wait for an event.  hipEventSynchronize polls (the calling thread stays busy for the length of the
// wait, whatever flags the event was created with -- measured); on a host whose CPUs are shared by
// several ranks that is a CPU per rank taken from the ranks' text and page-locking threads.  The
// sleeping form asks and naps: three batches are in flight, so a nap of 40 us in front of a 1.3 ms
// wait costs the pipeline nothing.
int wait_event(thr_handle* h, hipEvent_t ev) {
    if (!h->sleepy_waits) {
        HIP_TRY(hipEventSynchronize(ev));
        return THR_OK;
    }
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return THR_OK;
        if (e != hipErrorNotReady)
            return fail(THR_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();          // (hipErrorNotReady is sticky in hipGetLastError)
        std::this_thread::sleep_for(std::chrono::microseconds(40));
    }
}

// hand the finished chunk of buffer b to the caller (waits for it); THR_OK if nothing is pending
int pipe_drain(thr_handle* h, int b) {
    auto& p = h->hp;
    if (p.pend_n[b] == 0) return THR_OK;
    {
        const int wrc = wait_event(h, p.ev_done[b]);
        if (wrc != THR_OK) return wrc;
    }
    pipe_inputs_done(h, b);
    std::memcpy(p.pend_dst[b], p.h_rec[b], p.pend_n[b] * sizeof(thr_record));
    const size_t n = p.pend_n[b], first = p.pend_first[b];
    p.pend_n[b] = 0;
    if (p.pend_card[b] && p.h_bad[b] != 0)
        return fail(THR_ERR_ARG, "%d .card payload(s) in blocks [%zu, %zu) are not valid base64",
                    p.h_bad[b], first, first + n / size_t(h->cfg.n_templates));
    return THR_OK;
}

// after the chunk's H2D copies were enqueued on the copy stream: make the main stream wait for them
int pipe_inputs_enqueued(thr_handle* h, int b) {
    auto& p = h->hp;
    HIP_TRY(hipEventRecord(p.ev_h2d[b], p.copy));
    HIP_TRY(hipStreamWaitEvent(h->stream, p.ev_h2d[b], 0));
    return THR_OK;
}

// after the chunk's kernels were enqueued on the main stream: records -> pinned staging, async
int pipe_records_enqueued(thr_handle* h, int b, thr_record* dst, size_t n_rec, size_t first, bool card) {
    auto& p = h->hp;
    HIP_TRY(hipMemcpyAsync(p.h_rec[b], p.d_rec[b], n_rec * sizeof(thr_record), hipMemcpyDeviceToHost,
                           h->stream));
    if (card)
        HIP_TRY(hipMemcpyAsync(p.h_bad + b, p.d_bad[b], sizeof(int), hipMemcpyDeviceToHost, h->stream));

    HIP_TRY(hipEventRecord(p.ev_done[b], h->stream));
    p.pend_dst[b] = dst;
    p.pend_n[b] = n_rec;
    p.pend_first[b] = first;
    p.pend_card[b] = card;
    return THR_OK;
}

int pipe_finish(thr_handle* h, int rc) {   // drain every buffer; keeps the first error
    for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
        const int r = pipe_drain(h, b);
        if (rc == THR_OK) rc = r;
    }
    if (rc != THR_OK) {
        (void)hipStreamSynchronize(h->hp.copy);
        (void)hipStreamSynchronize(h->stream);
        for (int b = 0; b < thr_handle::kPipeDepth; ++b) h->hp.pend_n[b] = 0;
    }
    return rc;
}

int ensure_staging(thr_handle* h, int format) {
    const size_t need = size_t(h->cfg.max_batch) * h->cfg.block_len * (format == THR_IN_U8 ? 2 : 8);
    if (h->d_in_bytes < need) {
        if (h->d_in) (void)hipFree(h->d_in);
        h->d_in = nullptr;
        h->d_in_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_in, need));
        h->d_in_bytes = need;
    }
    if (!h->d_idx) HIP_TRY(hipMalloc(&h->d_idx, size_t(h->cfg.max_batch) * sizeof(long long)));
    if (!h->d_rec)
        HIP_TRY(hipMalloc(&h->d_rec,
                          size_t(h->cfg.max_batch) * h->cfg.n_templates * sizeof(thr_record)));
    return THR_OK;
}

int run_batch_fast(thr_handle* h, const void* d_samples_all, int format,
                   const long long* d_block_idx_all, int n_blocks_all, thr_record* d_out_all,
                   float2* dump_fft, float2* dump_xhat, float2* dump_corr, int dump_template,
                   bool carrier_only) {
    h->prof = h->prof_every > 0 && (h->batch_no++ % h->prof_every) == 0;
    if (h->preshift_num) {
        const int grid = std::min(n_blocks_all, h->n_cu);
        if (dump_fft || dump_xhat || dump_corr || carrier_only)
            return fail(THR_ERR_ARG, "the fused preshift kernel has no stage dumps: create the handle with "
                                     "THR_PATH_MULTIPASS (thr_create_ex) for them");
        {
            ProfScope p(h, 2);   // the fused kernel is accounted in k_correlate's slot
            HIP_TRY(thr::launch_preshift_16k(format, d_samples_all, n_blocks_all, h->dev, h->d_tables,
                                             h->d_bank, h->preshift_num, d_block_idx_all,
                                             h->d_corr_stats, d_out_all, grid, h->stream));
        }
        ProfScope p(h, 3);
        HIP_TRY(thr::launch_finish(n_blocks_all, h->dev, h->d_corr_stats, d_out_all, h->d_work_count,
                                   h->stream));
        return THR_OK;
    }
    // (Measured and not adopted: internal chunks of 4096 / 8192 blocks, so that k_correlate's read
    // of the samples finds them in the 256 MiB Infinity Cache after the carrier kernel -- 11.5 /
    // 12.1 M blocks/s against 12.4 M for the whole batch (one grid ramp per extra launch), and
    // FETCH_SIZE does not move: it counts L2-to-fabric requests, Infinity-Cache hits included.
    // profiles/README.md.)
    const int chunk = n_blocks_all;
    const size_t T = size_t(h->cfg.n_templates);
    for (int off = 0; off < n_blocks_all; off += chunk) {
        const int n_blocks = std::min(chunk, n_blocks_all - off);
        const void* d_samples =
            static_cast<const unsigned char*>(d_samples_all) + size_t(off) * size_t(h->dev.blk_stride);
        const long long* d_block_idx = d_block_idx_all ? d_block_idx_all + off : nullptr;
        thr_record* d_out = d_out_all + size_t(off) * T;
        const int grid = std::min(n_blocks, h->n_cu);
        {
            ProfScope p(h, 0);
            HIP_TRY(thr::launch_carrier_16k(format, d_samples, n_blocks, h->dev, h->d_tables, h->d_twn,
                                            h->d_stats, dump_fft, grid, h->stream));
        }
        if (carrier_only) continue;
        {
            ProfScope p(h, 1);
            HIP_TRY(thr::launch_fit(n_blocks, h->dev, h->d_stats, d_block_idx, h->d_shifts,
                                    h->d_work_list, h->d_work_count, d_out, h->d_corr_stats, h->stream,
                                    h->forced ? h->forced + off : nullptr));
        }
        // short template: 4096-sample sections, four 2-wave workgroups per CU (detect16k_sec.hip);
        // the stage dumps are the unsectioned kernel's
        const bool sec = h->sec4k && !dump_xhat && !dump_corr;
        {
            ProfScope p(h, 2);
            if (sec)
                HIP_TRY(thr::launch_correlate_4k(format, d_samples, h->dev, h->d_tables, h->d_twn, h->d_tspec4k,
                                                 h->d_shifts, h->d_work_list, h->d_work_count, h->d_seg_stats,
                                                 h->d_ctab_pair, h->d_park,
                                                 int(std::min<long long>((long long)n_blocks * h->dev.n_seg,
                                                                         4ll * h->n_cu)),
                                                 h->stream));
            else
                HIP_TRY(thr::launch_correlate_16k(
                    format, d_samples, h->dev, h->d_tables, h->d_twn, h->d_tspec, h->d_shifts, h->d_work_list,
                    h->d_work_count, h->d_corr_stats, dump_xhat, dump_corr, dump_template, grid, h->stream));
        }
        {
            ProfScope p(h, 3);
            HIP_TRY(thr::launch_finish(n_blocks * h->cfg.n_templates, h->dev, h->d_corr_stats, d_out,
                                       h->d_work_count, h->stream, sec ? h->d_seg_stats : nullptr));
        }
    }
    return THR_OK;
}

// Generic block lengths: multi-pass pipeline through HBM, in sub-batches of gen_batch blocks.
// The dump_* pointers (debug only) receive copies of the natural-order intermediates.
int run_batch_generic(thr_handle* h, const void* d_samples, int format,
                      const long long* d_block_idx, int n_blocks, thr_record* d_out,
                      float2* dump_fft, float2* dump_xhat, float2* dump_corr, int dump_template,
                      bool carrier_only) {
    const size_t n = size_t(h->cfg.block_len), T = size_t(h->cfg.n_templates);
    const size_t blk_bytes = size_t(h->dev.blk_stride);
    h->prof = h->prof_every > 0 && (h->batch_no++ % h->prof_every) == 0;
    for (int off = 0; off < n_blocks; off += h->gen_batch) {
        const int nb = std::min(h->gen_batch, n_blocks - off);
        const void* in = static_cast<const unsigned char*>(d_samples) + size_t(off) * blk_bytes;
        thr_record* out = d_out + size_t(off) * T;
        float2* spectrum = nullptr;
        {
            ProfScope p(h, 0);
            HIP_TRY(thr::generic_carrier(format, in, nb, h->dev, h->d_twn, h->d_gen_scratch,
                                         h->d_stats, &spectrum, h->stream));
        }
        if (dump_fft)
            HIP_TRY(hipMemcpyAsync(dump_fft + size_t(off) * n, spectrum, size_t(nb) * n * sizeof(float2),
                                   hipMemcpyDeviceToDevice, h->stream));
        if (carrier_only) continue;
        if (h->preshift_num) {
            {
                ProfScope p(h, 1);
                HIP_TRY(thr::launch_fit_preshift(nb, h->dev, h->preshift_num, h->d_stats,
                                                 d_block_idx ? d_block_idx + off : nullptr,
                                                 h->d_shifts, out, h->stream));
            }
            float2* cc = nullptr;
            {
                ProfScope p(h, 2);
                HIP_TRY(thr::generic_preshift_correlate(nb, h->dev, h->d_twn, h->d_bank, h->d_shifts,
                                                        out, h->d_gen_scratch, spectrum,
                                                        h->d_corr_stats,
                                                        dump_xhat ? dump_xhat + size_t(off) * n : nullptr,
                                                        dump_corr ? &cc : nullptr, h->stream));
            }
            if (dump_corr && cc)
                HIP_TRY(hipMemcpyAsync(dump_corr + size_t(off) * n, cc, size_t(nb) * n * sizeof(float2),
                                       hipMemcpyDeviceToDevice, h->stream));
            ProfScope p(h, 3);
            HIP_TRY(thr::launch_finish(nb, h->dev, h->d_corr_stats, out, h->d_work_count, h->stream));
            continue;
        }
        {
            ProfScope p(h, 1);
            HIP_TRY(thr::launch_fit(nb, h->dev, h->d_stats, d_block_idx ? d_block_idx + off : nullptr,
                                    h->d_shifts, h->d_work_list, h->d_work_count, out, nullptr,
                                    h->stream, h->forced ? h->forced + off : nullptr));
        }
        float2 *xh = nullptr, *cc = nullptr;
        {
            ProfScope p(h, 2);
            HIP_TRY(thr::generic_correlate(format, in, nb, h->dev, h->d_twn, h->d_tspec_nat,
                                           h->d_shifts, out, h->d_gen_scratch, h->d_corr_stats,
                                           dump_template, dump_xhat ? &xh : nullptr,
                                           dump_corr ? &cc : nullptr, h->stream));
        }
        if (dump_xhat && xh)
            HIP_TRY(hipMemcpyAsync(dump_xhat + size_t(off) * n, xh, size_t(nb) * n * sizeof(float2),
                                   hipMemcpyDeviceToDevice, h->stream));
        if (dump_corr && cc)
            HIP_TRY(hipMemcpyAsync(dump_corr + size_t(off) * n, cc, size_t(nb) * n * sizeof(float2),
                                   hipMemcpyDeviceToDevice, h->stream));
        {
            ProfScope p(h, 3);
            HIP_TRY(thr::launch_finish(nb * int(T), h->dev, h->d_corr_stats, out, h->d_work_count,
                                       h->stream));
        }
    }
    return THR_OK;
}

// Long blocks (2 or 4 x 16384): R0 LDS-resident sub-transforms per block, sub-batched.
int run_batch_long(thr_handle* h, const void* d_samples, int format,
                   const long long* d_block_idx, int n_blocks, thr_record* d_out, float2* dump_fft,
                   float2* dump_xhat, float2* dump_corr, int dump_template, bool carrier_only) {
    const size_t n = size_t(h->cfg.block_len), T = size_t(h->cfg.n_templates);
    const size_t blk_bytes = size_t(h->dev.blk_stride);
    const int r0 = int(n / 16384);
    h->prof = h->prof_every > 0 && (h->batch_no++ % h->prof_every) == 0;
    for (int off = 0; off < n_blocks; off += h->long_batch) {
        const int nb = std::min(h->long_batch, n_blocks - off);
        const void* in = static_cast<const unsigned char*>(d_samples) + size_t(off) * blk_bytes;
        thr_record* out = d_out + size_t(off) * T;
        {
            ProfScope p(h, 0);
            HIP_TRY(thr::launch_carrier_long(format, in, nb, h->dev, h->d_tables, h->d_twn,
                                             h->d_win_pow, h->d_partial, h->d_stats,
                                             dump_fft ? dump_fft + size_t(off) * n : nullptr,
                                             std::min(nb * r0, h->n_cu), h->stream));
        }
        if (carrier_only) continue;
        {
            ProfScope p(h, 1);
            HIP_TRY(thr::launch_fit(nb, h->dev, h->d_stats, d_block_idx ? d_block_idx + off : nullptr,
                                    h->d_shifts, h->d_work_list, h->d_work_count, out,
                                    h->d_corr_stats, h->stream,     // (sub-batch-local indices)
                                    h->forced ? h->forced + off : nullptr));
        }
        // correlate stage: one fused launch (sub-transforms + combination per workgroup) does
        // every batch with at least one carrier-positive block per workgroup; smaller ones fall
        // to the two-kernel form, in chunks of work-list slots (one chunk's d_k0 exchange stays
        // in the Infinity Cache between the two kernels).  Each form returns at once when the
        // batch is the other's (the work count lives on the device).
        if (h->seg && !dump_xhat && !dump_corr) {
            // overlap-save sections of the 16384-point kernel: one work item per (block, section)
            {
                ProfScope p(h, 2);
                HIP_TRY(thr::launch_correlate_seg(format, in, h->dev, h->d_tables, h->d_twn, h->d_tspec16k,
                                                  h->d_shifts, h->d_work_list, h->d_work_count,
                                                  h->d_seg_stats, std::min(nb * h->dev.n_seg, h->n_cu),
                                                  h->stream));
            }
            ProfScope p(h, 3);
            HIP_TRY(thr::launch_finish(nb * int(T), h->dev, h->d_corr_stats, out, h->d_work_count,
                                       h->stream, h->d_seg_stats));
            continue;
        }
        const int fused_grid = std::min(nb, h->n_cu);
        float2* dcorr = dump_corr ? dump_corr + size_t(off) * n : nullptr;
        float2* dxhat = dump_xhat ? dump_xhat + size_t(off) * n : nullptr;
        {
            ProfScope p(h, 2);
            HIP_TRY(thr::launch_correlate_long(true, format, in, h->dev, h->d_tables, h->d_twn,
                                               h->d_tspec, h->d_shifts, h->d_work_list,
                                               h->d_work_count, h->d_dsub, h->d_xhat_scratch, dxhat,
                                               h->d_corr_stats, dcorr, dump_template, fused_grid, 0,
                                               nb, fused_grid, h->stream));
        }
        for (int base = 0; base < fused_grid; base += h->long_chunk) {   // (fewer than fused_grid slots)
            const int cap = std::min(h->long_chunk, fused_grid - base);
            ProfScope p(h, 4);
            HIP_TRY(thr::launch_correlate_long(false, format, in, h->dev, h->d_tables, h->d_twn,
                                               h->d_tspec, h->d_shifts, h->d_work_list,
                                               h->d_work_count, h->d_dsub, h->d_xhat_scratch, dxhat,
                                               h->d_corr_stats, dcorr, dump_template,
                                               std::min(nb * r0, h->n_cu), base, cap, fused_grid,
                                               h->stream));
            HIP_TRY(thr::launch_combine_long(h->dev, h->d_twn, h->d_work_list, h->d_work_count,
                                             h->d_dsub, h->d_corr_stats, dcorr, dump_template, base,
                                             cap, fused_grid, h->stream));
        }
        {
            ProfScope p(h, 3);
            HIP_TRY(thr::launch_finish(nb * int(T), h->dev, h->d_corr_stats, out, h->d_work_count,
                                       h->stream));
        }
    }
    return THR_OK;
}

// Short blocks (1024 ... 8192): 16 / R1 blocks per workgroup, LDS-resident (detect_small.hip).
int run_batch_small(thr_handle* h, const void* d_samples, int format,
                    const long long* d_block_idx, int n_blocks, thr_record* d_out, float2* dump_fft,
                    float2* dump_xhat, float2* dump_corr, int dump_template, bool carrier_only) {
    h->prof = h->prof_every > 0 && (h->batch_no++ % h->prof_every) == 0;
    {
        ProfScope p(h, 0);
        HIP_TRY(thr::launch_carrier_small(format, d_samples, n_blocks, h->dev, h->d_tables, h->d_gtw,
                                          h->d_stats, dump_fft, h->n_cu, h->stream));
    }
    if (carrier_only) return THR_OK;
    {
        ProfScope p(h, 1);
        HIP_TRY(thr::launch_fit(n_blocks, h->dev, h->d_stats, d_block_idx, h->d_shifts,
                                h->d_work_list, h->d_work_count, d_out, h->d_corr_stats, h->stream, h->forced));
    }
    {
        ProfScope p(h, 2);
        HIP_TRY(thr::launch_correlate_small(format, d_samples, n_blocks, h->dev, h->d_tables, h->d_gtw,
                                            h->d_twn, h->d_tspec, h->d_shifts, h->d_work_list,
                                            h->d_work_count, h->d_corr_stats, dump_xhat, dump_corr,
                                            dump_template, h->n_cu, h->stream));
    }
    {
        ProfScope p(h, 3);
        HIP_TRY(thr::launch_finish(n_blocks * h->cfg.n_templates, h->dev, h->d_corr_stats, d_out,
                                   h->d_work_count, h->stream));
    }
    return THR_OK;
}

int run_batch(thr_handle* h, const void* d_samples, int format, const long long* d_block_idx,
              int n_blocks, thr_record* d_out, float2* dump_fft, float2* dump_xhat,
              float2* dump_corr, int dump_template, bool carrier_only, size_t stride = 0) {
    // stride 0: blocks packed back to back; otherwise raw-stream framing (overlapping blocks)
    h->dev.blk_stride = stride ? stride : size_t(h->cfg.block_len) * (format == THR_IN_U8 ? 2 : 8);
#ifdef THR_DEV
    // dev A/B only (results are those of block 0): every block reads the SAME samples, which
    // then come from L2 -- what is left of a kernel's time is what it costs WITHOUT its HBM fetch
    static const bool stride0 = getenv("THR_DEV_STRIDE0") != nullptr;
    if (stride0) h->dev.blk_stride = 0;
#endif
    return (h->small ? run_batch_small : h->fast ? run_batch_fast : h->lng ? run_batch_long : run_batch_generic)(
        h, d_samples, format, d_block_idx, n_blocks, d_out, dump_fft, dump_xhat, dump_corr,
        dump_template, carrier_only);
}

}  // namespace

extern "C" {

int thr_abi_version(void) { return THR_ABI_VERSION; }

const char* thr_last_error(void) { return g_last_error.c_str(); }

const char* thr_kernel_name(int slot) {
    static const char* names[THR_N_KERNEL_SLOTS] = {"k_carrier", "k_fit",    "k_correlate",
                                                    "k_finish",  "k_correlate_sub+k_combine (small batches)"};
    return (slot >= 0 && slot < THR_N_KERNEL_SLOTS) ? names[slot] : "";
}

static int create_impl(const thr_settings* s, int preshift_num, thr_handle** out, int variant = -1,
                       int path = THR_PATH_AUTO, int interp = 0);

int thr_create(const thr_settings* s, thr_handle** out) { return create_impl(s, 0, out); }

static int create_fastdet(const thr_settings* s, thr_handle** out, int path) {
    if (!s || !out) return fail(THR_ERR_ARG, "thr_create_fastdet: null argument");
    if (s->n_templates != 1) return fail(THR_ERR_ARG, "the fastdet variant takes exactly one template");
    if (s->carrier_thresh[2] != 0.0 || s->corr_thresh[2] != 0.0)
        return fail(THR_ERR_ARG, "fastdet thresholds are constant + snr * noise_power (no stddev term)");
    if (s->carrier_window[0] < 0 && s->carrier_window[1] >= 0)   // cardet.c:44-48
        return fail(THR_ERR_ARG, "Carrier frequency window range not supported.");
    return create_impl(s, 1, out, 2, path);
}

int thr_create_fastdet(const thr_settings* s, thr_handle** out) try {
    return create_fastdet(s, out, THR_PATH_AUTO);
} catch (...) {
    return thr::on_exception("thr_create_fastdet");
}

static int create_preshift(const thr_settings* s, int num_arg, thr_handle** out, int path) {
    const int num_shifts = num_arg & 0xFFFF, interp = (num_arg >> 16) & 0xFFFF;
    if (num_arg < 0 || interp > THR_INTERP_COSINE)
        return fail(THR_ERR_ARG, "unknown carrier interpolator %d", interp);
    if (num_shifts < 1 || num_shifts > 4096)
        return fail(THR_ERR_ARG, "num_shifts %d out of range [1, 4096]", num_shifts);
    if (s && s->n_templates != 1)
        return fail(THR_ERR_ARG, "the preshift variant takes exactly one template");
    return create_impl(s, num_shifts, out, -1, path, interp);
}

int thr_create_preshift(const thr_settings* s, int num_shifts, thr_handle** out) try {
    return create_preshift(s, num_shifts, out, THR_PATH_AUTO);
} catch (...) {
    return thr::on_exception("thr_create_preshift");
}

namespace {
constexpr uintptr_t kPage = 4096;
}
int thr_host_register(const void* p, size_t bytes) try {
    if (!p || bytes == 0) return fail(THR_ERR_ARG, "thr_host_register: empty range");
    const uintptr_t a = reinterpret_cast<uintptr_t>(p) & ~(kPage - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + bytes + kPage - 1) & ~(kPage - 1);
    const hipError_t rc = hipHostRegister(reinterpret_cast<void*>(a), size_t(e - a), hipHostRegisterDefault);
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        return fail(THR_ERR_DEVICE, "hipHostRegister(%zu bytes) failed: %s", size_t(e - a), hipGetErrorString(rc));
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_host_register");
}

int thr_host_unregister(const void* p) try {
    if (!p) return fail(THR_ERR_ARG, "thr_host_unregister: null");
    const uintptr_t a = reinterpret_cast<uintptr_t>(p) & ~(kPage - 1);
    const hipError_t rc = hipHostUnregister(reinterpret_cast<void*>(a));
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        return fail(THR_ERR_DEVICE, "hipHostUnregister failed: %s", hipGetErrorString(rc));
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_host_unregister");
}

int thr_input_window(thr_handle* h, const void* p, size_t bytes) try {
    return thr_input_window_ex(h, p, bytes, 0, 0);
} catch (...) {
    return thr::on_exception("thr_input_window");
}

int thr_input_window_ex(thr_handle* h, const void* p, size_t bytes, int populate_threads, size_t segment_bytes) try {
    if (!h) return fail(THR_ERR_ARG, "thr_input_window: null handle");
    if (populate_threads < 0 || populate_threads > 16)
        return fail(THR_ERR_ARG, "thr_input_window_ex: populate_threads %d out of range [0, 16]", populate_threads);
    if (segment_bytes && (segment_bytes < (size_t(1) << 16) || (segment_bytes & (segment_bytes - 1))))
        return fail(THR_ERR_ARG, "thr_input_window_ex: segment_bytes %zu is not a power of two >= 64 KiB", segment_bytes);
    if (hipSetDevice(h->device) != hipSuccess) return fail(THR_ERR_DEVICE, "hipSetDevice(%d) failed", h->device);
    if (h->hp.async_open != 0)
        return fail(THR_ERR_STATE, "thr_input_window: %d submitted batch(es) not collected yet", h->hp.async_open);
    if (h->hp.copy) (void)hipStreamSynchronize(h->hp.copy);     // no copy may still read the old window
    h->win.close();
    for (auto& e : h->hp.win_end) e = 0;
    for (auto& e : h->hp.win_lo) e = 0;
    if (p && bytes)
        h->win.open(p, bytes, h->device, populate_threads ? populate_threads : InputWindow::kPopulators, segment_bytes);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_input_window_ex");
}

int thr_debug_window(thr_handle* h, size_t out[4]) try {
    if (!h || !out) return fail(THR_ERR_ARG, "thr_debug_window: null argument");
    std::lock_guard<std::mutex> lk(h->win.mu);
    out[0] = h->win.base ? h->win.consumed * h->win.kSeg : 0;
    out[1] = h->win.reg_lo * h->win.kSeg;
    out[2] = h->win.reg_hi * h->win.kSeg;
    out[3] = h->win.base ? h->win.kSeg : 0;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_debug_window");
}

int thr_debug_correlate_geom(thr_handle* h, int* rows_lo, int* rows_hi) try {
    if (!h || !rows_lo || !rows_hi) return fail(THR_ERR_ARG, "thr_debug_correlate_geom: null argument");
    *rows_lo = *rows_hi = -1;
    int lo = -1, hi = -1;
    bool got = false;
    if (h->fast && !h->preshift_num)
        got = thr::correlate_geom_16k(h->dev, &lo, &hi);
    else if (h->seg)
        got = thr::correlate_geom_seg(h->dev, &lo, &hi);
    if (got) {
        *rows_lo = lo;
        *rows_hi = hi;
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_debug_correlate_geom");
}

int thr_debug_sections(thr_handle* h, int* n_sections, int* section_len) try {
    if (!h || !n_sections || !section_len) return fail(THR_ERR_ARG, "thr_debug_sections: null argument");
    *n_sections = (h->seg || h->sec4k) ? h->dev.n_seg : 0;
    *section_len = h->sec4k ? 4096 : h->seg ? 16384 : 0;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_debug_sections");
}

int thr_get_path_info(thr_handle* h, thr_path_info* out) try {
    if (!h || !out) return fail(THR_ERR_ARG, "thr_get_path_info: null argument");
    std::memset(out, 0, sizeof(*out));
    out->n_templates = h->cfg.n_templates;
    out->n_sections = (h->seg || h->sec4k) ? h->dev.n_seg : 0;
    out->section_len = h->sec4k ? 4096 : h->seg ? 16384 : 0;
    out->why_unsectioned = h->why_unsectioned;
    out->rows_lo = out->rows_hi = -1;
    const int n = h->cfg.block_len;
    const char* car;
    const char* cor;
    if (h->preshift_num && h->fast) {
        car = cor = "k_preshift";
    } else if (h->fast) {
        car = h->dev.car_prune == 1 ? "k_carrier_pruned" : h->dev.car_prune == 2 ? "k_carrier_pruned (pre-shifted window)"
                                                                                  : "k_carrier";
        cor = h->sec4k ? "k_correlate_4k" : "k_correlate";
        if (!h->sec4k) thr::correlate_geom_16k(h->dev, &out->rows_lo, &out->rows_hi);
    } else if (h->lng && !h->preshift_num) {
        car = h->dev.car_prune == 1 ? "k_carrier_dit+k_select_dit" : "k_carrier_sub+k_select";
        cor = h->seg ? "k_correlate_seg" : "k_correlate_sub";
        if (h->seg) thr::correlate_geom_seg(h->dev, &out->rows_lo, &out->rows_hi);
    } else if (h->small) {
        car = "k_carrier_small";
        cor = "k_correlate_small";
    } else {
        car = "g_* (multi-pass)";
        cor = "g_* (multi-pass)";
    }
    std::snprintf(out->carrier_kernel, sizeof(out->carrier_kernel), "%s", car);
    std::snprintf(out->correlate_kernel, sizeof(out->correlate_kernel), "%s", cor);
    static const char* const why[] = {
        "",
        "the handle was created with an unsectioned / multi-pass kernel path",
        "preshift / fastdet variant: one fused kernel per block",
        "corr_thresh has a stddev term, whose sums run over every kept lag",
        "the unique window of this history / template length needs more sections than pay",
        "this block length has no sectioned form"};
    if (out->n_sections)
        std::snprintf(out->text, sizeof(out->text),
                      "block_len %d, %d template(s): carrier stage %s, correlate stage %s in %d sections of %d samples",
                      n, h->cfg.n_templates, car, cor, out->n_sections, out->section_len);
    else if (out->rows_lo >= 0)
        std::snprintf(out->text, sizeof(out->text),
                      "block_len %d, %d template(s): carrier stage %s, correlate stage %s (window rows %d, %d), "
                      "unsectioned: %s",
                      n, h->cfg.n_templates, car, cor, out->rows_lo, out->rows_hi, why[h->why_unsectioned]);
    else
        std::snprintf(out->text, sizeof(out->text),
                      "block_len %d, %d template(s): carrier stage %s, correlate stage %s, unsectioned: %s", n,
                      h->cfg.n_templates, car, cor, why[h->why_unsectioned]);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_get_path_info");
}

int thr_debug_pipe_times(thr_handle* h, double out[16]) try {
    if (!h || !out) return fail(THR_ERR_ARG, "thr_debug_pipe_times: null argument");
    for (int i = 0; i < 8; ++i) {
        out[i] = h->t_pipe[i];
        out[8 + i] = h->t_pipe_max[i];
        h->t_pipe[i] = h->t_pipe_max[i] = 0;
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_debug_pipe_times");
}

int thr_debug_window_times(thr_handle* h, double out[6]) try {
    if (!h || !out) return fail(THR_ERR_ARG, "thr_debug_window_times: null argument");
    std::lock_guard<std::mutex> lk(h->win.mu);
    out[0] = h->win.t_populate;
    out[1] = h->win.t_register;
    out[2] = h->win.t_unregister;
    out[3] = h->win.t_acquire;
    out[4] = double(h->win.n_acquire_waits);
    out[5] = double(h->win.n_pageable);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_debug_window_times");
}

int thr_input_window_release(thr_handle* h) try {
    if (!h) return fail(THR_ERR_ARG, "thr_input_window_release: null handle");
    if (hipSetDevice(h->device) != hipSuccess) return fail(THR_ERR_DEVICE, "hipSetDevice(%d) failed", h->device);
    if (h->hp.async_open != 0)
        return fail(THR_ERR_STATE, "thr_input_window_release: %d submitted batch(es) not collected yet",
                    h->hp.async_open);
    if (h->hp.copy) (void)hipStreamSynchronize(h->hp.copy);     // no copy reads the window any more
    for (auto& e : h->hp.win_end) e = 0;
    for (auto& e : h->hp.win_lo) e = 0;
    h->win.release_all();
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_input_window_release");
}

int thr_plan_sections(int block_len, int history_len, int template_len, int* n_sections, int* start,
                      int* win_lo, int* win_hi, int* sum_lo, int* sum_hi) try {
    if (!n_sections || !start || !win_lo || !win_hi || !sum_lo || !sum_hi)
        return fail(THR_ERR_ARG, "thr_plan_sections: null argument");
    if (block_len <= 0 || (block_len & (block_len - 1)) || template_len < 1 || template_len > block_len ||
        history_len < template_len - 1 || history_len >= block_len)
        return fail(THR_ERR_ARG, "thr_plan_sections: bad geometry (%d, %d, %d)", block_len, history_len,
                    template_len);
    thr::DevCfg d{};
    d.block_len = block_len;
    d.corr_len = block_len - template_len + 1;
    const int pad = history_len - template_len + 1;   // soa_estimator.py:20-39
    d.corr_lo = pad / 2;
    d.corr_hi = d.corr_len - (pad - pad / 2);
    if (block_len == 16384)
        plan_sections_4k(d, template_len);   // (4096-sample sections; the sums are not sectioned: 0, 0)
    else
        plan_sections(d, template_len);
    *n_sections = d.n_seg;
    for (int g = 0; g < d.n_seg; ++g) {
        start[g] = d.seg_start[g];
        win_lo[g] = d.seg_lo[g] + d.seg_start[g];
        win_hi[g] = d.seg_hi[g] + d.seg_start[g];
        sum_lo[g] = d.seg_sum_lo[g] + d.seg_start[g];
        sum_hi[g] = d.seg_sum_hi[g] + d.seg_start[g];
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_plan_sections");
}

int thr_create_ex(const thr_settings* s, int variant, int variant_arg, int path, thr_handle** out) try {
    if (path != THR_PATH_AUTO && path != THR_PATH_MULTIPASS && path != THR_PATH_UNSECTIONED &&
        path != THR_PATH_GENERIC_ROWS && path != THR_PATH_UNSECTIONED_GENERIC_ROWS)
        return fail(THR_ERR_ARG, "thr_create_ex: unknown path %d", path);
    switch (variant) {
        case THR_VARIANT_DEFAULT: return create_impl(s, 0, out, -1, path);
        case THR_VARIANT_PRESHIFT: return create_preshift(s, variant_arg, out, path);
        case THR_VARIANT_FASTDET: return create_fastdet(s, out, path);
    }
    return fail(THR_ERR_ARG, "thr_create_ex: unknown variant %d", variant);
} catch (...) {
    return thr::on_exception("thr_create_ex");
}


static int create_body(thr_handle*& h, const thr_settings* s, int preshift_num, thr_handle** out, int variant,
                       int path, int interp);

static int create_impl(const thr_settings* s, int preshift_num, thr_handle** out, int variant, int path,
                       int interp) {
    if (!s || !out) return fail(THR_ERR_ARG, "thr_create: null argument");
    *out = nullptr;
    thr_handle* h = nullptr;       // what create_body had built when it threw (host memory) goes back
    try {
        return create_body(h, s, preshift_num, out, variant, path, interp);
    } catch (...) {
        if (h) thr_destroy(h);
        return thr::on_exception("thr_create");
    }
}

static int create_body(thr_handle*& h, const thr_settings* s, int preshift_num, thr_handle** out, int variant,
                       int path, int interp) {
    const int n = s->block_len;
    if (n <= 0 || (n & (n - 1))) return fail(THR_ERR_ARG, "block_len %d is not a power of two", n);
    if (n < 64 || n > (1 << 20))
        return fail(THR_ERR_ARG, "block_len %d out of range [64, 1048576]", n);
    if (s->n_templates < 1 || s->n_templates > thr::kMaxTemplates)
        return fail(THR_ERR_ARG, "n_templates %d out of range [1, %d]", s->n_templates,
                    thr::kMaxTemplates);
    if (!s->templates || s->template_len < 1 || s->template_len > n)
        return fail(THR_ERR_ARG, "bad template (len %d)", s->template_len);
    if (s->history_len < s->template_len - 1 || s->history_len >= n)
        return fail(THR_ERR_ARG, "history_len %d must satisfy template_len-1 <= history_len < block_len",
                    s->history_len);  // soa_estimator.py:32 asserts the lower bound
    if (s->max_batch < 1) return fail(THR_ERR_ARG, "max_batch must be >= 1");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(THR_ERR_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (s->device_id < 0 || s->device_id >= ndev)
        return fail(THR_ERR_ARG, "device_id %d out of range (%d devices)", s->device_id, ndev);

    h = new thr_handle();
    h->cfg = *s;
    h->cfg.templates = nullptr;  // not retained beyond this call (re-pointed below)
    h->device = s->device_id;
    h->preshift_num = preshift_num;
    h->path = path;
    const bool multipass = path == THR_PATH_MULTIPASS;
    h->fast = (n == 16384) && !multipass;
    // the preshift variant has a fused kernel for 16384 only; other lengths use the multi-pass pipeline
    h->lng = thr::long_supported(n) && !multipass && !preshift_num;
    int rc = THR_OK;
    do {
        if (hipSetDevice(h->device) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "hipSetDevice(%d) failed", h->device);
            break;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "hipGetDeviceProperties failed");
            break;
        }
        h->n_cu = prop.multiProcessorCount;
        if ((h->fast || h->lng || thr::small_supported(n)) && size_t(prop.maxSharedMemoryPerMultiProcessor) <
                           thr::lds_bytes_16k()) {
            rc = fail(THR_ERR_DEVICE, "device has %zu B LDS per CU, need %zu",
                      size_t(prop.maxSharedMemoryPerMultiProcessor), thr::lds_bytes_16k());
            break;
        }
        thr::DevCfg& d = h->dev;
        d.block_len = n;
        d.history_len = s->history_len;
        d.n_templates = s->n_templates;
        d.carrier_len = s->carrier_len > 0 ? s->carrier_len : s->template_len;
        if ((rc = window_indices(s->carrier_window[0], s->carrier_window[1], n, &d.win_lo,
                                 &d.win_count)) != THR_OK)
            break;
        // soa_estimator.py:20-39
        const int corr_len = n - s->template_len + 1;
        const int pad = s->history_len - s->template_len + 1;
        d.corr_lo = pad / 2;
        d.corr_hi = corr_len - (pad - pad / 2);
        d.corr_len = corr_len;
        if (d.corr_hi <= d.corr_lo) {
            rc = fail(THR_ERR_ARG, "empty correlation window [%d, %d)", d.corr_lo, d.corr_hi);
            break;
        }
        for (int i = 0; i < 3; ++i) {
            d.car_thr[i] = s->carrier_thresh[i];
            d.cor_thr[i] = s->corr_thresh[i];
        }
#ifdef THR_DEV
        d.timeline = nullptr;
        if (hipMalloc(&d.timeline, 128 * sizeof(unsigned long long)) == hipSuccess)
            hipMemset(d.timeline, 0, 128 * sizeof(unsigned long long));
#endif
        d.variant = variant >= 0 ? variant : (preshift_num ? 1 : 0);
        d.interp = d.variant == 1 ? interp : 0;
        d.car_want_std = s->carrier_thresh[2] != 0.0;
        d.car_prune = 0;
        bool prune_ok = !d.car_want_std;
#ifdef THR_DEV
        if (getenv("THR_NO_PRUNE")) prune_ok = false;   // dev A/B: the full-spectrum carrier kernel
#endif
        if (prune_ok) {
            // long blocks: R0 sub-transforms, each pruned to its 128 lowest bins (mode 1 only)
            const int span = 128 * (h->lng ? n / 16384 : 1);
            if (d.win_lo >= 3 && d.win_lo + d.win_count + 3 <= span)
                d.car_prune = 1;  // window and fit margin already inside bins [0, span)
            else if (d.win_count + 6 <= 128 && !h->lng)
                d.car_prune = 2;  // any narrow window: pre-shift by win_lo - 3
        }
        d.cor_want_std = s->corr_thresh[2] != 0.0;
        d.no_row_geom = path == THR_PATH_GENERIC_ROWS || path == THR_PATH_UNSECTIONED_GENERIC_ROWS;
        h->small = thr::small_supported(n) && !multipass && !preshift_num;
        // long blocks: the correlate stage in overlap-save sections wherever the template allows
        const bool unsectioned = path == THR_PATH_UNSECTIONED || path == THR_PATH_UNSECTIONED_GENERIC_ROWS;
        h->seg = h->lng && !unsectioned && plan_sections(d, s->template_len);
        if (!h->seg) d.n_seg = 0;
        // block_len 16384, one short template, no stddev term: the correlate stage as 4096-sample
        // sections (detect16k_sec.hip); stage dumps and every other launch keep k_correlate
        h->sec4k = h->fast && !preshift_num && d.variant == 0 && !d.cor_want_std &&
                   !unsectioned && plan_sections_4k(d, s->template_len);
        if (!h->seg && !h->sec4k) d.n_seg = 0;
        // why not sectioned: the first reason that applies (thr_get_path_info)
        h->why_unsectioned = (h->seg || h->sec4k)                       ? THR_WHY_SECTIONED
                             : (multipass || unsectioned)               ? THR_WHY_PATH
                             : (preshift_num || d.variant != 0)         ? THR_WHY_VARIANT
                             : !(h->fast || h->lng)                     ? THR_WHY_BLOCK_LEN
                             : (h->fast && d.cor_want_std)              ? THR_WHY_STDDEV
                                                                        : THR_WHY_GEOMETRY;

        h->cfg.templates = s->templates;
        rc = build_constants(h);
        if (rc == THR_OK && preshift_num) rc = build_preshift_bank(h);
        h->cfg.templates = nullptr;
        if (rc != THR_OK) break;

#define CREATE_TRY(expr)                                                              \
    if ((expr) != hipSuccess) {                                                       \
        rc = fail(THR_ERR_DEVICE, "%s failed (%s)", #expr, hipGetErrorString(hipGetLastError())); \
        break;                                                                        \
    }
        // (the > 64 KiB dynamic-LDS opt-in is per device and kernel, not per handle: each family of
        // kernels is prepared once per device and process -- dozens of hipFuncSetAttribute calls
        // that a second handle on the same device need not repeat)
        static std::mutex prep_mu;
        static std::vector<unsigned> prepared;      // per device: bit per kernel family / block length
        auto once = [&](unsigned bit, auto&& fn) -> hipError_t {
            std::lock_guard<std::mutex> lk(prep_mu);
            if (prepared.size() <= size_t(h->device)) prepared.resize(size_t(h->device) + 1, 0u);
            if (prepared[h->device] & bit) return hipSuccess;
            const hipError_t e = fn();
            if (e == hipSuccess) prepared[h->device] |= bit;
            return e;
        };
        const unsigned len_bit = 1u << (8 + (31 - __builtin_clz(unsigned(n))) % 20);   // (per block length)
        if (h->fast) CREATE_TRY(once(1u, [] { return thr::prepare_16k(); }));
        if (h->fast && preshift_num) CREATE_TRY(once(2u, [] { return thr::prepare_preshift_16k(); }));
        if (h->small) CREATE_TRY(once(len_bit, [&] { return thr::prepare_small(n); }));
        if (h->seg) {
            CREATE_TRY(once(4u, [] { return thr::prepare_seg(); }));
        }
        if (h->lng) {
            CREATE_TRY(once(len_bit, [&] { return thr::prepare_long(n); }));
            const int r0 = n / 16384;
            // sub-batch: large enough to amortise the kernels' launch latency, ramps and tails
            // (the exchange rows no longer grow with it -- one row set per workgroup -- so the
            // sub-batch is as large as the small per-block buffers allow: fewer kernel ramps and
            // tails, +3.7 % from 4096 to 16384 blocks)
            h->long_batch = std::min(s->max_batch, std::max(64, 16384 / s->n_templates));
            const size_t lb = size_t(h->long_batch);
            const size_t win_w = size_t(std::min(h->dev.win_count + 6, n));
            // (the decimation-in-time carrier stage parks R0 complex values per window bin here)
            CREATE_TRY(hipMalloc(&h->d_win_pow, lb * win_w * sizeof(float) * 2 * r0));
            CREATE_TRY(hipMalloc(&h->d_partial, lb * r0 * 2 * sizeof(float)));
            h->long_chunk = std::min(h->long_batch, thr::long_chunk_blocks(n, s->n_templates));
            const size_t lc = size_t(h->long_chunk);
            // (one chunk of the two-kernel form, or one row per workgroup of the fused form)
            const size_t rows = std::max(lc, size_t(std::min(h->long_batch, h->n_cu)));
            CREATE_TRY(hipMalloc(&h->d_dsub, rows * s->n_templates * size_t(n) * sizeof(float2)));
            CREATE_TRY(hipMalloc(&h->d_xhat_scratch, size_t(h->n_cu) * 16384 * sizeof(float2)));
            if (h->seg)
                CREATE_TRY(hipMalloc(&h->d_seg_stats, lb * s->n_templates * size_t(d.n_seg) * sizeof(thr::CorrStats)));
        }
        if (!h->fast && !h->lng) {
            // sub-batch so that the 3 ping-pong buffers stay near 256 MiB (Infinity-Cache sized)
            const size_t per_block = size_t(3) * n * sizeof(float2);
            h->gen_batch = int(std::max<size_t>(1, std::min<size_t>(size_t(s->max_batch),
                                                                    (size_t(256) << 20) / per_block)));
            if (!h->small)   // (short blocks run LDS-resident)
                CREATE_TRY(hipMalloc(&h->d_gen_scratch, thr::generic_scratch_bytes(n, h->gen_batch)));
        }
        CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        h->stream = h->own_stream;
        const size_t mb = size_t(s->max_batch);
        CREATE_TRY(hipMalloc(&h->d_stats, mb * sizeof(thr::CarStats)));
        CREATE_TRY(hipMalloc(&h->d_shifts, mb * sizeof(thr::ShiftParams)));
        CREATE_TRY(hipMalloc(&h->d_corr_stats, mb * s->n_templates * sizeof(thr::CorrStats)));
        if (h->sec4k)
            CREATE_TRY(hipMalloc(&h->d_seg_stats, mb * s->n_templates * size_t(d.n_seg) * sizeof(thr::CorrStats)));
        CREATE_TRY(hipMalloc(&h->d_work_list, mb * sizeof(int)));
        CREATE_TRY(hipMalloc(&h->d_work_count, 4 * sizeof(int)));  // [0] work count, [1] dynamic cursor
        CREATE_TRY(hipMemset(h->d_work_count, 0, 4 * sizeof(int)));  // re-armed by k_finish
        CREATE_TRY(hipMalloc(&h->d_ncompact, sizeof(int)));
#undef CREATE_TRY
    } while (0);
    if (rc != THR_OK) {
        thr_handle* dead = h;
        h = nullptr;
        thr_destroy(dead);
        return rc;
    }
    *out = h;
    return THR_OK;
}

void thr_destroy(thr_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    if (h->hp.copy) (void)hipStreamSynchronize(h->hp.copy);
    h->win.close();
    if (h->hp.ready || h->hp.copy) {
        auto& p = h->hp;
        if (p.copy) {
            (void)hipStreamSynchronize(p.copy);
            (void)hipStreamDestroy(p.copy);
        }
        if (p.h_bad) (void)hipHostFree(p.h_bad);
        for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
            if (p.ev_h2d[b]) (void)hipEventDestroy(p.ev_h2d[b]);
            if (p.ev_done[b]) (void)hipEventDestroy(p.ev_done[b]);
            if (p.h_rec[b]) (void)hipHostFree(p.h_rec[b]);
            if (p.h_meta[b]) (void)hipHostFree(p.h_meta[b]);
            for (void* q : {p.d_in[b], static_cast<void*>(p.d_idx[b]), static_cast<void*>(p.d_rec[b]),
                            static_cast<void*>(p.d_text[b]), static_cast<void*>(p.d_bad[b])})
                if (q) (void)hipFree(q);
        }
    }
    for (auto& v : h->pending)
        for (auto& e : v) h->free_events.push_back(e);
    for (auto& e : h->free_events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    void* bufs[] = {h->d_tables, h->d_twn, h->d_tspec, h->d_stats, h->d_shifts, h->d_corr_stats, h->d_gen_scratch, h->d_tspec_nat, h->d_bank, h->d_gtw, h->d_tspec16k, h->d_tspec4k, h->d_ctab_pair, h->d_park, h->d_seg_stats, h->d_win_pow, h->d_partial, h->d_dsub, h->d_work_list,
                    h->d_work_count, h->d_xhat_scratch, h->d_ncompact, h->d_compact_tiles, h->d_in, h->d_idx, h->d_rec, h->d_forced};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int thr_set_wait_mode(thr_handle* h, int sleeping) try {
    if (!h) return fail(THR_ERR_ARG, "thr_set_wait_mode: null handle");
    h->sleepy_waits = sleeping != 0;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_set_wait_mode");
}

int thr_get_settings(const thr_handle* h, thr_settings* out) try {
    if (!h || !out) return fail(THR_ERR_ARG, "thr_get_settings: null argument");
    *out = h->cfg;
    out->templates = nullptr;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_get_settings");
}

int thr_set_stream(thr_handle* h, void* hip_stream) try {
    if (!h) return fail(THR_ERR_ARG, "null handle");
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_set_stream");
}

int thr_set_stream_default(thr_handle* h) try {
    if (!h) return fail(THR_ERR_ARG, "null handle");
    h->stream = nullptr;   // the legacy default stream (handle value 0): ordered with every blocking stream
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_set_stream_default");
}

int thr_sync(thr_handle* h) try {
    if (!h) return fail(THR_ERR_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_sync");
}

int thr_detect_device(thr_handle* h, const void* d_samples, int format,
                      const int64_t* d_block_idx, size_t n_blocks, thr_record* d_out) try {
    if (!h || !d_samples || !d_out) return fail(THR_ERR_ARG, "thr_detect_device: null argument");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    if (n_blocks == 0) return THR_OK;
    if (n_blocks > size_t(h->cfg.max_batch))
        return fail(THR_ERR_ARG, "n_blocks %zu exceeds max_batch %d", n_blocks, h->cfg.max_batch);
    HIP_TRY(hipSetDevice(h->device));
    return run_batch(h, d_samples, format, reinterpret_cast<const long long*>(d_block_idx),
                     int(n_blocks), d_out, nullptr, nullptr, nullptr, 0, false);
} catch (...) {
    return thr::on_exception("thr_detect_device");
}

// Raw-stream framing on the device (block_data.py:70-98; fastcard raw_reader.c:15-46): block i
// is the 2N bytes that start 2 (N - H) i bytes into the stream, so consecutive blocks overlap
// by H samples and the history copy of the host-side readers disappears.
static int stream_stride(thr_handle* h, size_t* stride) {
    const size_t s = size_t(h->cfg.block_len - h->cfg.history_len) * 2;
    if (s % 4 != 0)
        return fail(THR_ERR_ARG, "raw-stream framing needs an even block_len - history_len (got %d)",
                    h->cfg.block_len - h->cfg.history_len);
    *stride = s;
    return THR_OK;
}

int thr_detect_stream_device(thr_handle* h, const uint8_t* d_stream, const int64_t* d_block_idx,
                             size_t n_blocks, thr_record* d_out) try {
    if (!h || !d_stream || !d_out) return fail(THR_ERR_ARG, "thr_detect_stream_device: null argument");
    if (n_blocks == 0) return THR_OK;
    if (n_blocks > size_t(h->cfg.max_batch))
        return fail(THR_ERR_ARG, "n_blocks %zu exceeds max_batch %d", n_blocks, h->cfg.max_batch);
    if (reinterpret_cast<uintptr_t>(d_stream) % 4 != 0)
        return fail(THR_ERR_ARG, "stream pointer must be 4-byte aligned");
    size_t stride = 0;
    int rc = stream_stride(h, &stride);
    if (rc != THR_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    return run_batch(h, d_stream, THR_IN_U8, reinterpret_cast<const long long*>(d_block_idx),
                     int(n_blocks), d_out, nullptr, nullptr, nullptr, 0, false, stride);
} catch (...) {
    return thr::on_exception("thr_detect_stream_device");
}

// ---- one chunk of each host entry point: stage the inputs into pipe buffer b (copy stream), run
// the batch (main stream), start the records' way back.  Shared by the synchronous loops and by
// thr_submit*().  Every HIP failure comes back as a status: the callers all leave through
// pipe_finish() / pipe_abort(), which synchronise both streams and clear what is pending (a
// chunk must never stay pending with `pend_dst` pointing into a caller array that is gone).
static int chunk_samples(thr_handle* h, int b, const void* src, int format, size_t blk_bytes, size_t stride,
                         const int64_t* block_idx, int64_t first_idx, size_t nb, thr_record* dst,
                         size_t first) {
    auto& p = h->hp;
    // dense blocks: nb * blk_bytes; raw stream: (nb - 1) strides + one whole block
    const size_t bytes = stride ? (nb - 1) * stride + blk_bytes : nb * blk_bytes;
    int rc;
    double t0 = InputWindow::now_s(), t1;
    auto lap = [&](int k) {
        t1 = InputWindow::now_s();
        h->t_pipe[k] += t1 - t0;
        h->t_pipe_max[k] = std::max(h->t_pipe_max[k], t1 - t0);
        t0 = t1;
    };
    if ((rc = pipe_grow(&p.d_in[b], &p.in_bytes[b], bytes)) != THR_OK) return rc;
    lap(0);
    if ((rc = pipe_h2d(h, b, p.d_in[b], src, bytes)) != THR_OK) return rc;
    lap(1);
    for (size_t i = 0; i < nb; ++i)
        p.h_meta[b][i] = block_idx ? (long long)block_idx[i] : (long long)(first_idx + int64_t(i));
    lap(7);
    HIP_TRY(hipMemcpyAsync(p.d_idx[b], p.h_meta[b], nb * sizeof(long long), hipMemcpyHostToDevice, p.copy));
    lap(2);
    // (raw streams: hipStreamWaitEvent in here is where this thread meets the device's pace -- it
    // returns ~0.45 ms late per 2048-block chunk whatever precedes it on either stream, whatever
    // engine does the copy; profiles/README.md, round 5)
    if ((rc = pipe_inputs_enqueued(h, b)) != THR_OK) return rc;
    lap(6);
    rc = run_batch(h, p.d_in[b], format, p.d_idx[b], int(nb), p.d_rec[b], nullptr, nullptr, nullptr, 0,
                   false, stride);
    if (rc != THR_OK) return rc;
    lap(3);
    rc = pipe_records_enqueued(h, b, dst, nb * size_t(h->cfg.n_templates), first, false);
    lap(4);
    h->t_pipe[5] += 1;
    return rc;
}

static int chunk_card(thr_handle* h, int b, const char* text, size_t text_len, const int64_t* payload_off,
                      const int64_t* block_idx, size_t first, size_t nb, thr_record* dst) {
    auto& p = h->hp;
    const size_t out_bytes = size_t(h->cfg.block_len) * 2;
    const size_t chars = ((out_bytes + 2) / 3) * 4;  // base64 payload length of one block
    // contiguous span of text covering the chunk's payloads: [lo, hi + chars)
    long long lo = payload_off[first], hi = payload_off[first];
    for (size_t i = 0; i < nb; ++i) {
        const long long o = payload_off[first + i];
        if (o < 0 || size_t(o) + chars > text_len)
            return fail(THR_ERR_ARG, "payload %zu (offset %lld, %zu chars) lies outside the text",
                        first + i, o, chars);
        lo = std::min(lo, o);
        hi = std::max(hi, o);
    }
    const size_t span = size_t(hi - lo) + chars;
    int rc;
    double t0 = InputWindow::now_s(), t1;
    auto lap = [&](int k) {
        t1 = InputWindow::now_s();
        h->t_pipe[k] += t1 - t0;
        h->t_pipe_max[k] = std::max(h->t_pipe_max[k], t1 - t0);
        t0 = t1;
    };
    if ((rc = pipe_grow(reinterpret_cast<void**>(&p.d_text[b]), &p.text_bytes[b], span)) != THR_OK) return rc;
    if ((rc = pipe_grow(&p.d_in[b], &p.in_bytes[b], nb * out_bytes)) != THR_OK) return rc;
    if (!p.d_bad[b]) HIP_TRY(hipMalloc(&p.d_bad[b], sizeof(int)));
    lap(0);
    long long* meta = p.h_meta[b];
    for (size_t i = 0; i < nb; ++i) {
        meta[i] = block_idx ? (long long)block_idx[first + i] : (long long)(first + i);
        meta[nb + i] = payload_off[first + i] - lo;
    }
    lap(2);
    if ((rc = pipe_h2d(h, b, p.d_text[b], text + lo, span)) != THR_OK) return rc;
    lap(1);
    HIP_TRY(hipMemcpyAsync(p.d_idx[b], meta, 2 * nb * sizeof(long long), hipMemcpyHostToDevice, p.copy));
    HIP_TRY(hipMemsetAsync(p.d_bad[b], 0, sizeof(int), p.copy));
    if ((rc = pipe_inputs_enqueued(h, b)) != THR_OK) return rc;
    lap(2);
    HIP_TRY(thr::launch_b64_decode(p.d_text[b], p.d_idx[b] + nb, int(nb), int(out_bytes),
                                   static_cast<unsigned char*>(p.d_in[b]), p.d_bad[b], h->stream));
    rc = run_batch(h, p.d_in[b], THR_IN_U8, p.d_idx[b], int(nb), p.d_rec[b], nullptr, nullptr, nullptr,
                   0, false);
    if (rc != THR_OK) return rc;
    lap(3);
    rc = pipe_records_enqueued(h, b, dst, nb * size_t(h->cfg.n_templates), first, true);
    lap(4);
    h->t_pipe[5] += 1;
    return rc;
}

// entry checks shared by the synchronous host entry points: device, staging, no open tickets
static int pipe_enter_sync(thr_handle* h, const char* who) {
    HIP_TRY(hipSetDevice(h->device));
    const int rc = ensure_pipe(h);
    if (rc != THR_OK) return rc;
    if (h->hp.async_open != 0)
        return fail(THR_ERR_STATE, "%s: %d submitted batch(es) not collected yet (thr_collect first)",
                    who, h->hp.async_open);
    return THR_OK;
}

int thr_detect_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                      thr_record* out, size_t out_capacity, size_t* n_blocks_out) try {
    if (!h || !stream || !out || !n_blocks_out)
        return fail(THR_ERR_ARG, "thr_detect_stream: null argument");
    *n_blocks_out = 0;
    size_t stride = 0;
    int rc = stream_stride(h, &stride);
    if (rc != THR_OK) return rc;
    const size_t blk = size_t(h->cfg.block_len) * 2;
    if (n_bytes < blk) return THR_OK;
    const size_t n_blocks = (n_bytes - blk) / stride + 1;
    if (n_blocks > out_capacity)
        return fail(THR_ERR_ARG, "stream holds %zu blocks, records array only %zu", n_blocks,
                    out_capacity);
    if ((rc = pipe_enter_sync(h, "thr_detect_stream")) != THR_OK) return rc;
    const size_t nt = size_t(h->cfg.n_templates);
    int chunk = 0;
    for (size_t done = 0; done < n_blocks; ++chunk) {
        const int b = chunk % thr_handle::kPipeDepth;
        const size_t nb = std::min(n_blocks - done, pipe_chunk_blocks(h, stride));
        if ((rc = pipe_drain(h, b)) != THR_OK) break;        // buffer b's previous chunk is handed out
        rc = chunk_samples(h, b, stream + done * stride, THR_IN_U8, blk, stride, nullptr,
                           first_block_idx + int64_t(done), nb, out + done * nt, done);
        if (rc != THR_OK) break;
        done += nb;
    }
    rc = pipe_finish(h, rc);
    if (rc != THR_OK) return rc;
    *n_blocks_out = n_blocks;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_detect_stream");
}

int thr_detect(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
               size_t n_blocks, thr_record* out) try {
    if (!h || !samples || !out) return fail(THR_ERR_ARG, "thr_detect: null argument");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    int rc = pipe_enter_sync(h, "thr_detect");
    if (rc != THR_OK) return rc;
    const size_t blk_bytes = size_t(h->cfg.block_len) * (format == THR_IN_U8 ? 2 : 8);
    const size_t nt = size_t(h->cfg.n_templates);
    // chunk i + 1 is copied (copy stream; the call blocks while the pageable source is staged)
    // while the kernels of chunk i run; records return through pinned staging
    int chunk = 0;
    for (size_t done = 0; done < n_blocks; ++chunk) {
        const int b = chunk % thr_handle::kPipeDepth;
        const size_t nb = std::min(n_blocks - done, pipe_chunk_blocks(h, blk_bytes));
        if ((rc = pipe_drain(h, b)) != THR_OK) break;
        rc = chunk_samples(h, b, static_cast<const unsigned char*>(samples) + done * blk_bytes, format,
                           blk_bytes, 0, block_idx ? block_idx + done : nullptr, int64_t(done), nb,
                           out + done * nt, done);
        if (rc != THR_OK) break;
        done += nb;
    }
    return pipe_finish(h, rc);
} catch (...) {
    return thr::on_exception("thr_detect");
}

int thr_frame_card(const char* text, size_t text_len, int block_len, int at_eof, size_t max_records,
                   double* timestamps, int64_t* block_idx, int64_t* payload_off, size_t* n_records,
                   size_t* consumed) try {
    if (!text || !timestamps || !block_idx || !payload_off || !n_records || !consumed)
        return fail(THR_ERR_ARG, "thr_frame_card: null argument");
    if (block_len <= 0) return fail(THR_ERR_ARG, "thr_frame_card: bad block_len %d", block_len);
    const size_t chars = ((size_t(block_len) * 2 + 2) / 3) * 4;
    size_t pos = 0, n = 0;
    *n_records = 0;
    *consumed = 0;
    while (n < max_records && pos < text_len) {
        // a data line is `<ts> <idx> ` + exactly `chars` characters: its end follows from the two
        // spaces of the short header -- no 43 KB newline scan
        const char* line = text + pos;
        const size_t left = text_len - pos;
        size_t end;  // index of the line terminator (or text_len)
        const char* sp1 = nullptr;
        const char* sp2 = nullptr;
        if (line[0] >= '0' && line[0] <= '9') {
            sp1 = static_cast<const char*>(memchr(line, ' ', std::min<size_t>(left, 40)));
            if (sp1) sp2 = static_cast<const char*>(memchr(sp1 + 1, ' ', std::min<size_t>(size_t(line + left - sp1 - 1), 32)));
        }
        if (sp2 && size_t(sp2 + 1 - line) + chars <= left) {
            end = size_t(sp2 + 1 - line) + chars;
            if (!(end == left || line[end] == '\n' || (line[end] == '\r' && end + 1 < left && line[end + 1] == '\n')))
                sp2 = nullptr;   // not the fixed layout: take the general path
        } else {
            sp2 = nullptr;
        }
        if (!sp2) {
            const char* nl = static_cast<const char*>(memchr(line, '\n', left));
            if (!nl && !at_eof) break;                    // incomplete last line: wait for more text
            end = nl ? size_t(nl - line) : left;
            if (end > 0 && line[end - 1] == '\r') --end;
            const size_t next = nl ? size_t(nl - line) + 1 : left;
            if (end == 0 || line[0] == '#' || (end >= 19 && memcmp(line, "Using Volk machine:", 19) == 0) ||
                (end >= 6 && memcmp(line, "linux;", 6) == 0)) {
                pos += next;
                continue;
            }
            sp1 = static_cast<const char*>(memchr(line, ' ', end));
            sp2 = sp1 ? static_cast<const char*>(memchr(sp1 + 1, ' ', size_t(line + end - sp1 - 1))) : nullptr;
            // a bad line ends the call: the records framed BEFORE it are handed out first (the
            // reference's per-line loop had processed them) and the next call, which starts at the
            // bad line, reports it
            if (!sp1 || !sp2) {
                if (n) break;
                return fail(THR_ERR_ARG, "malformed .card line at byte %zu: %.60s", pos, std::string(line, std::min<size_t>(end, 60)).c_str());
            }
            if (size_t(line + end - (sp2 + 1)) != chars) {
                if (n) break;
                return fail(THR_ERR_ARG, "block %.*s: payload of %zu base64 characters, expected %zu (block_len %d)",
                            int(sp2 - sp1 - 1), sp1 + 1, size_t(line + end - (sp2 + 1)), chars, block_len);
            }
        } else if (end == left && !at_eof) {
            break;   // the payload is complete but its newline has not arrived: wait (the next read brings it)
        }
        double ts = 0;
        long long idx = 0;
        const auto r1 = std::from_chars(line, sp1, ts);
        const auto r2 = std::from_chars(sp1 + 1, sp2, idx);
        if (r1.ec != std::errc() || r1.ptr != sp1 || r2.ec != std::errc() || r2.ptr != sp2) {
            if (n) break;
            return fail(THR_ERR_ARG, "malformed .card header at byte %zu: %.40s", pos,
                        std::string(line, size_t(sp2 - line)).c_str());
        }
        timestamps[n] = ts;
        block_idx[n] = idx;
        payload_off[n] = (long long)(pos + size_t(sp2 + 1 - line));
        ++n;
        // step over the terminator
        size_t adv = size_t(sp2 + 1 - line) + chars;
        if (adv < left && line[adv] == '\r') ++adv;
        if (adv < left && line[adv] == '\n') ++adv;
        pos += adv;
    }
    *n_records = n;
    *consumed = pos;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_frame_card");
}


int thr_detect_card(thr_handle* h, const char* text, size_t text_len, const int64_t* payload_off,
                    const int64_t* block_idx, size_t n_blocks, thr_record* out) try {
    if (!h || !text || !payload_off || !out) return fail(THR_ERR_ARG, "thr_detect_card: null argument");
    int rc = pipe_enter_sync(h, "thr_detect_card");
    if (rc != THR_OK) return rc;
    const size_t chars = ((size_t(h->cfg.block_len) * 2 + 2) / 3) * 4;
    const size_t nt = size_t(h->cfg.n_templates);
    int chunk = 0;
    for (size_t done = 0; done < n_blocks; ++chunk) {
        const int b = chunk % thr_handle::kPipeDepth;
        const size_t nb = std::min(n_blocks - done, pipe_chunk_blocks(h, chars + 32));
        if ((rc = pipe_drain(h, b)) != THR_OK) break;
        rc = chunk_card(h, b, text, text_len, payload_off, block_idx, done, nb, out + done * nt);
        if (rc != THR_OK) break;
        done += nb;
    }
    return pipe_finish(h, rc);
} catch (...) {
    return thr::on_exception("thr_detect_card");
}

// ---------------------------------------------------------------------------------------------
// Asynchronous host boundary: thr_submit*() stages one batch and returns a ticket while the GPU
// works; thr_collect() hands that batch's records out.  kPipeDepth batches may be open.
// ---------------------------------------------------------------------------------------------
static int submit_enter(thr_handle* h, const char* who, size_t n_blocks, uint64_t* ticket, int* slot) {
    if (!ticket) return fail(THR_ERR_ARG, "%s: null ticket pointer", who);
    *ticket = 0;
    if (n_blocks > size_t(h->cfg.max_batch))
        return fail(THR_ERR_ARG, "%s: n_blocks %zu exceeds max_batch %d (one submit = one batch)", who,
                    n_blocks, h->cfg.max_batch);
    HIP_TRY(hipSetDevice(h->device));
    const int rc = ensure_pipe(h);
    if (rc != THR_OK) return rc;
    auto& p = h->hp;
    for (int k = 0; k < thr_handle::kPipeDepth; ++k) {
        const int b = int((p.next_ticket + uint64_t(k)) % thr_handle::kPipeDepth);
        if (p.pend_n[b] == 0 && p.slot_ticket[b] == 0) {
            *slot = b;
            return THR_OK;
        }
    }
    return fail(THR_ERR_STATE, "%s: %d batches are already in flight; thr_collect() one first", who,
                thr_handle::kPipeDepth);
}

// a failed submit must not leave a half-enqueued chunk behind: wait for the streams, clear the slot
static int submit_leave(thr_handle* h, int b, int rc, uint64_t* ticket) {
    auto& p = h->hp;
    if (rc != THR_OK) {
        (void)hipStreamSynchronize(p.copy);
        (void)hipStreamSynchronize(h->stream);
        p.pend_n[b] = 0;
        p.slot_ticket[b] = 0;
        return rc;
    }
    p.slot_ticket[b] = p.next_ticket++;
    p.async_open += 1;
    *ticket = p.slot_ticket[b];
    return THR_OK;
}

int thr_submit(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
               size_t n_blocks, thr_record* out, uint64_t* ticket) try {
    if (!h || !samples || !out) return fail(THR_ERR_ARG, "thr_submit: null argument");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    int b = 0;
    int rc = submit_enter(h, "thr_submit", n_blocks, ticket, &b);
    if (rc != THR_OK || n_blocks == 0) return rc;   // (an empty batch: ticket 0, nothing to collect)
    const size_t blk_bytes = size_t(h->cfg.block_len) * (format == THR_IN_U8 ? 2 : 8);
    rc = chunk_samples(h, b, samples, format, blk_bytes, 0, block_idx, 0, n_blocks, out, 0);
    return submit_leave(h, b, rc, ticket);
} catch (...) {
    return thr::on_exception("thr_submit");
}

int thr_submit_card(thr_handle* h, const char* text, size_t text_len, const int64_t* payload_off,
                    const int64_t* block_idx, size_t n_blocks, thr_record* out, uint64_t* ticket) try {
    if (!h || !text || !payload_off || !out) return fail(THR_ERR_ARG, "thr_submit_card: null argument");
    int b = 0;
    int rc = submit_enter(h, "thr_submit_card", n_blocks, ticket, &b);
    if (rc != THR_OK || n_blocks == 0) return rc;
    rc = chunk_card(h, b, text, text_len, payload_off, block_idx, 0, n_blocks, out);
    return submit_leave(h, b, rc, ticket);
} catch (...) {
    return thr::on_exception("thr_submit_card");
}

int thr_submit_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                      thr_record* out, size_t out_capacity, size_t* n_blocks_out, uint64_t* ticket) try {
    if (!h || !stream || !out || !n_blocks_out)
        return fail(THR_ERR_ARG, "thr_submit_stream: null argument");
    *n_blocks_out = 0;
    size_t stride = 0;
    int rc = stream_stride(h, &stride);
    if (rc != THR_OK) return rc;
    const size_t blk = size_t(h->cfg.block_len) * 2;
    const size_t n_blocks = n_bytes < blk ? 0 : (n_bytes - blk) / stride + 1;
    if (n_blocks > out_capacity)
        return fail(THR_ERR_ARG, "stream holds %zu blocks, records array only %zu", n_blocks,
                    out_capacity);
    int b = 0;
    rc = submit_enter(h, "thr_submit_stream", n_blocks, ticket, &b);
    if (rc != THR_OK || n_blocks == 0) return rc;
    rc = chunk_samples(h, b, stream, THR_IN_U8, blk, stride, nullptr, first_block_idx, n_blocks, out, 0);
    rc = submit_leave(h, b, rc, ticket);
    if (rc == THR_OK) *n_blocks_out = n_blocks;
    return rc;
} catch (...) {
    return thr::on_exception("thr_submit_stream");
}

int thr_collect(thr_handle* h, uint64_t ticket) try {
    if (!h) return fail(THR_ERR_ARG, "thr_collect: null handle");
    if (ticket == 0) return THR_OK;   // the ticket of an empty batch
    auto& p = h->hp;
    for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
        if (p.slot_ticket[b] != ticket) continue;
        HIP_TRY(hipSetDevice(h->device));
        int rc = pipe_drain(h, b);
        if (rc != THR_OK && p.pend_n[b] != 0) {   // the wait itself failed: nothing may stay pending
            (void)hipStreamSynchronize(h->stream);
            p.pend_n[b] = 0;
        }
        p.slot_ticket[b] = 0;
        p.async_open -= 1;
        return rc;
    }
    return fail(THR_ERR_STATE, "thr_collect: ticket %llu is not open (never issued, or collected already)",
                (unsigned long long)ticket);
} catch (...) {
    return thr::on_exception("thr_collect");
}

int thr_inputs_consumed(thr_handle* h, uint64_t ticket) try {
    if (!h) return fail(THR_ERR_ARG, "thr_inputs_consumed: null handle");
    if (ticket == 0) return THR_OK;
    auto& p = h->hp;
    for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
        if (p.slot_ticket[b] != ticket) continue;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipEventSynchronize(p.ev_h2d[b]));   // recorded behind the chunk's last H2D copy
        pipe_inputs_done(h, b);
        return THR_OK;
    }
    return fail(THR_ERR_STATE, "thr_inputs_consumed: ticket %llu is not open", (unsigned long long)ticket);
} catch (...) {
    return thr::on_exception("thr_inputs_consumed");
}

int thr_poll(thr_handle* h, uint64_t ticket, int* done) try {
    if (!h || !done) return fail(THR_ERR_ARG, "thr_poll: null argument");
    *done = 1;
    if (ticket == 0) return THR_OK;
    auto& p = h->hp;
    for (int b = 0; b < thr_handle::kPipeDepth; ++b) {
        if (p.slot_ticket[b] != ticket) continue;
        const hipError_t e = hipEventQuery(p.ev_done[b]);
        if (e == hipErrorNotReady) {
            *done = 0;
            return THR_OK;
        }
        if (e != hipSuccess) return fail(THR_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
        return THR_OK;
    }
    return fail(THR_ERR_STATE, "thr_poll: ticket %llu is not open", (unsigned long long)ticket);
} catch (...) {
    return thr::on_exception("thr_poll");
}

// ---------------------------------------------------------------------------------------------
// .toad text (DetectionResult.serialize, toads_data.py:47-61) for a batch of detected records.
// ---------------------------------------------------------------------------------------------
extern "C++" {
namespace {
// Python's repr(float): the shortest digit string that round-trips (float_repr_style 'short'),
// fixed notation while -4 <= exponent10 < 16, else d.ddde+XX with at least two exponent digits
// (PyOS_double_to_string(x, 'r', 0, Py_DTSF_ADD_DOT_0)).
char* py_repr_double(char* out, double v) {
    if (std::isnan(v)) return static_cast<char*>(memcpy(out, "nan", 3)) + 3;
    if (std::isinf(v)) {
        const char* s = v < 0 ? "-inf" : "inf";
        const size_t n = strlen(s);
        return static_cast<char*>(memcpy(out, s, n)) + n;
    }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci - 1, v, std::chars_format::scientific);
    *r.ptr = 0;
    // sci = [-]d[.ddd]e[+-]XX
    char* s = sci;
    if (*s == '-') *out++ = *s++;
    char digits[24];
    int nd = 0;
    char* e = s;
    for (; e < r.ptr && *e != 'e'; ++e)
        if (*e != '.') digits[nd++] = *e;
    const int exp10 = atoi(e + 1);           // value = d.ddd x 10^exp10
    const int decpt = exp10 + 1;              // value = 0.dddd x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            memcpy(out, digits + 1, size_t(nd - 1));
            out += nd - 1;
        }
        *out++ = 'e';
        *out++ = exp10 < 0 ? '-' : '+';
        const int a = exp10 < 0 ? -exp10 : exp10;
        if (a >= 100) *out++ = char('0' + a / 100);
        *out++ = char('0' + (a / 10) % 10);
        *out++ = char('0' + a % 10);
        return out;
    }
    if (decpt <= 0) {
        *out++ = '0';
        *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        memcpy(out, digits, size_t(nd));
        return out + nd;
    }
    if (decpt >= nd) {
        memcpy(out, digits, size_t(nd));
        out += nd;
        for (int i = nd; i < decpt; ++i) *out++ = '0';
        *out++ = '.';
        *out++ = '0';
        return out;
    }
    memcpy(out, digits, size_t(decpt));
    out += decpt;
    *out++ = '.';
    memcpy(out, digits + decpt, size_t(nd - decpt));
    return out + (nd - decpt);
}
char* put_int(char* out, long long v) {
    const auto r = std::to_chars(out, out + 24, v);
    return r.ptr;
}
// "%.<DEC>f" of a finite double, correctly rounded (ties to even on the exact binary value, as
// glibc's printf and Python's '%.6f' do): v = m 2^-k exactly, so v 10^DEC = m 10^DEC / 2^k in
// 128-bit integer arithmetic.  Magnitudes the integers cannot hold take snprintf.
template <int DEC>
char* put_fixed(char* out, double v) {
    static_assert(DEC >= 1 && DEC <= 9, "10^DEC must fit 32 bits");
    int e = 0;
    const double fr = std::frexp(std::fabs(v), &e);            // |v| = fr 2^e, fr in [0.5, 1)
    const unsigned long long m = (unsigned long long)std::ldexp(fr, 53);   // 53-bit integer mantissa
    const int k = 53 - e;                                      // |v| = m 2^-k
    if (!std::isfinite(v) || k < 0 || k > 120 || e > 62) {
        if (std::isfinite(v) && k > 120) {                     // |v| < 2^-67: prints as zero
            if (std::signbit(v)) *out++ = '-';
            *out++ = '0';
            *out++ = '.';
            for (int i = 0; i < DEC; ++i) *out++ = '0';
            return out;
        }
        return out + snprintf(out, 64, "%.*f", DEC, v);
    }
    unsigned long long p10 = 1;
    for (int i = 0; i < DEC; ++i) p10 *= 10;
    const unsigned __int128 num = (unsigned __int128)m * p10;  // < 2^53 * 2^30
    unsigned __int128 q = k >= 128 ? 0 : num >> k;
    const unsigned __int128 rem = num - (q << k), half = (unsigned __int128)1 << (k - 1);
    if (k > 0 && (rem > half || (rem == half && (q & 1)))) ++q;
    const unsigned long long ip = (unsigned long long)(q / p10), fp = (unsigned long long)(q % p10);
    if (std::signbit(v)) *out++ = '-';
    out = std::to_chars(out, out + 24, ip).ptr;
    *out++ = '.';
    char d[DEC];
    unsigned long long f = fp;
    for (int i = DEC - 1; i >= 0; --i) {
        d[i] = char('0' + f % 10);
        f /= 10;
    }
    memcpy(out, d, DEC);
    return out + DEC;
}
}  // namespace
}  // extern "C++"

int thr_format_toad(const thr_record* recs, const double* timestamps, size_t n, int64_t new_len,
                    int with_rxid, int64_t rxid, int with_txid, int carrier_offset_f32, char* out,
                    size_t out_capacity, size_t* out_len) try {
    if ((!recs || !timestamps) && n) return fail(THR_ERR_ARG, "thr_format_toad: null argument");
    if (!out || !out_len) return fail(THR_ERR_ARG, "thr_format_toad: null output");
    *out_len = 0;
    if (out_capacity < n * size_t(THR_TOAD_LINE_MAX))
        return fail(THR_ERR_ARG, "thr_format_toad: %zu bytes for %zu lines, need %zu", out_capacity, n,
                    n * size_t(THR_TOAD_LINE_MAX));
    char* p = out;
    for (size_t i = 0; i < n; ++i) {
        const thr_record& r = recs[i];
        if (with_rxid) {
            p = put_int(p, rxid);
            *p++ = ' ';
        }
        if (with_txid) {
            p = put_int(p, r.template_id);
            *p++ = ' ';
        }
        if (!(std::fabs(timestamps[i]) < 1e15))
            return fail(THR_ERR_ARG, "thr_format_toad: timestamp %g of record %zu out of range", timestamps[i], i);
        p = put_fixed<6>(p, timestamps[i]);
        *p++ = ' ';
        p = put_int(p, r.block_idx);
        *p++ = ' ';
        // soa = new_len * block_idx + sample + offset: the integer part exactly, one float64 add (detect.py:69)
        const double soa = double(new_len * r.block_idx + int64_t(r.corr_sample)) + r.corr_offset;
        p = put_fixed<8>(p, soa);
        *p++ = ' ';
        p = put_int(p, r.corr_sample);
        *p++ = ' ';
        p = py_repr_double(p, r.corr_offset);
        *p++ = ' ';
        p = py_repr_double(p, double(r.corr_energy));
        *p++ = ' ';
        p = py_repr_double(p, double(r.corr_noise));
        *p++ = ' ';
        p = put_int(p, r.carrier_bin);
        *p++ = ' ';
        if (carrier_offset_f32 == 2 || (r.flags & THR_FLAG_INT_OFFSET))   // an int-typed offset (interpolator
                                                                          // `none`; cosine's `return 0`): "0"
            p = put_int(p, (long long)r.carrier_offset);
        else
            p = py_repr_double(p, carrier_offset_f32 ? double(float(r.carrier_offset)) : r.carrier_offset);
        *p++ = ' ';
        p = py_repr_double(p, double(r.carrier_energy));
        *p++ = ' ';
        p = py_repr_double(p, double(r.carrier_noise));
        *p++ = '\n';
    }
    *out_len = size_t(p - out);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_format_toad");
}

int thr_compact_device(thr_handle* h, const thr_record* d_in, size_t n_records, thr_record* d_out,
                       size_t* n_kept) try {
    if (!h || !d_in || !d_out || !n_kept) return fail(THR_ERR_ARG, "thr_compact_device: null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (n_records > size_t(1) << 30) return fail(THR_ERR_ARG, "too many records");
    const int tiles = thr::compact_tiles(int(n_records));
    if (tiles > h->compact_tiles_cap) {
        if (h->d_compact_tiles) (void)hipFree(h->d_compact_tiles);
        h->d_compact_tiles = nullptr;
        h->compact_tiles_cap = 0;
        HIP_TRY(hipMalloc(&h->d_compact_tiles, size_t(tiles) * sizeof(int)));
        h->compact_tiles_cap = tiles;
    }
    HIP_TRY(thr::launch_compact(d_in, int(n_records), d_out, h->d_ncompact, h->d_compact_tiles,
                                h->stream));
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, h->d_ncompact, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n_kept = size_t(n);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_compact_device");
}

int thr_profile_enable(thr_handle* h, int on) try {
    if (!h) return fail(THR_ERR_ARG, "null handle");
    h->prof_every = on < 0 ? 0 : on;
    h->batch_no = 0;
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_profile_enable");
}

int thr_profile_read(thr_handle* h, double ms[THR_N_KERNEL_SLOTS],
                     int64_t launches[THR_N_KERNEL_SLOTS]) try {
    if (!h || !ms || !launches) return fail(THR_ERR_ARG, "thr_profile_read: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int s = 0; s < THR_N_KERNEL_SLOTS; ++s) {
        for (auto& e : h->pending[s]) {
            float t = 0;
            if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) {
                h->ms[s] += t;
                h->launches[s] += 1;
            }
            h->free_events.push_back(e);
        }
        h->pending[s].clear();
        ms[s] = h->ms[s];
        launches[s] = h->launches[s];
        h->ms[s] = 0;
        h->launches[s] = 0;
    }
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_profile_read");
}

#ifdef THR_DEV
int thr_debug_timeline(thr_handle* h, unsigned long long* out128) try {
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    return hipMemcpy(out128, h->dev.timeline, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost) ==
                   hipSuccess ? 0 : -2;
} catch (...) {
    return thr::on_exception("thr_debug_timeline");
}
#endif

int thr_debug_fft(thr_handle* h, const void* samples, int format, size_t n_blocks,
                  float* spectra_out) try {
    if (!h || !samples || !spectra_out) return fail(THR_ERR_ARG, "thr_debug_fft: null argument");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    if (n_blocks > size_t(h->cfg.max_batch)) return fail(THR_ERR_ARG, "n_blocks exceeds max_batch");
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_staging(h, format);
    if (rc != THR_OK) return rc;
    const size_t n = size_t(h->cfg.block_len);
    const size_t blk_bytes = n * (format == THR_IN_U8 ? 2 : 8);
    float2* d_dump = nullptr;
    HIP_TRY(hipMalloc(&d_dump, n_blocks * n * sizeof(float2)));
    rc = THR_OK;
    do {
        if (hipMemcpyAsync(h->d_in, samples, n_blocks * blk_bytes, hipMemcpyHostToDevice, h->stream) !=
            hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "H2D copy failed");
            break;
        }
        rc = run_batch(h, h->d_in, format, nullptr, int(n_blocks), h->d_rec, d_dump, nullptr, nullptr,
                       0, true);
        if (rc != THR_OK) break;
        if (hipMemcpyAsync(spectra_out, d_dump, n_blocks * n * sizeof(float2), hipMemcpyDeviceToHost,
                           h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    (void)hipFree(d_dump);
    return rc;
} catch (...) {
    return thr::on_exception("thr_debug_fft");
}

int thr_detect_offsets(thr_handle* h, const void* samples, int format, const int64_t* block_idx,
                       size_t n_blocks, const double* carrier_offset, thr_record* out) try {
    if (!h || !samples || !carrier_offset || !out) return fail(THR_ERR_ARG, "thr_detect_offsets: null argument");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    if (n_blocks > size_t(h->cfg.max_batch))
        return fail(THR_ERR_ARG, "n_blocks %zu exceeds max_batch %d", n_blocks, h->cfg.max_batch);
    if (h->preshift_num) return fail(THR_ERR_ARG, "thr_detect_offsets: the default detector only (this variant "
                                                  "interpolates inside its fused kernel)");
    if (n_blocks == 0) return THR_OK;
    // (k_fit splits the shift into integer and fractional parts: a NaN or infinite offset has neither.
    // The reference's shifter raises on such a block -- int(round(nan)), carrier_sync.py:241-245)
    for (size_t i = 0; i < n_blocks; ++i)
        if (!std::isfinite(carrier_offset[i]))
            return fail(THR_ERR_ARG, "thr_detect_offsets: carrier_offset[%zu] is not finite", i);
    HIP_TRY(hipSetDevice(h->device));
    if (h->hp.async_open != 0)
        return fail(THR_ERR_STATE, "thr_detect_offsets: %d submitted batch(es) not collected yet", h->hp.async_open);
    int rc = ensure_staging(h, format);
    if (rc != THR_OK) return rc;
    if (!h->d_forced) HIP_TRY(hipMalloc(&h->d_forced, size_t(h->cfg.max_batch) * sizeof(double)));
    const size_t blk_bytes = size_t(h->cfg.block_len) * (format == THR_IN_U8 ? 2 : 8);
    const size_t nt = size_t(h->cfg.n_templates);
    std::vector<long long> idx(n_blocks);
    for (size_t i = 0; i < n_blocks; ++i) idx[i] = block_idx ? (long long)block_idx[i] : (long long)i;
    rc = THR_OK;
    do {
        if (hipMemcpyAsync(h->d_in, samples, n_blocks * blk_bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(h->d_idx, idx.data(), n_blocks * sizeof(long long), hipMemcpyHostToDevice, h->stream) !=
                hipSuccess ||
            hipMemcpyAsync(h->d_forced, carrier_offset, n_blocks * sizeof(double), hipMemcpyHostToDevice,
                           h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "staging failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        h->forced = h->d_forced;
        rc = run_batch(h, h->d_in, format, h->d_idx, int(n_blocks), h->d_rec, nullptr, nullptr, nullptr, 0, false);
        h->forced = nullptr;
        if (rc != THR_OK) break;
        if (hipMemcpyAsync(out, h->d_rec, n_blocks * nt * sizeof(thr_record), hipMemcpyDeviceToHost, h->stream) !=
                hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    h->forced = nullptr;
    if (rc != THR_OK) (void)hipStreamSynchronize(h->stream);
    return rc;
} catch (...) {
    return thr::on_exception("thr_detect_offsets");
}

int thr_debug_stage(thr_handle* h, const void* samples, int format, size_t n_blocks,
                    int template_id, float* shifted_fft_out, float* corr_out) try {
    return thr_debug_stage_offsets(h, samples, format, n_blocks, template_id, nullptr, shifted_fft_out, corr_out);
} catch (...) {
    return thr::on_exception("thr_debug_stage");
}

int thr_debug_stage_offsets(thr_handle* h, const void* samples, int format, size_t n_blocks, int template_id,
                            const double* carrier_offset, float* shifted_fft_out, float* corr_out) try {
    if (!h || !samples) return fail(THR_ERR_ARG, "thr_debug_stage: null argument");
    if (carrier_offset && h->preshift_num)
        return fail(THR_ERR_ARG, "thr_debug_stage_offsets: the default detector only");
    if (format != THR_IN_U8 && format != THR_IN_C64) return fail(THR_ERR_ARG, "bad format %d", format);
    if (n_blocks > size_t(h->cfg.max_batch)) return fail(THR_ERR_ARG, "n_blocks exceeds max_batch");
    if (template_id < 0 || template_id >= h->cfg.n_templates) return fail(THR_ERR_ARG, "bad template_id");
    HIP_TRY(hipSetDevice(h->device));
    int rc = ensure_staging(h, format);
    if (rc != THR_OK) return rc;
    const size_t n = size_t(h->cfg.block_len);
    const size_t blk_bytes = n * (format == THR_IN_U8 ? 2 : 8);
    const size_t dump_bytes = n_blocks * n * sizeof(float2);
    float2 *d_x = nullptr, *d_c = nullptr;
    HIP_TRY(hipMalloc(&d_x, dump_bytes));
    if (hipMalloc(&d_c, dump_bytes) != hipSuccess) {
        (void)hipFree(d_x);
        return fail(THR_ERR_DEVICE, "hipMalloc failed");
    }
    rc = THR_OK;
    do {
        if (hipMemsetAsync(d_x, 0, dump_bytes, h->stream) != hipSuccess ||
            hipMemsetAsync(d_c, 0, dump_bytes, h->stream) != hipSuccess ||
            hipMemcpyAsync(h->d_in, samples, n_blocks * blk_bytes, hipMemcpyHostToDevice, h->stream) !=
                hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "staging failed");
            break;
        }
        if (carrier_offset) {
            if (!h->d_forced && hipMalloc(&h->d_forced, size_t(h->cfg.max_batch) * sizeof(double)) != hipSuccess) {
                rc = fail(THR_ERR_DEVICE, "hipMalloc failed");
                break;
            }
            if (hipMemcpyAsync(h->d_forced, carrier_offset, n_blocks * sizeof(double), hipMemcpyHostToDevice,
                               h->stream) != hipSuccess) {
                rc = fail(THR_ERR_DEVICE, "staging failed");
                break;
            }
            h->forced = h->d_forced;
        }
        rc = run_batch(h, h->d_in, format, nullptr, int(n_blocks), h->d_rec, nullptr, d_x, d_c,
                       template_id, false);
        h->forced = nullptr;
        if (rc != THR_OK) break;
        if (shifted_fft_out &&
            hipMemcpyAsync(shifted_fft_out, d_x, dump_bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "D2H copy failed");
            break;
        }
        if (corr_out &&
            hipMemcpyAsync(corr_out, d_c, dump_bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "D2H copy failed");
            break;
        }
        if (hipStreamSynchronize(h->stream) != hipSuccess) {
            rc = fail(THR_ERR_DEVICE, "sync failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    (void)hipFree(d_x);
    (void)hipFree(d_c);
    return rc;
} catch (...) {
    return thr::on_exception("thr_debug_stage_offsets");
}

}  // extern "C"
