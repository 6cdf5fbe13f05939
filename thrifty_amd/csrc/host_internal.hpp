// Internal header of the host side of libthriftyhip.so (include/thrifty_hip.h is the public one):
// the engine handle, its input window and pipeline slots, error reporting, and the functions the
// host translation units share.
//   handle.hip    error state, constants (twiddles, template spectra, section plans), thr_create* /
//                 thr_destroy, settings / path / profile queries
//   window.hip    thr_input_window* (the page-locked input file), thr_host_register
//   pipeline.hip  the double-buffered chunk pipeline (H2D staging, run_batch*, record copies)
//   entry.hip     thr_detect* / thr_submit* / thr_collect and the debug entry points
//   text.hip      host-only text routines: thr_frame_card, thr_format_toad
// (run_file.hip, identify.hip and card_ingest.hip never look inside a handle.)
#pragma once
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

namespace thr {
int fail_msg(int code, const char* fmt, ...);     // error text for thr_last_error(); returns `code`
int on_exception(const char* who) noexcept;      // the catch (...) of every entry point
namespace host {
int fail(int code, const char* fmt, ...);         // the same, for the host translation units

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(THR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,                \
                        hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
};


}  // namespace host
}  // namespace thr
using namespace thr::host;

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


namespace thr {
namespace host {

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


// ---- handle.hip
void host_fft(std::vector<std::complex<double>>& a);
float2 unit_root(long long num, long long den);
int window_indices(int start, int stop, int n, int* lo, int* count);
bool plan_sections(thr::DevCfg& d, int template_len);
bool plan_sections_4k(thr::DevCfg& d, int template_len);
int build_constants(thr_handle* h);
int build_preshift_bank(thr_handle* h);
// ---- pipeline.hip
int ensure_pipe(thr_handle* h);
int pipe_h2d(thr_handle* h, int b, void* d_dst, const void* src, size_t bytes);
void pipe_inputs_done(thr_handle* h, int b);
size_t pipe_chunk_blocks(const thr_handle* h, size_t bytes_per_block);
int pipe_grow(void** buf, size_t* have, size_t need);
int wait_event(thr_handle* h, hipEvent_t ev);
int pipe_drain(thr_handle* h, int b);
int pipe_inputs_enqueued(thr_handle* h, int b);
int pipe_records_enqueued(thr_handle* h, int b, thr_record* dst, size_t n_rec, size_t first, bool card);
int pipe_finish(thr_handle* h, int rc);
int ensure_staging(thr_handle* h, int format);
int run_batch(thr_handle* h, const void* d_samples, int format, const long long* d_block_idx,
              int n_blocks, thr_record* d_out, float2* dump_fft, float2* dump_xhat,
              float2* dump_corr, int dump_template, bool carrier_only, size_t stride = 0);
int stream_stride(thr_handle* h, size_t* stride);
int chunk_samples(thr_handle* h, int b, const void* src, int format, size_t blk_bytes, size_t stride,
                  const int64_t* block_idx, int64_t first_idx, size_t nb, thr_record* dst, size_t first);
int chunk_card(thr_handle* h, int b, const char* text, size_t text_len, const int64_t* payload_off,
               const int64_t* block_idx, size_t first, size_t nb, thr_record* dst);
int pipe_enter_sync(thr_handle* h, const char* who);

}  // namespace host
}  // namespace thr
