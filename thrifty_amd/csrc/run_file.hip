// thr_run_card / thr_run_stream: the whole `thrifty detect rx.card -o rx.toad` loop in one call.
//
// The reference runs, per block, card_reader / block_reader -> Detector.detect -> `if detected:
// print(result.serialize())` (detect.py:197-223, block_data.py:70-131).  thrifty_amd.detect drove the
// batched form of that loop from Python through five ctypes calls per batch (thr_frame_card,
// thr_submit_card two ahead, thr_collect, thr_format_toad, file.write); at a million blocks a second
// that interpreter thread was the bound.  Here the same loop is C++ on top of the SAME public entry
// points -- nothing in this file reaches into a handle -- with the text on a thread of its own:
//
//   caller's thread : frame a batch (thr_frame_card) -> thr_submit_card / thr_submit_stream, up to
//                     THR_MAX_IN_FLIGHT open -> thr_collect the oldest -> queue its records
//   formatter thread: keep the detected records (THR_FLAG_CORR; a THR_FLAG_INDEX_ERROR record ends
//                     the run where the reference's loop raised) -> thr_format_toad -> write(fd)
//                     and / or append them to the caller's record array
//
// A handle stays single-threaded (only the caller's thread touches it); the formatter uses the
// handle-free thr_format_toad.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <time.h>
#include <unistd.h>

#include "detect_common.hpp"

namespace thr {
int fail_msg(int code, const char* fmt, ...);
int on_exception(const char* who) noexcept;
}

namespace {

using Clock = std::chrono::steady_clock;
inline double secs(Clock::time_point a, Clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
}

double wall_clock() {
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

struct Batch {
    std::vector<double> ts;           // per block
    std::vector<int64_t> idx, off;    // per block (off: .card payload offsets)
    std::vector<thr_record> recs;     // [nb][T], filled by thr_collect
    size_t nb = 0;
    uint64_t ticket = 0;
    size_t first = 0;                 // ordinal of the batch's first block in this run
};

struct Runner {
    thr_handle* h;
    thr_run_opts o;
    thr_run_stats st{};
    int T = 1, block_len = 0, history_len = 0;
    int64_t new_len = 0;
    size_t max_batch = 0;
    // ring of batches: free -> in flight (caller's thread) -> queued (formatter) -> free
    std::vector<Batch> ring;
    std::deque<int> free_slots, queued;
    std::mutex mu;
    std::condition_variable cv;
    bool producer_done = false;
    std::atomic<bool> stop{false};    // the formatter hit the end of the run (index error / write error)
    int fmt_rc = THR_OK;
    std::string fmt_err;
    size_t rec_n = 0;
    std::thread formatter;

    // the end of the input for the formatter: it drains its queue and returns
    void finish() {
        if (!formatter.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            producer_done = true;
        }
        cv.notify_all();
        formatter.join();
    }
    ~Runner() { finish(); }

    int take_free() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !free_slots.empty(); });
        const int s = free_slots.front();
        free_slots.pop_front();
        return s;
    }
    void give_free(int s) {
        {
            std::lock_guard<std::mutex> lk(mu);
            free_slots.push_back(s);
        }
        cv.notify_all();
    }
    void enqueue(int s) {
        {
            std::lock_guard<std::mutex> lk(mu);
            queued.push_back(s);
        }
        cv.notify_all();
    }

    int write_all(const char* p, size_t n) {
        while (n) {
            const ssize_t w = ::write(o.out_fd, p, n);
            if (w < 0) {
                if (errno == EINTR) continue;
                fmt_err = std::string("write() to the .toad output failed: ") + strerror(errno);
                return THR_ERR_STATE;
            }
            p += w;
            n -= size_t(w);
        }
        return THR_OK;
    }

    void format_loop() {
        std::vector<thr_record> keep;
        std::vector<double> keep_ts;
        std::vector<char> text;
        for (;;) {
            int s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !queued.empty() || producer_done; });
                if (queued.empty()) return;
                s = queued.front();
                queued.pop_front();
            }
            Batch& b = ring[size_t(s)];
            // (this is a std::thread: an exception that left it -- std::bad_alloc from the three vectors
            // below -- would end the whole process in std::terminate.  It ends the RUN instead; the slot
            // goes back either way, so the caller's thread never blocks in take_free())
            if (fmt_rc == THR_OK && !stop.load()) try {
                const auto t0 = Clock::now();
                keep.clear();
                keep_ts.clear();
                const size_t n = b.nb * size_t(T);
                size_t end = n;
                for (size_t i = 0; i < n; ++i) {
                    const thr_record& r = b.recs[i];
                    if (r.flags & THR_FLAG_INDEX_ERROR) {   // carrier_sync.py:187: the reference's loop dies here
                        end = i;
                        st.index_error_block = r.block_idx;
                        st.index_error_bin = r.carrier_bin;
                        st.index_error_at = uint64_t(b.first + i / size_t(T));
                        break;
                    }
                    if (r.flags & THR_FLAG_CORR) {
                        keep.push_back(r);
                        keep_ts.push_back(b.ts[i / size_t(T)]);
                    }
                }
                if (!keep.empty()) {
                    if (o.out_fd >= 0) {
                        text.resize(keep.size() * size_t(THR_TOAD_LINE_MAX));
                        size_t used = 0;
                        const int rc = thr_format_toad(keep.data(), keep_ts.data(), keep.size(), new_len,
                                                       o.with_rxid, o.rxid, o.with_txid, o.carrier_offset_mode,
                                                       text.data(), text.size(), &used);
                        if (rc != THR_OK) {
                            fmt_rc = rc;
                            fmt_err = thr_last_error();     // (this thread's message)
                        } else {
                            const auto t1 = Clock::now();
                            st.format_s += secs(t0, t1);
                            const int wrc = write_all(text.data(), used);
                            st.write_s += secs(t1, Clock::now());
                            if (wrc != THR_OK) fmt_rc = wrc;
                            st.text_bytes += used;
                        }
                    }
                    if (fmt_rc == THR_OK && o.rec_out) {
                        if (rec_n + keep.size() > o.rec_capacity) {
                            fmt_rc = THR_ERR_ARG;
                            fmt_err = "thr_run: more detections than rec_capacity";
                        } else {
                            for (size_t i = 0; i < keep.size(); ++i) {   // the timestamp travels in `reserved`
                                thr_record r = keep[i];
                                std::memcpy(&r.reserved, &keep_ts[i], sizeof(double));
                                o.rec_out[rec_n + i] = r;
                            }
                            rec_n += keep.size();
                        }
                    }
                    st.detections += keep.size();
                }
                if (end != n || fmt_rc != THR_OK) stop.store(true);
            } catch (const std::exception& e) {
                fmt_rc = THR_ERR_STATE;
                fmt_err = std::string("thr_run: the text thread failed: ") + e.what();
                stop.store(true);
            } catch (...) {
                fmt_rc = THR_ERR_STATE;
                fmt_err = "thr_run: the text thread failed (unknown exception)";
                stop.store(true);
            }
            give_free(s);
        }
    }
};

// common driver: `next(batch)` frames the next batch (nb = 0 at the end of the input) and `submit`
// hands it to the engine
template <class Next, class Submit>
int drive(Runner& R, Next&& next, Submit&& submit) {
    const auto t_start = Clock::now();
    const int n_ring = THR_MAX_IN_FLIGHT + 3;
    R.ring.resize(size_t(n_ring));
    for (int i = 0; i < n_ring; ++i) R.free_slots.push_back(i);
    R.formatter = std::thread([&R] { R.format_loop(); });
    std::deque<int> flight;
    bool input_done = false;
    size_t ordinal = 0;
    // the first error of THIS thread, by where it lies in the input: a batch that fails at collect
    // (invalid base64) was submitted before whatever stopped the framing / submitting
    int in_rc = THR_OK, col_rc = THR_OK;
    std::string in_err, col_err;
    bool dead = false;        // a collect failed: what was submitted after it is waited for and dropped
    try {
    while (true) {
        if (!input_done && !dead && !R.stop.load() && flight.size() < size_t(THR_MAX_IN_FLIGHT)) {
            const int s = R.take_free();
            Batch& b = R.ring[size_t(s)];
            auto t0 = Clock::now();
            b.nb = 0;
            b.ticket = 0;
            int frc = next(b);
            auto t1 = Clock::now();
            R.st.frame_s += secs(t0, t1);
            if (frc == THR_OK && b.nb != 0) {
                b.first = ordinal;
                b.recs.resize(b.nb * size_t(R.T));
                frc = submit(b);
                R.st.submit_s += secs(t1, Clock::now());
            } else if (frc == THR_OK) {
                input_done = true;
                R.give_free(s);
                continue;
            }
            if (frc != THR_OK) {
                in_rc = frc;
                in_err = thr_last_error();
                input_done = true;
                R.give_free(s);
                continue;
            }
            ordinal += b.nb;
            R.st.batches += 1;
            flight.push_back(s);
            continue;
        }
        if (flight.empty()) break;
        const int s = flight.front();
        flight.pop_front();
        Batch& b = R.ring[size_t(s)];
        const auto t0 = Clock::now();
        const int crc = thr_collect(R.h, b.ticket);
        R.st.wait_s += secs(t0, Clock::now());
        if (crc != THR_OK && !dead) {
            col_rc = crc;
            col_err = thr_last_error();
            dead = true;
        }
        if (dead) {
            R.give_free(s);
            continue;
        }
        // (after a framing / submit error the batches submitted BEFORE it still go out: the
        // reference's per-line loop had emitted everything ahead of the bad input)
        R.st.blocks += b.nb;
        R.enqueue(s);
    }
    } catch (...) {
        // (host memory): the handle must not be left with open tickets, nor the formatter running
        in_rc = thr::on_exception("thr_run");
        in_err = thr_last_error();
        for (int s : flight) (void)thr_collect(R.h, R.ring[size_t(s)].ticket);
    }
    R.finish();
    R.st.total_s = secs(t_start, Clock::now());
    if (R.fmt_rc != THR_OK) return thr::fail_msg(R.fmt_rc, "%s", R.fmt_err.c_str());
    if (R.st.index_error_at != UINT64_MAX) {
        return thr::fail_msg(THR_ERR_INDEX,
                             "block %lld: carrier bin %d + fit reach >= block_len -- the reference raises "
                             "IndexError here (carrier_sync.py:187); detections before it were written",
                             (long long)R.st.index_error_block, R.st.index_error_bin);
    }
    if (col_rc != THR_OK) return thr::fail_msg(col_rc, "%s", col_err.c_str());
    if (in_rc != THR_OK) return thr::fail_msg(in_rc, "%s", in_err.c_str());
    return THR_OK;
}

int check(const char* who, thr_handle* h, const thr_run_opts* o, thr_run_stats* st, Runner& R) {
    if (!h || !o || !st) return thr::fail_msg(THR_ERR_ARG, "%s: null argument", who);
    if (o->struct_bytes != sizeof(thr_run_opts))
        return thr::fail_msg(THR_ERR_ARG, "%s: thr_run_opts.struct_bytes %u, this library's is %zu", who,
                             o->struct_bytes, sizeof(thr_run_opts));
    if (o->out_fd < 0 && !o->rec_out)
        return thr::fail_msg(THR_ERR_ARG, "%s: neither an output descriptor nor a record array", who);
    thr_settings cfg;
    const int rc = thr_get_settings(h, &cfg);
    if (rc != THR_OK) return rc;
    if (o->batch_blocks < 0 || o->batch_blocks > cfg.max_batch)
        return thr::fail_msg(THR_ERR_ARG, "%s: batch_blocks %d exceeds the handle's max_batch %d", who,
                             o->batch_blocks, cfg.max_batch);
    std::memset(st, 0, sizeof *st);
    st->index_error_at = UINT64_MAX;
    st->index_error_block = -1;
    R.h = h;
    R.o = *o;
    R.T = cfg.n_templates;
    R.max_batch = size_t(o->batch_blocks ? o->batch_blocks : cfg.max_batch);
    R.block_len = cfg.block_len;
    R.history_len = cfg.history_len;
    R.new_len = int64_t(cfg.block_len) - cfg.history_len;
    R.st = *st;
    return THR_OK;
}

}  // namespace

extern "C" {

int thr_run_card(thr_handle* h, const char* text, size_t text_len, const thr_run_opts* opts,
                 thr_run_stats* stats) try {
    Runner R;
    int rc = check("thr_run_card", h, opts, stats, R);
    if (rc != THR_OK) return rc;
    if (!text && text_len) return thr::fail_msg(THR_ERR_ARG, "thr_run_card: null text");
    size_t pos = 0;
    auto next = [&](Batch& b) -> int {
        b.ts.resize(R.max_batch);
        b.idx.resize(R.max_batch);
        b.off.resize(R.max_batch);
        while (pos < text_len) {
            size_t n = 0, used = 0;
            const int frc = thr_frame_card(text + pos, text_len - pos, R.block_len, 1, R.max_batch, b.ts.data(),
                                           b.idx.data(), b.off.data(), &n, &used);
            if (frc != THR_OK) return frc;
            for (size_t i = 0; i < n; ++i) b.off[i] += int64_t(pos);
            pos += used;
            if (n) {
                b.nb = n;
                return THR_OK;
            }
            if (used == 0) break;      // (nothing framed, nothing skipped: the end)
        }
        b.nb = 0;
        return THR_OK;
    };
    auto submit = [&](Batch& b) -> int {
        return thr_submit_card(h, text, text_len, b.off.data(), b.idx.data(), b.nb, b.recs.data(), &b.ticket);
    };
    rc = drive(R, next, submit);
    R.st.bytes_in = pos;
    *stats = R.st;
    return rc;
} catch (...) {
    return thr::on_exception("thr_run_card");
}

int thr_run_stream(thr_handle* h, const uint8_t* stream, size_t n_bytes, int64_t first_block_idx,
                   const thr_run_opts* opts, thr_run_stats* stats) try {
    Runner R;
    int rc = check("thr_run_stream", h, opts, stats, R);
    if (rc != THR_OK) return rc;
    if (!stream && n_bytes) return thr::fail_msg(THR_ERR_ARG, "thr_run_stream: null stream");
    const size_t blk = size_t(R.block_len) * 2, stride = size_t(R.new_len) * 2;
    const size_t total = n_bytes < blk ? 0 : (n_bytes - blk) / stride + 1;
    size_t done = 0;
    auto next = [&](Batch& b) -> int {
        b.nb = std::min(R.max_batch, total - done);
        if (b.nb == 0) return THR_OK;
        // the reference stamps a block when its read returns (block_data.py:86-98); of a mapped file
        // every block of a batch is "read" at once
        const double now = std::isnan(R.o.timestamp) ? wall_clock() : R.o.timestamp;
        b.ts.assign(b.nb, now);
        b.idx.resize(1);
        b.idx[0] = int64_t(done);      // (the batch's first block, relative to the stream)
        done += b.nb;
        return THR_OK;
    };
    auto submit = [&](Batch& b) -> int {
        const size_t at = size_t(b.idx[0]);
        size_t got = 0;
        const int src = thr_submit_stream(h, stream + at * stride, (b.nb - 1) * stride + blk,
                                          first_block_idx + int64_t(at), b.recs.data(), b.nb, &got, &b.ticket);
        if (src == THR_OK && got != b.nb)
            return thr::fail_msg(THR_ERR_STATE, "thr_run_stream: framed %zu blocks, engine took %zu", b.nb, got);
        return src;
    };
    rc = drive(R, next, submit);
    R.st.bytes_in = done ? (done - 1) * stride + blk : 0;
    *stats = R.st;
    return rc;
} catch (...) {
    return thr::on_exception("thr_run_stream");
}

}  // extern "C"
