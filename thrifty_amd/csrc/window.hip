// thr_input_window*: the page-locked input file (InputWindow in host_internal.hpp), thr_host_register.
#include "host_internal.hpp"

extern "C" {

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


}  // extern "C"
