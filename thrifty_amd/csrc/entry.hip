// The data-path entry points: thr_detect* / thr_submit* / thr_collect, compaction, debug hooks.
#include "host_internal.hpp"

extern "C" {

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
