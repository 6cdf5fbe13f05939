// The double-buffered chunk pipeline of the host entry points: staging, run_batch*, record copies.
#include "host_internal.hpp"

namespace thr {
namespace host {

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
              float2* dump_corr, int dump_template, bool carrier_only, size_t stride) {
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


// Raw-stream framing on the device (block_data.py:70-98; fastcard raw_reader.c:15-46): block i
// is the 2N bytes that start 2 (N - H) i bytes into the stream, so consecutive blocks overlap
// by H samples and the history copy of the host-side readers disappears.
int stream_stride(thr_handle* h, size_t* stride) {
    const size_t s = size_t(h->cfg.block_len - h->cfg.history_len) * 2;
    if (s % 4 != 0)
        return fail(THR_ERR_ARG, "raw-stream framing needs an even block_len - history_len (got %d)",
                    h->cfg.block_len - h->cfg.history_len);
    *stride = s;
    return THR_OK;
}


// ---- one chunk of each host entry point: stage the inputs into pipe buffer b (copy stream), run
// the batch (main stream), start the records' way back.  Shared by the synchronous loops and by
// thr_submit*().  Every HIP failure comes back as a status: the callers all leave through
// pipe_finish() / pipe_abort(), which synchronise both streams and clear what is pending (a
// chunk must never stay pending with `pend_dst` pointing into a caller array that is gone).
int chunk_samples(thr_handle* h, int b, const void* src, int format, size_t blk_bytes, size_t stride,
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

int chunk_card(thr_handle* h, int b, const char* text, size_t text_len, const int64_t* payload_off,
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
int pipe_enter_sync(thr_handle* h, const char* who) {
    HIP_TRY(hipSetDevice(h->device));
    const int rc = ensure_pipe(h);
    if (rc != THR_OK) return rc;
    if (h->hp.async_open != 0)
        return fail(THR_ERR_STATE, "%s: %d submitted batch(es) not collected yet (thr_collect first)",
                    who, h->hp.async_open);
    return THR_OK;
}


}  // namespace host
}  // namespace thr

extern "C" {

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


}  // extern "C"
