// The engine handle: error state, per-handle constants, creation / destruction, queries.
#include "host_internal.hpp"

namespace thr {
namespace host {

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

}  // namespace host
}  // namespace thr

namespace thr {
// error reporting for the other translation units of the library (identify.hip)
int fail_msg(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    host::g_last_error = buf;
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

namespace thr {
namespace host {

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
    // --- pass-1 / pass-B twiddles W_16384^(k1 q) of k_correlate and of the
    //     short-block kernels as one L2-resident table in global memory
    h->dev.gtw = nullptr;
    if (h->small || h->fast || h->seg || h->sec4k) {
        std::vector<float2> g(2 * 16 * 1024);
        for (int k1 = 0; k1 < 16; ++k1)
            for (int q = 0; q < 1024; ++q) g[k1 * 1024 + q] = unit_root((long long)k1 * q, 16384);
        // the same table again in the PAIRS a thread of pass B reads (thread n3 of row k1 needs
        // W^(k1 (32 n2 + n3)) for n2 = 0 .. 31): [(k1 * 16 + j) * 32 + n3] = (n2 = 2 j, n2 = 2 j + 1) --
        // sixteen 16-byte loads instead of thirty-two 8-byte ones (passes_w8.hpp: inv_passB)
        for (int k1 = 0; k1 < 16; ++k1)
            for (int j = 0; j < 16; ++j)
                for (int n3 = 0; n3 < 32; ++n3)
                    for (int e = 0; e < 2; ++e)
                        g[16 * 1024 + ((k1 * 16 + j) * 32 + n3) * 2 + e] =
                            unit_root((long long)k1 * (32 * (2 * j + e) + n3), 16384);
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


}  // namespace host
}  // namespace thr

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
        // block_len 16384, short template(s), no stddev term: the correlate stage as 4096-sample
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


}  // extern "C"
