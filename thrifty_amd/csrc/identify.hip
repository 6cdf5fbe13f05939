// identify on the device (reference thrifty/identify.py:26-181):
// transmitter classification from the carrier bin, then the duplicate filter
// (sort by (rxid, txid, block, timestamp), drop a detection whose sorted neighbour sits in
// the adjacent block with more energy, drop unidentified ones), then output order by
// timestamp.  Works on columns of detections; everything that scales with the number of
// detections runs on the GPU (histogram, classification, four stable radix-sort passes,
// neighbour test, compaction); the window edges of the auto mode come from a histogram of
// at most a few thousand bins and are found on the host.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/thrifty_hip.h"

namespace thr {
int fail_msg(int code, const char* fmt, ...);
int on_exception(const char* who) noexcept;  // handle.hip
}

namespace {

// order-preserving maps to unsigned keys
__device__ __forceinline__ unsigned key_i32(int v) { return unsigned(v) ^ 0x80000000u; }
__device__ __forceinline__ unsigned long long key_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ void k_iota(unsigned* idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = unsigned(i);
}
__global__ void k_keys_f64(const double* __restrict__ v, const unsigned* __restrict__ perm, int n,
                           unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = key_f64(v[perm[i]]);
}
__global__ void k_keys_i32(const int* __restrict__ v, const unsigned* __restrict__ perm, int n,
                           unsigned* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = key_i32(v[perm[i]]);
}

// identify.py:109-121 -- inclusive ranges on bin + offset, the LAST matching entry wins
__global__ void k_classify_map(const int* __restrict__ rxid, const int* __restrict__ cbin,
                               const double* __restrict__ coff, int n,
                               const thr_freq_range* __restrict__ map, int n_map,
                               int* __restrict__ txid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double freq = double(cbin[i]) + coff[i];
    int tx = -1;
    for (int m = 0; m < n_map; ++m)
        if (map[m].rxid == rxid[i] && freq >= map[m].lo && freq <= map[m].hi) tx = map[m].txid;
    txid[i] = tx;
}

// auto mode: per-RX histogram of carrier bins (np.bincount, identify.py:41)
__global__ void k_histogram(const int* __restrict__ rxid, const int* __restrict__ cbin, int n,
                            const int* __restrict__ rx_list, int n_rx, int bin_lo, int n_bins,
                            unsigned* __restrict__ hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int r = 0; r < n_rx; ++r)
        if (rx_list[r] == rxid[i]) atomicAdd(&hist[size_t(r) * n_bins + (cbin[i] - bin_lo)], 1u);
}
// np.digitize(bin, edges[:-1]) - 1 (identify.py:102-103): edges ascending
__global__ void k_digitize(const int* __restrict__ rxid, const int* __restrict__ cbin, int n,
                           const int* __restrict__ rx_list, int n_rx,
                           const double* __restrict__ edges, const int* __restrict__ edge_ptr,
                           int* __restrict__ txid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int tx = -1;
    for (int r = 0; r < n_rx; ++r) {
        if (rx_list[r] != rxid[i]) continue;
        const double x = double(cbin[i]);
        int cnt = 0;
        for (int e = edge_ptr[r]; e < edge_ptr[r + 1] - 1; ++e) cnt += (edges[e] <= x) ? 1 : 0;
        tx = cnt - 1;
    }
    txid[i] = tx;
}

// identify.py:153-165 on the sorted order `perm`; np.roll wraps at both ends
__global__ void k_dup_mask(const unsigned* __restrict__ perm, const int* __restrict__ block,
                           const double* __restrict__ energy, const int* __restrict__ txid, int n,
                           unsigned char* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned c = perm[i], p = perm[i == 0 ? n - 1 : i - 1], q = perm[i == n - 1 ? 0 : i + 1];
    const bool drop_prev = block[c] == block[p] + 1 && energy[c] < energy[p];
    const bool drop_next = block[c] == block[q] - 1 && energy[c] < energy[q];
    keep[c] = !(drop_prev || drop_next || txid[c] == -1);
}
// kept detections in timestamp order: `perm_ts` is the stable sort by timestamp
__global__ void k_flags_in_order(const unsigned* __restrict__ perm_ts,
                                 const unsigned char* __restrict__ keep, int n,
                                 unsigned char* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = keep[perm_ts[i]];
}
__global__ void k_widen(const unsigned* __restrict__ in, int n, long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (long long)in[i];
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

#define ID_TRY(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return thr::fail_msg(THR_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class K>
hipError_t sort_pass(DevBuf& tmp, size_t& tmp_bytes, K* keys_in, K* keys_out, unsigned* val_in,
                     unsigned* val_out, int n, hipStream_t s) {
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, val_in,
                                                      val_out, n, 0, int(sizeof(K) * 8), s);
    if (e != hipSuccess) return e;
    if (need > tmp_bytes) {
        if (tmp.p) (void)hipFree(tmp.p);
        tmp.p = nullptr;
        if ((e = tmp.alloc(need)) != hipSuccess) return e;
        tmp_bytes = need;
    }
    return hipcub::DeviceRadixSort::SortPairs(tmp.p, need, keys_in, keys_out, val_in, val_out, n, 0,
                                              int(sizeof(K) * 8), s);
}

// identify.py:26-77 on one RX's histogram (float64 like NumPy; np.std is the population std)
std::vector<double> window_edges(const unsigned* cnts, int n_bins, int first_bin) {
    double mean = 0;
    for (int i = 0; i < n_bins; ++i) mean += cnts[i];
    mean /= n_bins;
    double var = 0;
    for (int i = 0; i < n_bins; ++i) var += (cnts[i] - mean) * (cnts[i] - mean);
    const double sd = std::sqrt(var / n_bins), low = sd * 0.4, high = sd * 1.25;
    std::vector<std::pair<int, int>> peaks;
    bool below = true;
    int start = 0;
    for (int i = 0; i < n_bins; ++i) {
        if (!below && cnts[i] < low) {
            peaks.push_back({start, i});
            below = true;
        }
        if (below && cnts[i] > high) {
            start = i;
            below = false;
        }
    }
    if (!below) peaks.push_back({start, n_bins - 1});
    std::vector<double> edges{double(first_bin)};
    for (size_t i = 0; i + 1 < peaks.size(); ++i) {
        const int sum = peaks[i].second + peaks[i + 1].first;
        edges.push_back(double((sum >= 0 ? sum : sum - 1) / 2 + first_bin));  // floor division
    }
    edges.push_back(double(first_bin + n_bins));
    return edges;
}

}  // namespace

extern "C" int thr_identify(int device_id, size_t n_in, const int32_t* rxid, const int32_t* block,
                            const double* timestamp, const int32_t* carrier_bin,
                            const double* carrier_offset, const double* energy,
                            const thr_freq_range* map, size_t n_map, int32_t* txid_out,
                            uint8_t* keep_out, int64_t* kept_order_out, size_t* n_kept_out) try {
    if (n_kept_out) *n_kept_out = 0;
    if (n_in == 0) return THR_OK;
    if (!rxid || !block || !timestamp || !carrier_bin || !carrier_offset || !energy || !txid_out ||
        !keep_out || !kept_order_out || !n_kept_out)
        return thr::fail_msg(THR_ERR_ARG, "thr_identify: null argument");
    if (n_in > size_t(1) << 28) return thr::fail_msg(THR_ERR_ARG, "thr_identify: too many detections");
    if (n_map > 0 && !map) return thr::fail_msg(THR_ERR_ARG, "thr_identify: null frequency map");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return thr::fail_msg(THR_ERR_DEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return thr::fail_msg(THR_ERR_ARG, "bad device_id %d", device_id);
    ID_TRY(hipSetDevice(device_id));
    const int n = int(n_in);
    const dim3 blk(256), grid((n + 255) / 256);
    hipStream_t s = nullptr;

    DevBuf d_rx, d_blk, d_ts, d_bin, d_off, d_en, d_tx, d_keep, d_flag;
    ID_TRY(d_rx.alloc(size_t(n) * 4));
    ID_TRY(d_blk.alloc(size_t(n) * 4));
    ID_TRY(d_ts.alloc(size_t(n) * 8));
    ID_TRY(d_bin.alloc(size_t(n) * 4));
    ID_TRY(d_off.alloc(size_t(n) * 8));
    ID_TRY(d_en.alloc(size_t(n) * 8));
    ID_TRY(d_tx.alloc(size_t(n) * 4));
    ID_TRY(d_keep.alloc(n));
    ID_TRY(d_flag.alloc(n));
    ID_TRY(hipMemcpy(d_rx.p, rxid, size_t(n) * 4, hipMemcpyHostToDevice));
    ID_TRY(hipMemcpy(d_blk.p, block, size_t(n) * 4, hipMemcpyHostToDevice));
    ID_TRY(hipMemcpy(d_ts.p, timestamp, size_t(n) * 8, hipMemcpyHostToDevice));
    ID_TRY(hipMemcpy(d_bin.p, carrier_bin, size_t(n) * 4, hipMemcpyHostToDevice));
    ID_TRY(hipMemcpy(d_off.p, carrier_offset, size_t(n) * 8, hipMemcpyHostToDevice));
    ID_TRY(hipMemcpy(d_en.p, energy, size_t(n) * 8, hipMemcpyHostToDevice));

    // ---- 1. transmitter ids
    if (n_map > 0) {
        DevBuf d_map;
        ID_TRY(d_map.alloc(n_map * sizeof(thr_freq_range)));
        ID_TRY(hipMemcpy(d_map.p, map, n_map * sizeof(thr_freq_range), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_classify_map, grid, blk, 0, s, d_rx.as<int>(), d_bin.as<int>(),
                           d_off.as<double>(), n, d_map.as<thr_freq_range>(), int(n_map), d_tx.as<int>());
        ID_TRY(hipGetLastError());
        ID_TRY(hipDeviceSynchronize());  // d_map goes out of scope
    } else {
        // receivers and the overall bin range (host: two cheap passes over the caller's columns)
        std::vector<int> rx_list(rxid, rxid + n);
        std::sort(rx_list.begin(), rx_list.end());
        rx_list.erase(std::unique(rx_list.begin(), rx_list.end()), rx_list.end());
        const int n_rx = int(rx_list.size());
        const int bin_lo = *std::min_element(carrier_bin, carrier_bin + n);
        const int bin_hi = *std::max_element(carrier_bin, carrier_bin + n);
        const long long span = (long long)bin_hi - bin_lo + 1;
        if (span * n_rx > (1ll << 26))
            return thr::fail_msg(THR_ERR_ARG, "thr_identify: carrier bins span %lld x %d receivers", span, n_rx);
        const int n_bins = int(span);
        DevBuf d_rxl, d_hist, d_edges, d_eptr;
        ID_TRY(d_rxl.alloc(n_rx * 4));
        ID_TRY(d_hist.alloc(size_t(n_rx) * n_bins * 4));
        ID_TRY(hipMemcpy(d_rxl.p, rx_list.data(), n_rx * 4, hipMemcpyHostToDevice));
        ID_TRY(hipMemset(d_hist.p, 0, size_t(n_rx) * n_bins * 4));
        hipLaunchKernelGGL(k_histogram, grid, blk, 0, s, d_rx.as<int>(), d_bin.as<int>(), n,
                           d_rxl.as<int>(), n_rx, bin_lo, n_bins, d_hist.as<unsigned>());
        ID_TRY(hipGetLastError());
        std::vector<unsigned> hist(size_t(n_rx) * n_bins);
        ID_TRY(hipMemcpy(hist.data(), d_hist.p, hist.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> edges;
        std::vector<int> eptr{0};
        for (int r = 0; r < n_rx; ++r) {
            // np.bincount(freqs - first_bin): trim to this receiver's own first .. last bin
            const unsigned* h = hist.data() + size_t(r) * n_bins;
            int lo = 0, hi = n_bins - 1;
            while (h[lo] == 0) ++lo;
            while (h[hi] == 0) --hi;
            const std::vector<double> e = window_edges(h + lo, hi - lo + 1, bin_lo + lo);
            edges.insert(edges.end(), e.begin(), e.end());
            eptr.push_back(int(edges.size()));
        }
        ID_TRY(d_edges.alloc(edges.size() * 8));
        ID_TRY(d_eptr.alloc(eptr.size() * 4));
        ID_TRY(hipMemcpy(d_edges.p, edges.data(), edges.size() * 8, hipMemcpyHostToDevice));
        ID_TRY(hipMemcpy(d_eptr.p, eptr.data(), eptr.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_digitize, grid, blk, 0, s, d_rx.as<int>(), d_bin.as<int>(), n,
                           d_rxl.as<int>(), n_rx, d_edges.as<double>(), d_eptr.as<int>(), d_tx.as<int>());
        ID_TRY(hipGetLastError());
        ID_TRY(hipDeviceSynchronize());
    }

    // ---- 2. stable LSD sort: timestamp, then block, txid, rxid (least significant key first)
    DevBuf d_k64a, d_k64b, d_k32a, d_k32b, d_pa, d_pb, d_pts, d_tmp;
    size_t tmp_bytes = 0;
    ID_TRY(d_k64a.alloc(size_t(n) * 8));
    ID_TRY(d_k64b.alloc(size_t(n) * 8));
    ID_TRY(d_k32a.alloc(size_t(n) * 4));
    ID_TRY(d_k32b.alloc(size_t(n) * 4));
    ID_TRY(d_pa.alloc(size_t(n) * 4));
    ID_TRY(d_pb.alloc(size_t(n) * 4));
    ID_TRY(d_pts.alloc(size_t(n) * 4));
    unsigned *pa = d_pa.as<unsigned>(), *pb = d_pb.as<unsigned>();
    hipLaunchKernelGGL(k_iota, grid, blk, 0, s, pa, n);
    hipLaunchKernelGGL(k_keys_f64, grid, blk, 0, s, d_ts.as<double>(), pa, n, d_k64a.as<unsigned long long>());
    ID_TRY(sort_pass(d_tmp, tmp_bytes, d_k64a.as<unsigned long long>(), d_k64b.as<unsigned long long>(), pa, pb, n, s));
    ID_TRY(hipMemcpyAsync(d_pts.p, pb, size_t(n) * 4, hipMemcpyDeviceToDevice, s));  // order by timestamp alone
    std::swap(pa, pb);
    for (const int* col : {d_blk.as<int>(), d_tx.as<int>(), d_rx.as<int>()}) {
        hipLaunchKernelGGL(k_keys_i32, grid, blk, 0, s, col, pa, n, d_k32a.as<unsigned>());
        ID_TRY(sort_pass(d_tmp, tmp_bytes, d_k32a.as<unsigned>(), d_k32b.as<unsigned>(), pa, pb, n, s));
        std::swap(pa, pb);
    }

    // ---- 3. neighbour test in sorted order, 4. kept detections in timestamp order
    hipLaunchKernelGGL(k_dup_mask, grid, blk, 0, s, pa, d_blk.as<int>(), d_en.as<double>(),
                       d_tx.as<int>(), n, d_keep.as<unsigned char>());
    hipLaunchKernelGGL(k_flags_in_order, grid, blk, 0, s, d_pts.as<unsigned>(), d_keep.as<unsigned char>(),
                       n, d_flag.as<unsigned char>());
    ID_TRY(hipGetLastError());
    DevBuf d_sel, d_nsel, d_wide;
    ID_TRY(d_sel.alloc(size_t(n) * 4));
    ID_TRY(d_nsel.alloc(4));
    ID_TRY(d_wide.alloc(size_t(n) * 8));
    {
        size_t need = 0;
        ID_TRY(hipcub::DeviceSelect::Flagged(nullptr, need, d_pts.as<unsigned>(), d_flag.as<unsigned char>(),
                                             d_sel.as<unsigned>(), d_nsel.as<int>(), n, s));
        if (need > tmp_bytes) {
            if (d_tmp.p) (void)hipFree(d_tmp.p);
            d_tmp.p = nullptr;
            ID_TRY(d_tmp.alloc(need));
            tmp_bytes = need;
        }
        ID_TRY(hipcub::DeviceSelect::Flagged(d_tmp.p, need, d_pts.as<unsigned>(), d_flag.as<unsigned char>(),
                                             d_sel.as<unsigned>(), d_nsel.as<int>(), n, s));
    }
    hipLaunchKernelGGL(k_widen, grid, blk, 0, s, d_sel.as<unsigned>(), n, d_wide.as<long long>());
    ID_TRY(hipGetLastError());
    int n_kept = 0;
    ID_TRY(hipMemcpy(&n_kept, d_nsel.p, 4, hipMemcpyDeviceToHost));
    ID_TRY(hipMemcpy(txid_out, d_tx.p, size_t(n) * 4, hipMemcpyDeviceToHost));
    ID_TRY(hipMemcpy(keep_out, d_keep.p, n, hipMemcpyDeviceToHost));
    ID_TRY(hipMemcpy(kept_order_out, d_wide.p, size_t(n_kept) * 8, hipMemcpyDeviceToHost));
    *n_kept_out = size_t(n_kept);
    return THR_OK;
} catch (...) {
    return thr::on_exception("thr_identify");
}
