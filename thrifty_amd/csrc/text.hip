// Host-only text routines of the C ABI: .card line framing and .toad line formatting.
#include "host_internal.hpp"

// ---------------------------------------------------------------------------------------------
// .toad text (DetectionResult.serialize, toads_data.py:47-61) for a batch of detected records.
// ---------------------------------------------------------------------------------------------
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


extern "C" {

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


}  // extern "C"
