// Host-side decoder of the reference's region-feature wire format (include/cpt_io.h).
// No device code here: it lives in libcpt_hip.so so that one library carries the whole boundary.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "cpt_hip.h"
#include "cpt_io.h"

namespace cpt { int abi_fail(int code, const char* fmt, ...); }   // cpt_abi.hip: sets this thread's cpt_last_error()
namespace cpt { size_t b64_avx2_blocks(const unsigned char* s, size_t blocks, unsigned char* out); int b64_cpu_has_avx2(); }      // b64_avx2.cpp

namespace {

int io_fail(int code, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return cpt::abi_fail(code, "%s", buf);
}

// four tables: the 6-bit value of a character already shifted to its place in the 24-bit group; 0xffffffff = invalid
struct Tables {
    uint32_t t[4][256];
    Tables() {
        const char* abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int k = 0; k < 4; ++k)
            for (int c = 0; c < 256; ++c) t[k][c] = 0xffffffffu;
        for (uint32_t v = 0; v < 64; ++v) {
            const unsigned char c = (unsigned char)abc[v];
            t[0][c] = v << 18; t[1][c] = v << 12; t[2][c] = v << 6; t[3][c] = v;
        }
    }
};
const Tables g_tab;

// round 5: 32 characters per iteration where the CPU has AVX2 (b64_avx2.cpp, its own -mavx2 translation unit); checked once per process.
// CPT_B64_SCALAR=1 in the environment keeps the scalar loop (A/B measurements, tests of both paths).
const bool g_avx2 = [] {
    const char* e = getenv("CPT_B64_SCALAR");
    if (e && e[0] && e[0] != '0') return false;
    return cpt::b64_cpu_has_avx2() != 0;
}();

// decode exactly nbytes bytes from len base64 characters; 0 ok, else the offset+1 of the offending character / -1 size
long b64_decode(const unsigned char* s, size_t len, unsigned char* out, size_t nbytes) {
    if (len % 4 != 0) return -1;
    size_t pad = 0;
    if (len >= 1 && s[len - 1] == '=') pad = 1;
    if (len >= 2 && s[len - 2] == '=') pad = 2;
    if ((len / 4) * 3 - pad != nbytes) return -1;
    const size_t full = (pad ? len - 4 : len) / 4;
    const uint32_t(*t)[256] = g_tab.t;
    size_t o = 0, g0 = 0;
    if (g_avx2 && full >= 8) {
        // whole 32-character blocks of the unpadded part; a block with a character outside the alphabet stops the vector loop and the scalar
        // loop below finds and reports it (same results, same error positions as the scalar decoder alone)
        const size_t done = cpt::b64_avx2_blocks(s, full / 8, out);
        g0 = done * 8; s += done * 32; o = done * 24;
    }
    for (size_t g = g0; g < full; ++g, s += 4, o += 3) {
        const uint32_t v = t[0][s[0]] | t[1][s[1]] | t[2][s[2]] | t[3][s[3]];
        if (v & 0xff000000u) return (long)(g * 4) + 1;
        out[o] = (unsigned char)(v >> 16); out[o + 1] = (unsigned char)(v >> 8); out[o + 2] = (unsigned char)v;
    }
    if (pad) {
        const uint32_t v = t[0][s[0]] | t[1][s[1]] | (pad == 1 ? t[2][s[2]] : 0u);
        if (v & 0xff000000u) return (long)(full * 4) + 1;
        out[o] = (unsigned char)(v >> 16);
        if (pad == 1) out[o + 1] = (unsigned char)(v >> 8);
    }
    return 0;
}

int decode_one(const char* b64, size_t len, float* out, int dim) {
    const long r = b64_decode(reinterpret_cast<const unsigned char*>(b64), len, reinterpret_cast<unsigned char*>(out), (size_t)dim * 4);
    if (r == -1) return io_fail(CPT_ERR_SHAPE, "base64 feature of %zu characters does not decode to %d float32 values", len, dim);
    if (r > 0) return io_fail(CPT_ERR_SHAPE, "base64 feature: character %ld is outside the alphabet", r - 1);
    return CPT_OK;
}

int decode_seq(const char* const* b64, const size_t* lens, int n, int dim, int max_regions, float* out, int64_t* mask) {
    if (n < 0 || n > max_regions) return io_fail(CPT_ERR_SHAPE, "%d regions do not fit max_regions %d", n, max_regions);
    for (int i = 0; i < n; ++i) {
        if (!b64[i]) return io_fail(CPT_ERR_NULL, "region %d: null string", i);
        const int rc = decode_one(b64[i], lens[i], out + (size_t)i * dim, dim);
        if (rc != CPT_OK) return rc;
    }
    memset(out + (size_t)n * dim, 0, (size_t)(max_regions - n) * dim * sizeof(float));
    if (mask)
        for (int i = 0; i < max_regions; ++i) mask[i] = i < n ? 1 : 0;
    return CPT_OK;
}

// index of the quote that closes the JSON string opening at json[open] (len if unterminated): memchr for the next
// quote, then count the backslashes in front of it (an odd number escapes it)
size_t string_end(const char* json, size_t len, size_t open) {
    size_t p = open + 1;
    while (p < len) {
        const char* q = (const char*)memchr(json + p, '"', len - p);
        if (!q) return len;
        const size_t at = (size_t)(q - json);
        size_t bs = 0;
        while (at - 1 - bs > open && json[at - 1 - bs] == '\\') ++bs;
        if (bs % 2 == 0) return at;
        p = at + 1;
    }
    return len;
}

// One pass over a JSON text: string values of `key` located (offset / length between the quotes), optionally grouped
// by the array enclosing their objects, and the text copied to `stripped` with those values emptied.
struct Scan {
    size_t* offsets = nullptr; size_t* lens = nullptr; int* groups = nullptr;
    int max_values = 0, found = 0, n_groups = 0;
    size_t stripped_len = 0;
};

int scan_json(const char* json, size_t len, const char* key, char* stripped, size_t stripped_cap, Scan& sc) {
    const size_t klen = strlen(key);
    size_t i = 0, o = 0;
    // array nesting: children[d] = arrays opened so far as direct children of the open array at depth d
    int depth = 0, d_feat = -1;
    bool parent_open = false;
    std::vector<int> children(8, 0);
    auto put = [&](const char* p, size_t n) -> bool {
        if (o + n > stripped_cap) return false;
        memcpy(stripped + o, p, n);
        o += n;
        return true;
    };
    while (i < len) {
        if (json[i] != '"') {                    // outside strings: copy up to the next quote, tracking [ ]
            const char* q = (const char*)memchr(json + i, '"', len - i);
            const size_t n = q ? (size_t)(q - (json + i)) : len - i;
            for (size_t c = i; c < i + n; ++c) {
                if (json[c] == '[') {
                    ++depth;
                    if ((int)children.size() <= depth) children.resize(depth + 8, 0);
                    ++children[depth - 1];
                    children[depth] = 0;
                } else if (json[c] == ']') {
                    if (parent_open && depth == d_feat - 1) { sc.n_groups = children[depth]; parent_open = false; }
                    if (depth > 0) --depth;
                }
            }
            if (!put(json + i, n)) return io_fail(CPT_ERR_WORKSPACE, "stripped buffer too small");
            i += n;
            continue;
        }
        const size_t j = string_end(json, len, i);
        if (j >= len) return io_fail(CPT_ERR_SHAPE, "unterminated JSON string at byte %zu", i);
        const bool is_key = (j - i - 1 == klen) && memcmp(json + i + 1, key, klen) == 0;
        if (!put(json + i, j + 1 - i)) return io_fail(CPT_ERR_WORKSPACE, "stripped buffer too small");
        i = j + 1;
        if (!is_key) continue;
        size_t k = i;                            // a key is followed by  : "value"
        while (k < len && (json[k] == ' ' || json[k] == '\t')) ++k;
        if (k >= len || json[k] != ':') continue;
        ++k;
        while (k < len && (json[k] == ' ' || json[k] == '\t')) ++k;
        if (k >= len || json[k] != '"') continue;
        const size_t e = string_end(json, len, k);      // end of the value string
        if (e >= len) return io_fail(CPT_ERR_SHAPE, "unterminated \"%s\" value at byte %zu", key, k);
        if (sc.found >= sc.max_values) return io_fail(CPT_ERR_SHAPE, "more than %d \"%s\" values in the row", sc.max_values, key);
        if (sc.groups) {
            if (d_feat < 0) {
                if (depth < 1) return io_fail(CPT_ERR_SHAPE, "\"%s\" value outside any array", key);
                d_feat = depth;
                parent_open = true;
            }
            if (depth != d_feat || !parent_open) return io_fail(CPT_ERR_SHAPE, "\"%s\" values at different nesting depths", key);
            sc.groups[sc.found] = children[d_feat - 1] - 1;
        }
        sc.offsets[sc.found] = k + 1;
        sc.lens[sc.found] = e - (k + 1);
        ++sc.found;
        if (!put(json + i, k - i) || !put("\"\"", 2)) return io_fail(CPT_ERR_WORKSPACE, "stripped buffer too small");
        i = e + 1;
    }
    if (parent_open) sc.n_groups = children[d_feat - 1];
    sc.stripped_len = o;
    return CPT_OK;
}

}  // namespace

extern "C" {

int cpt_b64_decode_f32(const char* b64, size_t len, float* out, int dim) {
    if (!b64 || !out) return io_fail(CPT_ERR_NULL, "cpt_b64_decode_f32: null argument");
    if (dim <= 0) return io_fail(CPT_ERR_SHAPE, "cpt_b64_decode_f32: dim %d", dim);
    return decode_one(b64, len, out, dim);
}

int cpt_decode_regions(const char* const* b64, const size_t* lens, int n_regions, int dim, int max_regions,
                       float* out, int64_t* mask_img) {
    if (!out || (n_regions > 0 && (!b64 || !lens))) return io_fail(CPT_ERR_NULL, "cpt_decode_regions: null argument");
    if (dim <= 0 || max_regions <= 0) return io_fail(CPT_ERR_SHAPE, "cpt_decode_regions: dim %d max_regions %d", dim, max_regions);
    return decode_seq(b64, lens, n_regions, dim, max_regions, out, mask_img);
}

int cpt_decode_regions_batch(const char* base, const size_t* offsets, const size_t* lens, const int* first,
                             const int* n_regions, int n_seq, int dim, int max_regions, float* out,
                             int64_t* mask_img, int n_threads) {
    if (!out || !first || !n_regions || !base || !offsets || !lens) return io_fail(CPT_ERR_NULL, "cpt_decode_regions_batch: null argument");
    if (n_seq < 0 || dim <= 0 || max_regions <= 0) return io_fail(CPT_ERR_SHAPE, "cpt_decode_regions_batch: bad sizes");
    const size_t seq_elems = (size_t)max_regions * dim;
    auto one = [&](int s) -> int {
        const int n = n_regions[s];
        if (n < 0 || n > max_regions) return io_fail(CPT_ERR_SHAPE, "sequence %d: %d regions do not fit max_regions %d", s, n, max_regions);
        float* o = out + s * seq_elems;
        for (int i = 0; i < n; ++i) {
            const int k = first[s] + i;
            const int rc = decode_one(base + offsets[k], lens[k], o + (size_t)i * dim, dim);
            if (rc != CPT_OK) return rc;
        }
        memset(o + (size_t)n * dim, 0, (size_t)(max_regions - n) * dim * sizeof(float));
        if (mask_img)
            for (int i = 0; i < max_regions; ++i) mask_img[(size_t)s * max_regions + i] = i < n ? 1 : 0;
        return CPT_OK;
    };
    auto work = [&](int s0, int s1, int* status) {
        for (int s = s0; s < s1 && *status == CPT_OK; ++s) *status = one(s);
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > n_seq ? (n_seq > 0 ? n_seq : 1) : n_threads);
    if (nt == 1) {
        int st = CPT_OK;
        work(0, n_seq, &st);
        return st;
    }
    std::vector<std::thread> th;
    std::vector<int> st(nt, CPT_OK);
    for (int t = 0; t < nt; ++t)
        th.emplace_back(work, (int)((long)n_seq * t / nt), (int)((long)n_seq * (t + 1) / nt), &st[t]);
    for (auto& x : th) x.join();
    for (int t = 0; t < nt; ++t)
        if (st[t] != CPT_OK) {      // the message went to the worker's thread-local slot: redo that chunk here to report it
            int again = CPT_OK;
            work((int)((long)n_seq * t / nt), (int)((long)n_seq * (t + 1) / nt), &again);
            return again != CPT_OK ? again : st[t];
        }
    return CPT_OK;
}

int cpt_json_find_strings(const char* json, size_t len, const char* key, size_t* offsets, size_t* lens,
                          int max_values, char* stripped, size_t stripped_cap, int* n_values, size_t* stripped_len) {
    if (!json || !key || !offsets || !lens || !stripped || !n_values || !stripped_len) return io_fail(CPT_ERR_NULL, "cpt_json_find_strings: null argument");
    if (max_values < 0) return io_fail(CPT_ERR_SHAPE, "cpt_json_find_strings: max_values %d", max_values);
    Scan sc;
    sc.offsets = offsets; sc.lens = lens; sc.max_values = max_values;
    const int rc = scan_json(json, len, key, stripped, stripped_cap, sc);
    if (rc != CPT_OK) return rc;
    *n_values = sc.found;
    *stripped_len = sc.stripped_len;
    return CPT_OK;
}

}  // extern "C"

namespace {
// cpt_decode_tsv_rows (out: decoded float32) and cpt_pack_tsv_rows (text: the located strings copied as they are, for the device decoder) share the row walk
int tsv_rows_impl(const char* who, const char* const* rows, const size_t* lens, int n_rows, const char* key, int dim,
                  int max_regions, int max_seqs, float* out, char* text, int64_t* mask_img, char* const* stripped,
                  const size_t* stripped_cap, size_t* stripped_len, int* seqs_per_row, int* regions_per_seq,
                  int n_threads) {
    if (!rows || !lens || !key || (!out && !text) || !stripped || !stripped_cap || !stripped_len || !seqs_per_row || !regions_per_seq)
        return io_fail(CPT_ERR_NULL, "%s: null argument", who);
    if (n_rows < 0 || dim <= 0 || max_regions <= 0 || max_seqs < 0) return io_fail(CPT_ERR_SHAPE, "%s: bad sizes", who);
    const size_t chars = cpt_b64_chars(dim);
    const size_t min_chars = ((size_t)dim * 16) / 3;       // a value is the base64 of dim float32
    struct Row { std::vector<size_t> off, len; std::vector<int> group; int groups = 0, status = CPT_OK; };
    std::vector<Row> R(n_rows);
    const int nt = n_threads < 1 ? 1 : n_threads;
    auto parallel = [&](int n, auto&& fn) {
        const int t_use = nt > n ? (n > 0 ? n : 1) : nt;
        if (t_use == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::vector<std::thread> th;
        std::atomic<int> next(0);
        for (int t = 0; t < t_use; ++t)
            th.emplace_back([&] { for (int i = next++; i < n; i = next++) fn(i); });
        for (auto& x : th) x.join();
    };
    // phase 1: scan every row (values located + grouped, JSON stripped)
    auto scan_row = [&](int r) {
        Row& w = R[r];
        const int cap = (int)(lens[r] / (min_chars ? min_chars : 1)) + 1;
        w.off.resize(cap); w.len.resize(cap); w.group.resize(cap);
        Scan sc;
        sc.offsets = w.off.data(); sc.lens = w.len.data(); sc.groups = w.group.data(); sc.max_values = cap;
        w.status = scan_json(rows[r], lens[r], key, stripped[r], stripped_cap[r], sc);
        if (w.status != CPT_OK) return;
        w.off.resize(sc.found); w.len.resize(sc.found); w.group.resize(sc.found);
        w.groups = sc.n_groups;
        stripped_len[r] = sc.stripped_len;
    };
    // Round 5: a row is DECODED by the thread that scanned it, right behind the scan (its 4.4 MB of text are still in that core's caches: the
    // two-phase form read all rows' text from memory twice).  Where a row's sequences land depends on the rows in front of it only through
    // their sequence COUNT, which every row publishes as soon as its scan is done; rows are handed out in order, so the rows a thread waits
    // for are always already being scanned by someone.
    const size_t seq_elems = (size_t)max_regions * dim;
    std::vector<std::atomic<int>> published(n_rows > 0 ? n_rows : 1);
    for (auto& p : published) p.store(-1, std::memory_order_relaxed);
    std::vector<int> dec_status(n_rows > 0 ? n_rows : 1, CPT_OK);
    std::atomic<int> overflow(0);
    auto decode_row = [&](int r, int base, bool report) -> int {          // sequences base .. base + groups of row r
        const Row& w = R[r];
        std::vector<int> first(w.groups, 0), count(w.groups, 0);
        for (size_t v = 0; v < w.group.size(); ++v) {
            const int g = w.group[v];
            if (count[g] == 0) first[g] = (int)v;
            ++count[g];
        }
        for (int g = 0; g < w.groups; ++g) {
            const int s = base + g;
            regions_per_seq[s] = count[g];
            if (count[g] > max_regions)
                return report ? io_fail(CPT_ERR_SHAPE, "sequence %d: %d regions do not fit max_regions %d", s, count[g], max_regions) : CPT_ERR_SHAPE;
            if (text) {
                // device decode: the strings travel as text, [max_seqs][max_regions][chars]; slots without a region are left alone (the mask tells)
                char* o = text + (size_t)s * max_regions * chars;
                for (int i = 0; i < count[g]; ++i) {
                    if (w.len[first[g] + i] != chars)
                        return report ? io_fail(CPT_ERR_SHAPE, "base64 feature of %zu characters does not decode to %d float32 values", w.len[first[g] + i], dim) : CPT_ERR_SHAPE;
                    memcpy(o + (size_t)i * chars, rows[r] + w.off[first[g] + i], chars);
                }
            } else {
            float* o = out + s * seq_elems;
            for (int i = 0; i < count[g]; ++i) {
                const int rc = decode_one(rows[r] + w.off[first[g] + i], w.len[first[g] + i], o + (size_t)i * dim, dim);
                if (rc != CPT_OK) return rc;
            }
            memset(o + (size_t)count[g] * dim, 0, (size_t)(max_regions - count[g]) * dim * sizeof(float));
            }
            if (mask_img)
                for (int i = 0; i < max_regions; ++i) mask_img[(size_t)s * max_regions + i] = i < count[g] ? 1 : 0;
        }
        return CPT_OK;
    };
    auto row_job = [&](int r) {
        scan_row(r);
        published[r].store(R[r].status == CPT_OK ? R[r].groups : 0, std::memory_order_release);
        if (R[r].status != CPT_OK) return;
        int base = 0;
        for (int q = 0; q < r; ++q) {
            int g;
            while ((g = published[q].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
            base += g;
        }
        if (base + R[r].groups > max_seqs) { overflow.store(1); return; }
        dec_status[r] = decode_row(r, base, false);
    };
    parallel(n_rows, row_job);
    for (int r = 0; r < n_rows; ++r)
        if (R[r].status != CPT_OK) { scan_row(r); return R[r].status; }     // redo in this thread to report its message
    int n_seq = 0;
    for (int r = 0; r < n_rows; ++r) { seqs_per_row[r] = R[r].groups; n_seq += R[r].groups; }
    if (n_seq > max_seqs || overflow.load()) return io_fail(CPT_ERR_SHAPE, "%d sequences in %d rows do not fit max_seqs %d", n_seq, n_rows, max_seqs);
    for (int r = 0, base = 0; r < n_rows; base += R[r].groups, ++r)
        if (dec_status[r] != CPT_OK) return decode_row(r, base, true);       // redo in this thread to report its message
    return CPT_OK;
}
}  // namespace

extern "C" {

int cpt_decode_tsv_rows(const char* const* rows, const size_t* lens, int n_rows, const char* key, int dim,
                        int max_regions, int max_seqs, float* out, int64_t* mask_img, char* const* stripped,
                        const size_t* stripped_cap, size_t* stripped_len, int* seqs_per_row, int* regions_per_seq,
                        int n_threads) {
    if (!out) return io_fail(CPT_ERR_NULL, "cpt_decode_tsv_rows: null argument");
    return tsv_rows_impl("cpt_decode_tsv_rows", rows, lens, n_rows, key, dim, max_regions, max_seqs, out, nullptr, mask_img, stripped, stripped_cap, stripped_len,
                         seqs_per_row, regions_per_seq, n_threads);
}

int cpt_pack_tsv_rows(const char* const* rows, const size_t* lens, int n_rows, const char* key, int dim,
                      int max_regions, int max_seqs, char* text, int64_t* mask_img, char* const* stripped,
                      const size_t* stripped_cap, size_t* stripped_len, int* seqs_per_row, int* regions_per_seq,
                      int n_threads) {
    if (!text || !mask_img) return io_fail(CPT_ERR_NULL, "cpt_pack_tsv_rows: null argument");
    return tsv_rows_impl("cpt_pack_tsv_rows", rows, lens, n_rows, key, dim, max_regions, max_seqs, nullptr, text, mask_img, stripped, stripped_cap, stripped_len,
                         seqs_per_row, regions_per_seq, n_threads);
}

}  // extern "C"
