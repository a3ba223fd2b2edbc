// libspkio - text-ark embedding writer (the step after the hot path).
//
// Reference behaviour being replaced (scripts/decode.py:199-206): per utterance,
//     f.write(utt + ' [ ' + ' '.join(map(str, pred[i])) + ' ]\n')        with pred[i] a row of np.float32,
// i.e. 256 calls of numpy's float32 __str__ per utterance (~0.27 ms per utterance of interpreter time: 3.7 k utt/s per
// core, below what one GPU extracts).  Here a batch of rows is formatted by a few threads into one byte buffer that
// the caller writes with a single f.write().
//
// The output is byte-identical to numpy's str(np.float32(x)): the shortest digit string that round-trips in float32
// (std::to_chars), laid out positionally with at least one fractional digit when 1e-4 <= |x| < 1e16 and as
// d[.ddd]e[+-]XX otherwise; 'nan', 'inf', '-inf' as numpy prints them.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <charconv>
#include <thread>
#include <vector>

// worst case per value: sign + 16 integer digits + '.' + fraction (positional of 1e-4 needs 4 zeros + 9 digits) < 32 bytes
static const int64_t kMaxPerValue = 32;

static char* format_f32(char* p, float x) {
    if (isnan(x)) {
        memcpy(p, "nan", 3);
        return p + 3;
    }
    if (isinf(x)) {
        if (x < 0) *p++ = '-';
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (x == 0.0f) {
        if (signbit(x)) *p++ = '-';
        memcpy(p, "0.0", 3);
        return p + 3;
    }
    char sci[32];
    auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
    const double ax = fabs((double)x);   // numpy compares the promoted value with 1e-4 / 1e16 (float32(1e-4) < 1e-4)
    if (ax < 1e-4 || ax >= 1e16) {
        const size_t n = (size_t)(r.ptr - sci);
        memcpy(p, sci, n);
        return p + n;
    }
    // split into digits and decimal exponent
    const char* s = sci;
    if (*s == '-') {
        *p++ = '-';
        ++s;
    }
    char digits[16];
    int nd = 0;
    while (*s != 'e') {
        if (*s != '.') digits[nd++] = *s;
        ++s;
    }
    ++s;
    const bool neg = (*s == '-');
    ++s;
    int e = 0;
    while (s < r.ptr) e = e * 10 + (*s++ - '0');
    if (neg) e = -e;
    if (e >= 0) {
        for (int i = 0; i <= e; ++i) *p++ = i < nd ? digits[i] : '0';
        *p++ = '.';
        if (nd > e + 1) {
            for (int i = e + 1; i < nd; ++i) *p++ = digits[i];
        } else {
            *p++ = '0';
        }
    } else {
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -e - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = digits[i];
    }
    return p;
}

extern "C" int64_t spk_text_vectors_bound(int n, int D, const char* const* keys) {
    int64_t total = 0;
    for (int i = 0; i < n; ++i) total += (int64_t)strlen(keys[i]) + 6 + (int64_t)D * (kMaxPerValue + 1);
    return total;
}

// out <- "key [ v0 v1 ... ]\n" for each of the n rows of v[n][D]; returns the number of bytes written, or -1 when cap
// is smaller than spk_text_vectors_bound(n, D, keys).
extern "C" int64_t spk_format_text_vectors(int n, int D, const float* v, const char* const* keys, char* out, int64_t cap,
                                           int nthreads) {
    if (n <= 0) return 0;
    std::vector<int64_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (int64_t)strlen(keys[i]) + 6 + (int64_t)D * (kMaxPerValue + 1);
    if (cap < off[n]) return -1;
    std::vector<int64_t> len(n, 0);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = n;
    auto work = [&](int t) {
        for (int i = t; i < n; i += nthreads) {
            char* p = out + off[i];
            const size_t kl = strlen(keys[i]);
            memcpy(p, keys[i], kl);
            p += kl;
            memcpy(p, " [ ", 3);
            p += 3;
            const float* row = v + (size_t)i * D;
            for (int d = 0; d < D; ++d) {
                p = format_f32(p, row[d]);
                *p++ = ' ';
            }
            *p++ = ']';
            *p++ = '\n';
            len[i] = p - (out + off[i]);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    // compact the rows (each was formatted at its worst-case offset)
    int64_t w = len[0];
    for (int i = 1; i < n; ++i) {
        memmove(out + w, out + off[i], (size_t)len[i]);
        w += len[i];
    }
    return w;
}

// ---- vector-ark reader (the scoring back end's input: scripts/compute_mean.py:9-33, scripts/cosine_score.py:52-60 of the reference
// read the embedding file back through kaldi_io.read_vec_flt_ark, one Python-level parse per utterance - seconds per 100 k
// utterances before a single score is computed).  Here the whole ark - text 'key [ v0 ... ]' lines as decode writes them, or binary
// FV / DV records - is parsed natively into one [n][D] float64 matrix (text values are parsed as DOUBLES, as numpy does: the
// reference subtracts the mean in float64 before casting to float32) and a NUL-separated key buffer.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" void spk_io_set_error(const char* msg);

struct VecRec {
    int64_t key_off, key_len, data_off;   // data_off: text: first byte after '['; binary: first payload byte
    int32_t dim;                          // binary: from the header; text: -1 until parsed
    char kind;                            // 't' text, 'f' float32, 'd' float64
};

static bool parse_text_values(const char* p, const char* end, double* out, int D, int* got) {
    int n = 0;
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t')) ++p;
        if (p >= end || *p == ']' || *p == '\n') break;
        double v;
        auto r = std::from_chars(p, end, v);
        if (r.ec != std::errc()) {
            // from_chars does not take a leading '+'; numpy never prints one.  Anything else is a format error
            return false;
        }
        if (n < D) out[n] = v;
        ++n;
        p = r.ptr;
    }
    *got = n;
    return true;
}

extern "C" int spk_vec_ark_load(const char* path, int nthreads, int64_t* n_out, int32_t* D_out, double** data_out, char** keys_out,
                                int64_t* keys_bytes_out) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        spk_io_set_error("spk_vec_ark_load: cannot open file");
        return -1;
    }
    struct stat st;
    fstat(fd, &st);
    const int64_t size = st.st_size;
    if (size == 0) {
        close(fd);
        *n_out = 0; *D_out = 0; *data_out = nullptr; *keys_out = nullptr; *keys_bytes_out = 0;
        return 0;
    }
    const char* base = (const char*)mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) {
        spk_io_set_error("spk_vec_ark_load: mmap failed");
        return -1;
    }
    std::vector<VecRec> recs;
    int64_t pos = 0, key_bytes = 0;
    bool bad = false;
    while (pos < size) {
        // key: up to the first space (kaldi_io.read_key)
        int64_t k0 = pos;
        while (pos < size && base[pos] != ' ' && base[pos] != '\n') ++pos;
        if (pos >= size) break;
        if (base[pos] == '\n') { ++pos; continue; }      // blank line
        VecRec r;
        r.key_off = k0; r.key_len = pos - k0;
        ++pos;
        if (pos + 1 < size && base[pos] == '\0' && base[pos + 1] == 'B') {
            pos += 2;
            if (pos + 3 > size) { bad = true; break; }
            const bool f32 = memcmp(base + pos, "FV ", 3) == 0, f64 = memcmp(base + pos, "DV ", 3) == 0;
            if (!f32 && !f64) { bad = true; break; }
            pos += 3;
            if (pos + 5 > size || base[pos] != 4) { bad = true; break; }
            int32_t dim;
            memcpy(&dim, base + pos + 1, 4);
            pos += 5;
            r.kind = f32 ? 'f' : 'd';
            r.dim = dim;
            r.data_off = pos;
            pos += (int64_t)dim * (f32 ? 4 : 8);
            if (pos > size || dim < 0) { bad = true; break; }
        } else {
            while (pos < size && base[pos] == ' ') ++pos;
            if (pos >= size || base[pos] != '[') { bad = true; break; }
            r.kind = 't';
            r.dim = -1;
            r.data_off = pos + 1;
            while (pos < size && base[pos] != '\n') ++pos;
            ++pos;
        }
        key_bytes += r.key_len + 1;
        recs.push_back(r);
    }
    if (bad || recs.empty()) {
        munmap((void*)base, (size_t)size);
        spk_io_set_error("spk_vec_ark_load: not a vector ark ('key [ v ... ]' lines or binary FV / DV records)");
        return -2;
    }
    // dimension: from the first record
    int D = recs[0].dim;
    if (recs[0].kind == 't') {
        const char* p = base + recs[0].data_off;
        const char* e = p;
        while (e < base + size && *e != '\n') ++e;
        int got = 0;
        if (!parse_text_values(p, e, nullptr, 0, &got)) { munmap((void*)base, (size_t)size); spk_io_set_error("spk_vec_ark_load: bad number"); return -2; }
        D = got;
    }
    const int64_t n = (int64_t)recs.size();
    double* data = (double*)malloc((size_t)(n * (int64_t)D > 0 ? n * (int64_t)D : 1) * sizeof(double));
    char* keys = (char*)malloc((size_t)key_bytes);
    if (!data || !keys) {
        free(data); free(keys);
        munmap((void*)base, (size_t)size);
        spk_io_set_error("spk_vec_ark_load: out of memory");
        return -3;
    }
    {
        char* kp = keys;
        for (const VecRec& r : recs) {
            memcpy(kp, base + r.key_off, (size_t)r.key_len);
            kp[r.key_len] = 0;
            kp += r.key_len + 1;
        }
    }
    std::vector<int> fail((size_t)(nthreads > 0 ? nthreads : 1), 0);
    auto work = [&](int t, int nt) {
        for (int64_t i = t; i < n; i += nt) {
            const VecRec& r = recs[(size_t)i];
            double* dst = data + i * (int64_t)D;
            if (r.kind == 't') {
                const char* p = base + r.data_off;
                const char* e = p;
                while (e < base + size && *e != '\n') ++e;
                int got = 0;
                if (!parse_text_values(p, e, dst, D, &got) || got != D) fail[(size_t)t] = 1;
            } else if (r.dim != D) {
                fail[(size_t)t] = 1;
            } else if (r.kind == 'f') {
                const char* p = base + r.data_off;
                for (int d = 0; d < D; ++d) {
                    float v;
                    memcpy(&v, p + 4 * d, 4);
                    dst[d] = (double)v;
                }
            } else {
                memcpy(dst, base + r.data_off, (size_t)D * 8);
            }
        }
    };
    const int nt = nthreads > 1 ? nthreads : 1;
    if (nt == 1) work(0, 1);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, t, nt);
        for (auto& x : th) x.join();
    }
    munmap((void*)base, (size_t)size);
    for (int f : fail)
        if (f) {
            free(data); free(keys);
            spk_io_set_error("spk_vec_ark_load: a vector has another dimension than the first one, or a value does not parse");
            return -2;
        }
    *n_out = n; *D_out = D; *data_out = data; *keys_out = keys; *keys_bytes_out = key_bytes;
    return 0;
}

extern "C" void spk_vec_ark_free(double* data, char* keys) {
    free(data);
    free(keys);
}
