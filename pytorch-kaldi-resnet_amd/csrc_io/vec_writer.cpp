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
