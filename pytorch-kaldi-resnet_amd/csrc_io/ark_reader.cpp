// libspkio - native batch ingest for Kaldi 'FM ' matrices (the step before the hot path).
//
// Reference behaviour being replaced (scripts/datasets.py:59-72 + scripts/kaldi_io.py:41-71,376-410): per sample,
// open the ark, seek to the scp offset, read the WHOLE utterance, crop seq_len frames at a random start, transpose to
// [F, T].  Here a batch is read by a small thread pool with pread(): only the cropped rows are read, and they are
// transposed straight into the caller's (pinned) [B][F][T] staging buffer, so the DataLoader worker processes, the
// full-utterance read and the collate copy disappear.
//
// C ABI, host pointers only:
//   spk_ark_probe      - parse the matrix headers at (path, offset): rows / cols / payload offset (done once per scp)
//   spk_ark_read_crop  - fill out[b][f][t] = M_b[start_b + t][f] for t < T, all b in the batch
// Returns 0 on success, < 0 on error (message via spk_io_last_error).
#include <fcntl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";
static std::mutex g_err_mu;
static char g_err_shared[512] = "";

static void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(g_err_shared, g_err, sizeof(g_err));
}

extern "C" const char* spk_io_last_error(void) { return g_err_shared; }
extern "C" void spk_io_set_error(const char* msg) { set_err("%s", msg); }      // for the other translation units of libspkio

// file-descriptor cache: arks are few and large, reopen per sample is what the reference pays
static std::mutex g_fd_mu;
static std::map<std::string, int> g_fds;

static int get_fd(const char* path) {
    std::lock_guard<std::mutex> lk(g_fd_mu);
    auto it = g_fds.find(path);
    if (it != g_fds.end()) return it->second;
    int fd = open(path, O_RDONLY);
    if (fd >= 0) g_fds[path] = fd;
    return fd;
}

extern "C" void spk_ark_close_all(void) {
    std::lock_guard<std::mutex> lk(g_fd_mu);
    for (auto& kv : g_fds) close(kv.second);
    g_fds.clear();
}

static bool pread_all(int fd, void* buf, size_t n, int64_t off) {
    char* p = (char*)buf;
    while (n) {
        ssize_t r = pread(fd, p, n, off);
        if (r <= 0) return false;
        p += r;
        off += r;
        n -= (size_t)r;
    }
    return true;
}

// header at `off`: "\0B" "FM " \x04 int32 rows \x04 int32 cols  (15 bytes), then rows*cols float32 row-major
static int parse_header(int fd, int64_t off, int32_t* rows, int32_t* cols, int64_t* data_off, const char* path) {
    unsigned char h[15];
    if (!pread_all(fd, h, sizeof(h), off)) {
        set_err("%s:%lld: short read in matrix header", path, (long long)off);
        return -2;
    }
    if (h[0] != 0 || h[1] != 'B') {
        set_err("%s:%lld: not a binary Kaldi object (text arks are not supported by the native reader)", path, (long long)off);
        return -3;
    }
    if (memcmp(h + 2, "FM ", 3) != 0) {
        set_err("%s:%lld: matrix type '%c%c%c' unsupported (float32 'FM ' only; store features uncompressed like the "
                "reference does, feature_pre.sh:193)", path, (long long)off, h[2], h[3], h[4]);
        return -4;
    }
    if (h[5] != 4 || h[10] != 4) {
        set_err("%s:%lld: bad int32 size markers in matrix header", path, (long long)off);
        return -5;
    }
    memcpy(rows, h + 6, 4);
    memcpy(cols, h + 11, 4);
    *data_off = off + 15;
    if (*rows < 0 || *cols <= 0) {
        set_err("%s:%lld: bad matrix shape %d x %d", path, (long long)off, *rows, *cols);
        return -6;
    }
    return 0;
}

extern "C" int spk_ark_probe(int n, const char* const* paths, const int64_t* offsets, int32_t* rows, int32_t* cols,
                             int64_t* data_offsets) {
    for (int i = 0; i < n; ++i) {
        int fd = get_fd(paths[i]);
        if (fd < 0) {
            set_err("cannot open %s", paths[i]);
            return -1;
        }
        int rc = parse_header(fd, offsets[i], &rows[i], &cols[i], &data_offsets[i], paths[i]);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int spk_ark_read_crop(int B, const char* const* paths, const int64_t* data_offsets, const int32_t* rows,
                                 const int32_t* starts, int F, int T, float* out, int nthreads) {
    if (B <= 0 || F <= 0 || T <= 0 || !out) {
        set_err("spk_ark_read_crop: bad arguments");
        return -1;
    }
    for (int b = 0; b < B; ++b) {
        if (starts[b] < 0 || starts[b] + T > rows[b]) {   // the reference asserts len(full_mat) >= seq_len (datasets.py:65)
            set_err("spk_ark_read_crop: crop [%d, %d) outside utterance %d of %d frames", starts[b], starts[b] + T, b, rows[b]);
            return -7;
        }
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B;
    std::atomic<int> next(0), fail(0);
    auto work = [&]() {
        std::vector<float> tmp((size_t)T * F);
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B || fail.load()) break;
            int fd = get_fd(paths[b]);
            if (fd < 0) {
                set_err("cannot open %s", paths[b]);
                fail.store(1);
                break;
            }
            const int64_t off = data_offsets[b] + (int64_t)starts[b] * F * (int64_t)sizeof(float);
            if (!pread_all(fd, tmp.data(), tmp.size() * sizeof(float), off)) {
                set_err("%s: short read of %d frames at frame %d", paths[b], T, starts[b]);
                fail.store(1);
                break;
            }
            float* dst = out + (size_t)b * F * T;     // [F][T], time innermost (reference datasets.py:68 `.T`)
            // blocked transpose [T][F] -> [F][T]: 16x16 tiles keep both the reads and the writes inside a few cache lines
            constexpr int TB = 16;
            for (int t0 = 0; t0 < T; t0 += TB) {
                const int t1 = t0 + TB < T ? t0 + TB : T;
                for (int f0 = 0; f0 < F; f0 += TB) {
                    const int f1 = f0 + TB < F ? f0 + TB : F;
                    for (int f = f0; f < f1; ++f) {
                        float* d = dst + (size_t)f * T;
                        const float* sp = tmp.data() + f;
                        for (int t = t0; t < t1; ++t) d[t] = sp[(size_t)t * F];
                    }
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return fail.load() ? -8 : 0;
}

extern "C" int spk_io_version(void) { return 100; }
