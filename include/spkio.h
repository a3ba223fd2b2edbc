/* libspkio - native batch ingest of Kaldi float32 matrices (host-side C ABI, no device code).
 * Replaces, for fixed-length training batches, the per-sample path of the reference:
 *   kaldi_io.read_mat (scripts/kaldi_io.py:376-410, open_or_fd :41-71) -> random crop + transpose
 *   (scripts/datasets.py:59-72) -> default collate.
 * All pointers are HOST pointers; `out` should be pinned memory so the H2D copy can be asynchronous. */
#ifndef SPKIO_H
#define SPKIO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int spk_io_version(void);
const char* spk_io_last_error(void);
/* parse n matrix headers at (paths[i], offsets[i]) (the scp's "path:offset"): rows, cols, payload offset */
int spk_ark_probe(int n, const char* const* paths, const int64_t* offsets, int32_t* rows, int32_t* cols,
                  int64_t* data_offsets);
/* out[b][f][t] = M_b[starts[b] + t][f], t < T; reads only the cropped frames with pread() on `nthreads` threads */
int spk_ark_read_crop(int B, const char* const* paths, const int64_t* data_offsets, const int32_t* rows,
                      const int32_t* starts, int F, int T, float* out, int nthreads);
void spk_ark_close_all(void);
/* text-ark embedding writer: out <- "key [ v0 v1 ... ]\n" per row of v[n][D], each value printed exactly as numpy's
 * str(np.float32) does - the line format of the reference's scripts/decode.py:199-206.  Returns the bytes written
 * (-1 if cap < spk_text_vectors_bound). */
int64_t spk_text_vectors_bound(int n, int D, const char* const* keys);
int64_t spk_format_text_vectors(int n, int D, const float* v, const char* const* keys, char* out, int64_t cap,
                                int nthreads);
/* vector-ark reader for the scoring back end (reference: kaldi_io.read_vec_flt_ark as used by scripts/compute_mean.py:9-33 and
 * scripts/cosine_score.py:52-60): the whole ark - text 'key [ v0 ... ]' lines as scripts/decode.py:206 writes them, or binary FV / DV
 * records - into one malloc'ed [n][D] float64 matrix (text values parsed as doubles, as numpy does) and a buffer of n NUL-terminated
 * keys; parsed on `nthreads` threads.  Free both with spk_vec_ark_free.  0 on success. */
int spk_vec_ark_load(const char* path, int nthreads, int64_t* n, int32_t* D, double** data, char** keys, int64_t* keys_bytes);
void spk_vec_ark_free(double* data, char* keys);
void spk_io_set_error(const char* msg);
#ifdef __cplusplus
}
#endif
#endif
