// Shared declarations for libspkhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// error plumbing: every export returns 0 on success, <0 for bad arguments, >0 = hipError_t
void spk_set_error(const char* fmt, ...);
#define SPK_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            spk_set_error(__VA_ARGS__);        \
            return -1;                         \
        }                                      \
    } while (0)
#define SPK_LAUNCH_CHECK(name)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            spk_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

// fused input transform / epilogue flags of the MFMA convolution (see include/spkhip.h)
enum {
    SPK_IN_AFFINE_RELU = 1,   // a = max(in*scale[c]+shift[c], 0) while staging the input tile
    SPK_EPI_AFFINE = 2,       // v = v*epi_scale[c] + epi_shift[c]
    SPK_EPI_ADD = 4,          // v += epi_add[same address as out]
    SPK_EPI_RELU = 8,         // v = max(v, 0)
    SPK_EPI_STATS = 16,       // per-wave per-channel (sum, sumsq) of the stored values
    SPK_EPI_BNBWD = 32,       // with EPI_STATS: (sum dz, sum dz*xhat) of the BatchNorm the output is a gradient of
    SPK_IN_BNBWD = 64         // the staged input is BatchNorm-backward(in) computed on the fly (stride-1 data gradients)
};

static inline int spk_ceil_div(int a, int b) { return (a + b - 1) / b; }
