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
    SPK_IN_BNBWD = 64,        // the staged input is BatchNorm-backward(in) computed on the fly (stride-1 data gradients)
    SPK_CONV_WS = 128         // launch the producer/consumer (wave-specialised, persistent) kernel: bf16-split 3x3 only
};

static inline int spk_ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- bf16 operand splitting (device): an fp32 value is the exact sum of three bf16 terms (3 x 8 significand bits)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef __HIPCC__
static __device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi, float& rlo, float& rhi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;   // round to nearest even
    rlo = lo - (float)a;                           // exact in fp32
    rhi = hi - (float)b;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
// w -> three bf16 quads (8 bytes each): w = t0 + t1 + t2
static __device__ __forceinline__ void split3(f32x4 w, uint2& t0, uint2& t1, uint2& t2) {
    float r0, r1, r2, r3, q0, q1, q2, q3;
    t0.x = pack_bf16x2(w[0], w[1], r0, r1);
    t0.y = pack_bf16x2(w[2], w[3], r2, r3);
    t1.x = pack_bf16x2(r0, r1, q0, q1);
    t1.y = pack_bf16x2(r2, r3, q2, q3);
    t2.x = pack_bf16x2(q0, q1, r0, r1);
    t2.y = pack_bf16x2(q2, q3, r2, r3);
}
#endif

