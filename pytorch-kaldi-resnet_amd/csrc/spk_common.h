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
    SPK_CONV_WS = 128,        // launch the producer/consumer (wave-specialised, persistent) kernel: bf16-split 3x3 only
                              //   (bits 8-9 of the flags word then carry log2 of its consumer-wave channel groups)
    SPK_CONV_PIPE = 1024,     // launch the in-wave pipelined kernel (conv_kernel.h, PIPE): f16x3, 3x3, plain input
    SPK_WGRAD_GROUPS = 2048,  // spk_conv_wgrad, 1x1, f16x3: conv_wgrad_1x1_kernel with 1 << (flags bits 12-13) channel groups
    // f16x3 "pair" tensors (see below): a gradient tensor written ONCE in the two-term fp16 form by its producer and staged by
    // plain copy in every matrix-core kernel that consumes it
    SPK_IN_PRESPLIT = 1 << 14,    // spk_conv_mfma: `in` holds f16 pairs scaled by the sigma of in_amax (no input transform)
    SPK_SIDE_PRESPLIT = 1 << 15,  // spk_conv_mfma + IN_BNBWD: side_draw is written as f16 pairs (scale: the sigma of in_amax)
    SPK_DY_PRESPLIT = 1 << 16,    // spk_conv_wgrad: `dy` holds f16 pairs scaled by the sigma of dy_amax
    SPK_CONV_M16 = 1 << 17,       // with SPK_CONV_PIPE: the v_mfma_f32_16x16x32_f16 form of the pipelined kernel (taps paired per K step)
    SPK_WGRAD_NOSHIFT = 1 << 18,  // spk_conv_wgrad, 3x3 grouped kernel: keep the plain K loop where the shifted-window form applies (A/B)
    SPK_WGRAD_M16 = 1 << 19       // spk_conv_wgrad, 3x3 grouped kernel with a pair-tensor dy: the v_mfma_f32_16x16x32_f16 form, dy by LDS DMA
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
// ---- fp16 operand splitting ("f16x3" mode): u = v * sigma (sigma a power of two chosen so that the tensor's largest
// magnitude lands in [2^14, 2^15): spk_sigma_from_amax_bits) is the sum of two fp16 terms up to 2^-22 |u| (2 x 11 significand bits); a product is formed
// from the three cross terms h1*g1 + h1*g2 + h2*g1 on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped h2*g2 is
// 2^-22 relative) and the accumulator is scaled back by 1 / (sigma_a * sigma_b).  Measured against fp64
// (tools/probe/split_probe.hip, profiles/r02_split_probe.log): the same error as the native fp32 matrix instruction and
// the 6-term bf16 split - the fp32 accumulation dominates - at half the matrix instructions of the latter.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define SPK_F16_ACT_SIGMA 64.0f       // fallback input scale when no absmax slot is given (the engine always gives one)
// w * sigma (saturated to the fp16 range) -> two fp16 quads (8 bytes each).  Written on vectors so that the compiler
// emits the packed forms (v_pk_mul_f32, v_cvt_pk_f16_f32: round to nearest even, two values per instruction): 18 VALU
// instructions per float4 instead of 32 - the staging phases of the f16x3 kernels are VALU-bound on exactly this.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void split2h(f32x4 w, float sigma, uint2& t0, uint2& t1) {
    f32x4 u = w * sigma;
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = __builtin_amdgcn_fmed3f(u[k], -65504.f, 65504.f);
    const f16x4 a = __builtin_convertvector(u, f16x4);
    const f32x4 r = u - __builtin_convertvector(a, f32x4);      // exact in fp32
    const f16x4 b = __builtin_convertvector(r, f16x4);
    t0 = __builtin_bit_cast(uint2, a);
    t1 = __builtin_bit_cast(uint2, b);
}
// ---- f16 pair tensors.  Same shape and addressing as the fp32 NHWC tensor they stand for; every aligned group of four
// floats (16 bytes) is replaced by [4 x fp16 high term][4 x fp16 low term] of value * sigma, sigma = the power of two that
// spk_sigma_from_amax_bits derives from the tensor's scale slot (a rigorous upper bound of its absmax, known BEFORE the
// producer runs).  The producer (spk_bn_bwd_apply, or the side output of a fused BatchNorm-backward data gradient) converts
// each value exactly once; the data gradient and the weight gradient that consume it stage 16-byte groups by plain copy
// (two 8-byte LDS writes, no conversion), where they used to convert the same value Cout/64 + Cin/64 times.
static __device__ __forceinline__ f32x4 spk_pair_pack(uint2 t0, uint2 t1) {
    return __builtin_bit_cast(f32x4, (uint4){t0.x, t0.y, t1.x, t1.y});
}
static __device__ __forceinline__ void spk_pair_unpack(f32x4 v, uint2& t0, uint2& t1) {
    const uint4 u = __builtin_bit_cast(uint4, v);
    t0 = (uint2){u.x, u.y};
    t1 = (uint2){u.z, u.w};
}
// the two fp16 terms of one float4 of an operand: converted here, or - `pairs`, wave-uniform - taken as stored (f16 pair tensor)
static __device__ __forceinline__ void spk_terms(f32x4 w, float sigma, bool pairs, uint2& t0, uint2& t1) {
    if (pairs) spk_pair_unpack(w, t0, t1);
    else split2h(w, sigma, t0, t1);
}
// power-of-two scale from the bits of a tensor's absmax (or of a rigorous upper bound of it): bound * sigma in [2^14, 2^15),
// just under the fp16 maximum (65504), so nothing can saturate.  Precision of a staged value u = v * sigma:
//   |u| >= 2^-3  (v >= bound * 2^-18): both terms are normal fp16 numbers: 22 significand bits;
//   below that the LOW term is an fp16 subnormal (resolution 2^-24), below 2^-14 the high term too.  v_mfma_f32_32x32x16_f16
//   KEEPS subnormal operands (tools/probe/run_split_probe.py, profiles/r03_split_probe.log: 2^-24 ... 2^-15 come through
//   exactly, as A and as B operand; round 2 claimed the opposite from a broken probe), and v_cvt_f16_f32 produces them,
//   so such values are carried with an ABSOLUTE error <= 2^-25 / sigma <= bound * 2^-39 - relative precision degrades
//   gracefully from 22 bits at bound * 2^-18 to 11 bits at bound * 2^-28 and is lost at bound * 2^-39.
// (tests/test_kernels_gpu.py::test_f16x3_precision_floor_below_the_scale_window; the bf16 modes have fp32's exponent range.)
// Every scale comes from a true absmax or from a rigorous bound (bn.hip: bn_finalize_kernel for relu(raw*scale+shift),
// bn_bwd_finalize_kernel / bnbwd_bound for the BatchNorm-backward values): no heuristic and no headroom factor is left.
static __device__ __forceinline__ float spk_sigma_from_amax_bits(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu);            // biased exponent; amax = m * 2^(e - 127), m in [1, 2)
    if (e == 0 || e == 255) return 1.f;                   // zero / subnormal / inf / nan: no scaling
    int se = 127 + 14 - (e - 127);                        // sigma = 2^(14 - (e - 127))
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}
// |v| for the absmax hand-offs, with non-finite values left out (0): an inf in a gradient tensor - the pooling layer's sqrt'(0)
// at a dead channel row, dropped one step later by the ReLU mask select exactly as in the reference - would turn the slot into
// inf and the operand scale of the WHOLE tensor into 1 (measured on a trained checkpoint: 40 % gradient error in layer 4)
static __device__ __forceinline__ float spk_finite_abs(float v) {
    const float a = fabsf(v);
    return a < __builtin_inff() ? a : 0.f;      // false for inf and NaN
}
// wave-wide max of a non-negative float, then one atomicMax on its bit pattern (order-independent: deterministic)
static __device__ __forceinline__ void spk_wave_amax_commit(float v, unsigned* dst) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    // atomics on one address serialise at the memory side (~50-100 ns each): a launch with 10^5 waves would spend
    // milliseconds there.  The slot only grows, so a wave first looks at it (device-scope load: may lag, never leads) and
    // skips the atomic when its own maximum cannot raise it - after the first few waves almost all of them skip.
    if ((threadIdx.x & 63) == 0) {
        const unsigned bits = __float_as_uint(v);
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}
#endif

// Small-range integer arithmetic on the full-rate 24-bit multiplier (v_mul_u32_u24; the 32-bit v_mul_lo / v_mul_hi run at a
// quarter of the VALU rate and the staging code of the conv kernels is VALU-bound): p / d for p < 2048 and p * d < 2^20 with
// m20 = ceil(2^20 / d), derived from the 32-bit reciprocal the launchers already pass (ceil(2^32 / d)).
static __device__ __forceinline__ unsigned spk_m20(unsigned magic32) { return (magic32 + 0xFFFu) >> 12; }
static __device__ __forceinline__ int spk_div20(int p, unsigned m20) { return (int)(__umul24((unsigned)p, m20) >> 20); }

