// Probe: fp32 GEMM tile on the matrix cores three ways - native v_mfma_f32_32x32x2_f32 and bf16 operand splitting
// (a = a1 + a2 + a3 with bf16 terms, products on v_mfma_f32_32x32x16_bf16, fp32 accumulate) with 9 / 6 / 3 cross terms.
// One wave computes C[32][32] = A[32][K] * B[K][32]; K % 16 == 0.  Used to measure the accuracy of each variant against
// an fp64 result and to confirm the operand layout of the 32x32x16 instruction.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static __device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

static __device__ __forceinline__ void split3(const float* v, s16x8& s1, s16x8& s2, s16x8& s3) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short b1 = bf16_rne(v[i]);
        const float r1 = v[i] - bf16_to_f32(b1);
        const unsigned short b2 = bf16_rne(r1);
        const float r2 = r1 - bf16_to_f32(b2);
        const unsigned short b3 = bf16_rne(r2);
        s1[i] = (short)b1; s2[i] = (short)b2; s3[i] = (short)b3;
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// fp16 two-term split of v * sigma (sigma a power of two): u = h1 + h2 + O(2^-22 |u|); saturating
static __device__ __forceinline__ void split2h(const float* v, float sigma, f16x8& h1, f16x8& h2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float u = v[i] * sigma;
        u = fminf(fmaxf(u, -65504.f), 65504.f);
        const _Float16 a = (_Float16)u;
        h1[i] = a;
        h2[i] = (_Float16)(u - (float)a);
    }
}

__global__ void split_probe_kernel(const float* A, const float* B, float* C, int K, int variant, int reps, float sa_, float sb_) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        if (variant == 23 || variant == 24) {
            // variant 23: h1g1 + h1g2 + h2g1 (3 products); 24: + h2g2
            for (int k = 0; k < K; k += 16) {
                float av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    av[i] = A[r * K + k + 8 * h + i];
                    bv[i] = B[(k + 8 * h + i) * 32 + r];
                }
                f16x8 a1, a2, b1, b2;
                split2h(av, sa_, a1, a2);
                split2h(bv, sb_, b1, b2);
                if (variant == 24) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
            }
        } else if (variant == 0) {
            for (int k = 0; k < K; k += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
        } else {
            for (int k = 0; k < K; k += 16) {
                float av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    av[i] = A[r * K + k + 8 * h + i];
                    bv[i] = B[(k + 8 * h + i) * 32 + r];
                }
                s16x8 a[3], b[3];
                split3(av, a[0], a[1], a[2]);
                split3(bv, b[0], b[1], b[2]);
                // smallest terms first
#pragma unroll
                for (int s = 4; s >= 0; --s)
#pragma unroll
                    for (int sa = 0; sa < 3; ++sa) {
                        const int sb = s - sa;
                        if (sb < 0 || sb > 2) continue;
                        if (variant == 6 && s > 2) continue;
                        if (variant == 3 && s > 1) continue;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[sa]), __builtin_bit_cast(bf16x8, b[sb]), acc, 0, 0, 0);
                    }
            }
        }
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        C[row * 32 + r] = (variant == 23 || variant == 24) ? acc[e] / (sa_ * sb_) : acc[e];
    }
}

extern "C" int split_probe(const float* A, const float* B, float* C, int K, int variant, int reps, void* stream, float sa, float sb) {
    hipLaunchKernelGGL(split_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K, variant, reps, sa, sb);
    return (int)hipGetLastError();
}


// fp16 subnormal operands of v_mfma_f32_32x32x16_f16: C[i][j] = sum_k A[i][k] B[k][j] with A[i][0] = the fp16 whose bits are
// a_bits (other k: 0) and B[0][j] = the fp16 whose bits are b_bits.  Operands are built from BIT PATTERNS (no conversion
// instruction in between that could flush), the accumulator is fp32: out[0] = C[0][0].
__global__ void subnormal_probe_kernel(unsigned short a_bits, unsigned short b_bits, float* out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
    if (h == 0) { a[0] = (short)a_bits; b[0] = (short)b_bits; }      // k = 8 h + 0
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}
extern "C" int subnormal_probe(unsigned a_bits, unsigned b_bits, float* out, void* stream) {
    hipLaunchKernelGGL(subnormal_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned short)a_bits, (unsigned short)b_bits, out);
    return (int)hipGetLastError();
}
