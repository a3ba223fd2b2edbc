// Do VALU instructions of one wave run under the MFMAs of another wave on the same SIMD (gfx950)?
// Blocks of 512 threads = 8 waves = 2 per SIMD (one block per CU, 256 blocks).  mode bit 0: waves 0-3 run a chain-free
// MFMA loop; bit 1: waves 4-7 run an independent-VALU loop (v_fma_f32 on 8 accumulators); mode 4: EVERY wave alternates
// 1 MFMA : R VALU in its own instruction stream.  Times come from the host (hipEvent); results are stored so that
// nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int VPM, int NOP>   // VALU instructions per MFMA in the mixed stream; s_nop operand after every MFMA of a paced MFMA wave
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.001f + k;
    const float m = 1.0001f, c = 0.5f;
    if (mode == 4) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VPM; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], m, c);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VPM; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], m, c);
        }
    } else if (wave < 4) {
        if (mode & 16) __builtin_amdgcn_s_setprio(3);     // or the other way round
        if (mode & 32)         // MFMA waves pace themselves: the next MFMA reaches the issue port only when the pipe is about free
            for (int i = 0; i < iters; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(NOP)); __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(NOP)); __builtin_amdgcn_sched_barrier(0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(NOP)); __builtin_amdgcn_sched_barrier(0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(NOP)); __builtin_amdgcn_sched_barrier(0);
            }
        else if (mode & 64)    // one VALU instruction of its own between the MFMAs
            for (int i = 0; i < iters; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                v[0] = __builtin_fmaf(v[0], m, c);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                v[1] = __builtin_fmaf(v[1], m, c);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
                v[2] = __builtin_fmaf(v[2], m, c);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
                v[3] = __builtin_fmaf(v[3], m, c);
            }
        else if (mode & 128)   // dependent chain: every MFMA accumulates into the same registers
            for (int i = 0; i < iters; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            }
        else if (mode & 1)
            for (int i = 0; i < iters; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
            }
    } else {
        if (mode & 8) __builtin_amdgcn_s_setprio(3);      // VALU waves ahead of the MFMA waves in the issue arbiter
        if (mode & 2)
            for (int i = 0; i < iters * VPM / 2; ++i) {     // same VALU count per SIMD as MFMAs * VPM
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], m, c);
            }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += v[k];
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int VPM, int NOP = 7>
static float run(float* out, int iters, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<VPM, NOP>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<VPM, NOP>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VPM>
static void report(float* out, int iters) {
    // MFMAs per SIMD: 4 * iters (one MFMA wave per SIMD); VALU per SIMD: 4 * iters * VPM (one VALU wave per SIMD)
    const float tm = run<VPM>(out, iters, 1), tv = run<VPM>(out, iters, 2), tb = run<VPM>(out, iters, 3), tx = run<VPM>(out, iters / 2, 4);
    const float tp = run<VPM>(out, iters, 3 | 8), tq = run<VPM>(out, iters, 3 | 16);
#define PACED(N) { const float tn1 = run<VPM, N>(out, iters, 1 | 32), tn3 = run<VPM, N>(out, iters, 3 | 32); \
    printf("   s_nop %d after every MFMA of the MFMA waves: alone %.3f ms, with the VALU waves %.3f ms (%.0f %% of max(MFMA alone unpaced, VALU alone))\n", N, tn1, tn3, 100.f * tn3 / (tm > tv ? tm : tv)); }
    PACED(0) PACED(7)
    { const float t1 = run<VPM>(out, iters, 1 | 64), t3 = run<VPM>(out, iters, 3 | 64);
      printf("   MFMA waves with one VALU of their own after every MFMA: alone %.3f ms, with the VALU waves %.3f ms\n", t1, t3); }
    { const float t1 = run<VPM>(out, iters, 1 | 128), t3 = run<VPM>(out, iters, 3 | 128);
      printf("   MFMA waves with a dependent chain (same accumulator): alone %.3f ms, with the VALU waves %.3f ms\n", t1, t3); }
    printf("   with s_setprio 3 on the VALU waves: %.3f ms (%.0f %% of the max); on the MFMA waves: %.3f ms\n", tp, 100.f * tp / (tm > tv ? tm : tv), tq);
    printf("VALU per MFMA %2d: MFMA wave alone %.3f ms | VALU wave alone %.3f ms | both (separate waves) %.3f ms | one stream, both waves mixed %.3f ms"
           "   -> separate waves: %.0f %% of the sum, %.0f %% of the max\n",
           VPM, tm, tv, tb, tx, 100.f * tb / (tm + tv), 100.f * tb / (tm > tv ? tm : tv));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 200000;
    report<4>(out, iters);
    report<6>(out, iters);
    report<10>(out, iters);
    hipFree(out);
    return 0;
}
