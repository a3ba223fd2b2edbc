// Probe: semantics of global_load_lds_dwordx4 (per-lane global address, LDS destination = wave-uniform base + lane*16).
#include <hip/hip_runtime.h>
extern "C" __global__ void dma_test(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < n; ++it) {
        const float* g = src + (size_t)idx[(it * 4 + wave) * 64 + lane] * 4;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                         (void __attribute__((address_space(3)))*)(lds + ((it * 4 + wave) * 64) * 4), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < n * 256 * 4; i += 256) out[i] = lds[i];
}
extern "C" int dma_probe(const float* src, const int* idx, float* out, int n, void* stream) {
    hipLaunchKernelGGL(dma_test, dim3(1), dim3(256), (size_t)n * 256 * 16, (hipStream_t)stream, src, idx, out, n);
    return (int)hipGetLastError();
}
