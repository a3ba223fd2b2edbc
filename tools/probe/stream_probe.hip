// How fast can an elementwise pass (two tensors read, one written, 786 MB each: the block-output BatchNorm apply of layer 1)
// stream on this chip, and does the cache policy of the loads / stores matter?  Variants: plain, non-temporal stores,
// non-temporal loads + stores; one 16-byte group per thread per tensor and grid-stride loops of 2 / 4 groups.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT_LD, int NT_ST, int U>
__global__ __launch_bounds__(256) void apply(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, long long n4,
                                             const float* __restrict__ sc) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const f32x4 s = *(const f32x4*)(sc + (threadIdx.x & 7) * 4);
    for (; i < n4; i += stride * U) {
        f32x4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long k = i + u * stride < n4 ? i + u * stride : i;
            va[u] = NT_LD ? __builtin_nontemporal_load(a + k) : a[k];
            vb[u] = NT_LD ? __builtin_nontemporal_load(b + k) : b[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 v = va[u] * s + vb[u];
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            if (i + u * stride < n4) {
                if (NT_ST) __builtin_nontemporal_store(v, o + i + u * stride);
                else o[i + u * stride] = v;
            }
        }
    }
}

int main() {
    const long long n = 256LL * 80 * 300 * 32, n4 = n / 4;
    f32x4 *a, *b, *o;
    float* sc;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&o, n * 4)); CHECK(hipMalloc(&sc, 256));
    CHECK(hipMemset(a, 0x3c, n * 4)); CHECK(hipMemset(b, 0x3b, n * 4)); CHECK(hipMemset(sc, 0x3e, 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int U, int blocks) {
        const int grid = blocks > 0 ? blocks : (int)((n4 + 256LL * U - 1) / (256LL * U));
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, o, n4, sc);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 10 < best) best = ms / 10;
        }
        printf("%-44s grid %7d  %.3f ms  %.2f TB/s\n", name, grid, best, 3.0 * n * 4 / (best * 1e-3) * 1e-12);
    };
    run("plain, 1 group/thread", apply<0, 0, 1>, 1, 0);
    run("plain, 2 groups/thread", apply<0, 0, 2>, 2, 0);
    run("plain, 4 groups/thread", apply<0, 0, 4>, 4, 0);
    run("nt stores, 1 group/thread", apply<0, 1, 1>, 1, 0);
    run("nt stores, 2 groups/thread", apply<0, 1, 2>, 2, 0);
    run("nt loads + stores, 1 group/thread", apply<1, 1, 1>, 1, 0);
    run("nt loads + stores, 2 groups/thread", apply<1, 1, 2>, 2, 0);
    run("nt loads + stores, 4 groups/thread", apply<1, 1, 4>, 4, 0);
    run("plain, grid-stride 2048 blocks x2", apply<0, 0, 2>, 2, 2048);
    run("plain, grid-stride 8192 blocks x2", apply<0, 0, 2>, 2, 8192);
    run("nt l+s, grid-stride 2048 blocks x2", apply<1, 1, 2>, 2, 2048);
    run("nt l+s, grid-stride 8192 blocks x4", apply<1, 1, 4>, 4, 8192);
    CHECK(hipGetLastError());
    return 0;
}
