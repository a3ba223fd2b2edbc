// Fused SGD with momentum and weight decay over one flat fp32 parameter arena
// (torch.optim.SGD as used at reference scripts/train_resnet.py:203-205,328):
//   g = grad*grad_scale + wd*p;  buf = first ? g : momentum*buf + g;  p -= lr*buf
// grad_scale carries the 1/world_size of data-parallel gradient averaging.
#include "spk_common.h"

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long long n, float lr, float momentum, float wd, float gscale, int first) {
    const long long nq = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        f32x4 pv = *(f32x4*)(p + i * 4);
        f32x4 gv = *(const f32x4*)(g + i * 4) * gscale + pv * wd;
        f32x4 bv = first ? gv : *(f32x4*)(buf + i * 4) * momentum + gv;
        *(f32x4*)(buf + i * 4) = bv;
        *(f32x4*)(p + i * 4) = pv - bv * lr;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (nq << 2) + threadIdx.x;
        const float gv = g[i] * gscale + p[i] * wd;
        const float bv = first ? gv : buf[i] * momentum + gv;
        buf[i] = bv;
        p[i] -= bv * lr;
    }
}

extern "C" int spk_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay,
                            float grad_scale, int first_step, void* stream) {
    SPK_REQUIRE(p && g && buf && n > 0, "spk_sgd_step: bad arguments");
    SPK_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)buf & 15) == 0,
                "spk_sgd_step: arenas must be 16-byte aligned");
    long long nb = ((n >> 2) + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, weight_decay,
                       grad_scale, first_step);
    SPK_LAUNCH_CHECK("spk_sgd_step");
    return 0;
}
