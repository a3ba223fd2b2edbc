// bf16-split instantiations of the implicit-GEMM convolution (see conv_kernel.h, ConvCfg<SPLIT>) (the matching weight
// packing is in pack.hip).  fp32 activations and weights go in, fp32 results come out; inside, every operand is the exact sum of three
// bf16 terms and the matrix cores multiply the 6 most significant cross terms (or all 9) with fp32 accumulation; split == 3
// is the fp16 two-term form (three products on v_mfma_f32_32x32x16_f16, power-of-two operand scales).
#include "conv_kernel.h"

template <int MT, int NT, int SPLIT>
static int launch_split(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    if (a.flags & SPK_IN_BNBWD)
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true, SPLIT>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false, SPLIT>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_mfma(split)");
    return 0;
}

int spk_launch_conv_split(const ConvArgs& a, size_t lds_bytes, int MT, int NT, int split, hipStream_t st) {
#define CASE(M, N)                                                      \
    if (MT == M && NT == N) {                                           \
        if (split == 3) return launch_split<M, N, 3>(a, lds_bytes, st); \
        if (split == 6) return launch_split<M, N, 6>(a, lds_bytes, st); \
        return launch_split<M, N, 9>(a, lds_bytes, st);                 \
    }
    CASE(1, 1) CASE(2, 1) CASE(3, 1) CASE(4, 1) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(1, 4)
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported split tile config MT=%d NT=%d", MT, NT);
    return -1;
}

#ifdef CONV_STAMPS
extern "C" int spk_debug_stamps(unsigned long long* out, int nblocks) {     // diagnostic builds only (never in the in-tree library)
    if (nblocks < 0) {        // reset
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_stamps)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 16 * CONV_STAMP_BLOCKS);
    }
    if (nblocks > CONV_STAMP_BLOCKS) nblocks = CONV_STAMP_BLOCKS;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), (size_t)nblocks * 16 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
