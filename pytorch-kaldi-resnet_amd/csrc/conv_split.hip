// bf16-split instantiations of the implicit-GEMM convolution (see conv_kernel.h, ConvCfg<SPLIT>) (the matching weight
// packing is in pack.hip).  fp32 activations and weights go in, fp32 results come out; inside, every operand is the exact sum of three
// bf16 terms and the matrix cores multiply the 6 most significant cross terms (or all 9) with fp32 accumulation; split == 3
// is the fp16 two-term form (three products on v_mfma_f32_32x32x16_f16, power-of-two operand scales).
#include "conv_kernel.h"

template <int MT, int NT, int SPLIT>
static int launch_split(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    if (a.flags & SPK_IN_BNBWD) {
#ifndef SPK_NO_FL_VARIANTS
        // the two fused BatchNorm-backward data gradients of a 32-channel BasicBlock as compile-time variants: conv2's (mask of the
        // block output as sign bits, pair side output, statistics of bn1's backward with the mask recomputed) and conv1's (mask
        // recomputed, masked shortcut add, statistics of the previous block's last BatchNorm by sign bits)
        if constexpr (SPLIT == 3 && MT == 2 && NT == 1) {
            const int var = a.flags | (a.add_mask ? SPK_FL_ADDMASK : 0) | (a.bn_mask ? SPK_FL_BNMASK : 0) | (a.in_mask ? SPK_FL_INMASK : 0);
            constexpr int V2 = SPK_IN_BNBWD | SPK_SIDE_PRESPLIT | SPK_EPI_STATS | SPK_EPI_BNBWD | SPK_FL_INMASK;
            constexpr int V1 = SPK_IN_BNBWD | SPK_SIDE_PRESPLIT | SPK_EPI_ADD | SPK_EPI_STATS | SPK_EPI_BNBWD | SPK_FL_ADDMASK | SPK_FL_BNMASK;
            if (!a.in_act && !a.bn_act && !a.side_dz && (var == V1 || var == V2)) {
                if (var == V2) hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true, SPLIT, V2>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
                else hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true, SPLIT, V1>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
                SPK_LAUNCH_CHECK("spk_conv_mfma(split)");
                return 0;
            }
        }
#endif
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, true, SPLIT>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    } else {
#ifndef SPK_NO_FL_VARIANTS
        // the forward launches of the training step as compile-time flag variants (conv_kernel.h, FL), f16x3 mode
        if constexpr (SPLIT == 3 && ((MT == 2 && NT == 1) || (MT == 1 && NT == 2) || (MT == 2 && NT == 2))) {
            if (!a.add_mask && !a.bn_mask && !a.bn_act) {
                if (a.flags == (SPK_IN_AFFINE_RELU | SPK_EPI_STATS)) {
                    hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false, SPLIT, SPK_IN_AFFINE_RELU | SPK_EPI_STATS>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
                    SPK_LAUNCH_CHECK("spk_conv_mfma(split)");
                    return 0;
                }
                if (a.flags == SPK_EPI_STATS) {
                    hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false, SPLIT, SPK_EPI_STATS>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
                    SPK_LAUNCH_CHECK("spk_conv_mfma(split)");
                    return 0;
                }
            }
        }
#endif
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, false, SPLIT>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    }
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
