// In-wave pipelined instantiations of the implicit-GEMM convolution (conv_kernel.h, PIPE): f16x3 operands, nine taps,
// the next chunk staged in the shadow of the current chunk's matrix instructions, two LDS tiles, one barrier per chunk.
#include "conv_kernel.h"

template <int MT, int NT>
static int launch_pipe(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    if (a.flags & SPK_IN_BNBWD) {
#ifndef SPK_EXPERIMENTAL
        spk_set_error("spk_conv_mfma: the pipelined fused-BatchNorm-backward kernel is an experimental form: build with SPK_EXPERIMENTAL=1");
        return -1;
#else
        // fused BatchNorm backward: the sign-bit form only (the form that recomputes the mask from the raw tensor has more
        // VALU work per item than a tap has MFMA shadow and measured slower than conv_mfma_kernel), register tiles <= 2 x 2
        if constexpr (MT * NT <= 4) {
            SPK_REQUIRE(a.in_mask, "spk_conv_mfma: the pipelined fused-BatchNorm-backward kernel takes the ReLU mask as sign bits");
            hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, true, true>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
        } else {
            spk_set_error("spk_conv_mfma: no pipelined fused-BatchNorm-backward kernel for MT=%d NT=%d", MT, NT);
            return -1;
        }
#endif
    } else if (a.flags & SPK_CONV_M16) {         // 16x16x32 matrix instruction, taps paired (conv_kernel.h, M16)
        ConvArgs b = a;
        b.flags = a.flags & ~SPK_CONV_M16;
        if constexpr (MT == 3 && NT == 2) {
            // the flag combinations of the training step as compile-time variants (conv_kernel.h, FL): pair-input data gradients with the BatchNorm-backward statistics (mask recomputed from
            // the raw tensor), with the masked shortcut add + the statistics by sign bits, with the masked add alone; anything
            // else (eval-mode epilogues, activation tensors as masks): the generic instantiation
            const int var = b.flags | (b.add_mask ? SPK_FL_ADDMASK : 0) | (b.bn_mask ? SPK_FL_BNMASK : 0) | (b.bn_act ? (1 << 30) : 0);
#define LP(PREV, VV) hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, false, false, PREV, true, VV>), dim3(a.nblocks), dim3(256), lds_bytes, st, b)
#ifdef SPK_NO_FL_VARIANTS
            if (a.flags & SPK_IN_PRESPLIT) LP(true, -1); else LP(false, -1);
#else
            switch (var) {
                // (the forward form sits at its 256-register bound: its variants spill 10 registers - generic instantiation unless
                //  SPK_FL_FWD_PIPE is defined: A/B builds)
#ifdef SPK_FL_FWD_PIPE
                case SPK_IN_AFFINE_RELU | SPK_EPI_STATS: LP(false, SPK_IN_AFFINE_RELU | SPK_EPI_STATS); break;
                case SPK_EPI_STATS: LP(false, SPK_EPI_STATS); break;
#endif
                case SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_BNBWD: LP(true, SPK_IN_PRESPLIT | SPK_EPI_STATS | SPK_EPI_BNBWD); break;
                case SPK_IN_PRESPLIT | SPK_EPI_ADD | SPK_EPI_STATS | SPK_EPI_BNBWD | SPK_FL_ADDMASK | SPK_FL_BNMASK:
                    LP(true, SPK_IN_PRESPLIT | SPK_EPI_ADD | SPK_EPI_STATS | SPK_EPI_BNBWD | SPK_FL_ADDMASK | SPK_FL_BNMASK); break;
                case SPK_IN_PRESPLIT | SPK_EPI_ADD | SPK_FL_ADDMASK: LP(true, SPK_IN_PRESPLIT | SPK_EPI_ADD | SPK_FL_ADDMASK); break;
                default:
                    if (a.flags & SPK_IN_PRESPLIT) LP(true, -1); else LP(false, -1);
            }
#endif
#undef LP
        } else {
            spk_set_error("spk_conv_mfma: no 16x16x32 pipelined kernel for MT=%d NT=%d", MT, NT);
            return -1;
        }
    } else if (a.flags & SPK_IN_PRESPLIT)        // f16 pair input: staging by plain copy
        hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, false, false, true>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    else
        hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, false>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_mfma(pipe)");
    return 0;
}

int spk_launch_conv_pipe(const ConvArgs& a, size_t lds_bytes, int MT, int NT, hipStream_t st) {
#define CASE(M, N) if (MT == M && NT == N) return launch_pipe<M, N>(a, lds_bytes, st);
    CASE(2, 1) CASE(3, 1) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(1, 4)
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported pipelined tile config MT=%d NT=%d", MT, NT);
    return -1;
}

#ifdef CONV_STAMPS
extern "C" int spk_debug_stamps_pipe(unsigned long long* out, int nblocks) {     // diagnostic builds only (never in the in-tree library)
    if (nblocks < 0) {        // reset
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_stamps)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 16 * CONV_STAMP_BLOCKS);
    }
    if (nblocks > CONV_STAMP_BLOCKS) nblocks = CONV_STAMP_BLOCKS;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), (size_t)nblocks * 16 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
