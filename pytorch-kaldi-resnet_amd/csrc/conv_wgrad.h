// Shared by conv_wgrad.hip (fp32 operands) and conv_wgrad_split.hip (bf16-split operands).
#pragma once
#include "spk_common.h"

#ifndef WGRAD_NX
#define WGRAD_NX 5   // prefetch registers (float4) per thread for the X halo tile: halo_pix <= 32*NX
#endif
#ifndef WGRAD_ND
#define WGRAD_ND 4   // and for the dY tile: npix <= (256/(8*WN))*ND
#endif

#define WGRAD_MAX_PIX_1X1 64   // pixels per region of conv_wgrad_1x1_kernel (its prefetch registers are sized for this)

struct WgradArgs {
    const float* x;
    const float* dy;
    float* partial;
    const float* in_scale;
    const float* in_shift;
    int B, IH, IW, Cin, OH, OW, Cout;
    int S, KW, pad;
    int TH, TW, tiles_y, tiles_x, nregions, nsplit;
    int halo_h, halo_w;
    unsigned halo_w_magic, tw_magic;   // ceil(2^32 / d) for exact small-range division
    int flags;
    const unsigned* dy_amax;   // f16x3 (split 3): float bits of absmax(dy), or of an upper bound -> the dY operand scale, the power
                               //   of two that takes that value into [2^14, 2^15) (spk_sigma_from_amax_bits); required
    const unsigned* x_amax;    //   same for the X operand (absmax of the STAGED values - after a fused BatchNorm+ReLU - or a bound)
};

// WGRAD_NT bit 0: non-temporal loads of dY, bit 1: of X, in the weight-gradient kernels (A/B builds; default off)
#ifndef WGRAD_NT
#define WGRAD_NT 0
#endif
template <int BIT>
static __device__ __forceinline__ f32x4 wgrad_ld(const void* p) {
    if constexpr ((WGRAD_NT & BIT) != 0) return __builtin_nontemporal_load((const f32x4*)p);
    else return *(const f32x4*)p;
}
