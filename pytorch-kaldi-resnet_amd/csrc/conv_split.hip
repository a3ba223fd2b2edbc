// bf16-split instantiations of the implicit-GEMM convolution (see conv_kernel.h, ConvCfg<SPLIT>) and the matching weight
// packing.  fp32 activations and weights go in, fp32 results come out; inside, every operand is the exact sum of three
// bf16 terms and the matrix cores multiply the 6 most significant cross terms (or all 9) with fp32 accumulation.
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
        if (split == 6) return launch_split<M, N, 6>(a, lds_bytes, st); \
        return launch_split<M, N, 9>(a, lds_bytes, st);                 \
    }
    CASE(1, 1) CASE(2, 1) CASE(3, 1) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(1, 4)
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported split tile config MT=%d NT=%d", MT, NT);
    return -1;
}

// OIHW [Cout][Cin][KH][KW] fp32 -> [tap][K/16][term 0..2][N/32][64 lanes][8 bf16]: lane l holds, for n = 32*nt + (l&31),
// the eight k values 16*g + 8*(l>>5) + {0..7} of one term (B operand of v_mfma_f32_32x32x16_bf16).
__global__ void pack_conv_weight_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                              int KHW, int transpose, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    int i = idx;
    const int e = i & 7; i >>= 3;
    const int lane = i & 63; i >>= 6;
    const int nt = i % (N >> 5); i /= (N >> 5);
    const int g = i % (K >> 4);
    const int t = i / (K >> 4);
    const int n = nt * 32 + (lane & 31);
    const int k = g * 16 + (lane >> 5) * 8 + e;
    const int co = transpose ? k : n, ci = transpose ? n : k;
    float x = w[((size_t)co * Cin + ci) * KHW + t];
    const size_t term = (size_t)(N >> 5) * 512;
    size_t o = ((((size_t)t * (K >> 4) + g) * 3) * (N >> 5) + nt) * 512 + lane * 8 + e;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const __bf16 b = (__bf16)x;
        wpk[o + s * term] = __builtin_bit_cast(unsigned short, b);
        x -= (float)b;
    }
}

extern "C" int spk_pack_conv_weight_split(const float* w, void* wpk, int Cout, int Cin, int KH, int KW, int transpose,
                                          void* stream) {
    SPK_REQUIRE(w && wpk, "spk_pack_conv_weight_split: null pointer");
    SPK_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "spk_pack_conv_weight_split: channels (%d,%d) must be multiples of 32", Cout, Cin);
    SPK_REQUIRE(KH * KW >= 1 && KH * KW <= 9, "spk_pack_conv_weight_split: kernel %dx%d unsupported", KH, KW);
    const int total = Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_conv_weight_split_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)wpk, Cout, Cin, KH * KW, transpose, total);
    SPK_LAUNCH_CHECK("spk_pack_conv_weight_split");
    return 0;
}
