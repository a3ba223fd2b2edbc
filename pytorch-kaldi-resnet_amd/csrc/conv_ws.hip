// Wave-specialised (producer / consumer) instantiations of the bf16-split 3x3 convolution: see conv_ws_kernel.h.
#include "conv_ws_kernel.h"

static int spk_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int MT, int NT, int WC, int SPLIT>
static int launch_ws(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    const int grid = a.nblocks < spk_num_cus() ? a.nblocks : spk_num_cus();     // persistent: one 512-thread block per CU
    if (a.flags & SPK_IN_BNBWD) {
        hipLaunchKernelGGL((conv_ws_kernel<MT, NT, WC, true, SPLIT>), dim3(grid), dim3(512), lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL((conv_ws_kernel<MT, NT, WC, false, SPLIT>), dim3(grid), dim3(512), lds_bytes, st, a);
    }
    SPK_LAUNCH_CHECK("spk_conv_mfma(ws)");
    return 0;
}

// LDS: two ring slots of the halo tile + one epilogue slab per consumer wave
size_t spk_conv_ws_lds_bytes(const ConvArgs& a, int NT, int lp4) {
    return (size_t)2 * a.halo_h * a.halo_w * lp4 * 16 + (size_t)4 * 32 * (NT * 32 + 4) * sizeof(float) + 32 * sizeof(int);   // + per-tap table
}

int spk_launch_conv_ws(const ConvArgs& a, int MT, int NT, int WC, int split, int lp4, hipStream_t st) {
    const size_t lds_bytes = spk_conv_ws_lds_bytes(a, NT, lp4);
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_mfma(ws): halo tile %dx%d needs %zu B of LDS (two ring slots + epilogue slabs)",
                a.halo_h, a.halo_w, lds_bytes);
    SPK_REQUIRE(a.kc == 1, "spk_conv_mfma(ws): kc must be 1");
#define CASE(M, N, W)                                                   \
    if (MT == M && NT == N && WC == W) {                                \
        if (split == 3) return launch_ws<M, N, W, 3>(a, lds_bytes, st); \
        if (split == 6) return launch_ws<M, N, W, 6>(a, lds_bytes, st); \
        return launch_ws<M, N, W, 9>(a, lds_bytes, st);                 \
    }
    CASE(2, 1, 1) CASE(4, 1, 1) CASE(3, 2, 1) CASE(3, 1, 2) CASE(6, 1, 2) CASE(3, 1, 4) CASE(6, 1, 4)
#undef CASE
    spk_set_error("spk_conv_mfma(ws): unsupported wave layout MT=%d NT=%d WC=%d", MT, NT, WC);
    return -1;
}
