// In-wave pipelined instantiations of the implicit-GEMM convolution (conv_kernel.h, PIPE): f16x3 operands, nine taps,
// the next chunk staged in the shadow of the current chunk's matrix instructions, two LDS tiles, one barrier per chunk.
#include "conv_kernel.h"

template <int MT, int NT>
static int launch_pipe(const ConvArgs& a, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL((conv_pipe_kernel<MT, NT, false>), dim3(a.nblocks), dim3(256), lds_bytes, st, a);
    SPK_LAUNCH_CHECK("spk_conv_mfma(pipe)");
    return 0;
}

int spk_launch_conv_pipe(const ConvArgs& a, size_t lds_bytes, int MT, int NT, hipStream_t st) {
#define CASE(M, N) if (MT == M && NT == N) return launch_pipe<M, N>(a, lds_bytes, st);
    CASE(1, 1) CASE(2, 1) CASE(3, 1) CASE(4, 1) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(1, 4)
#undef CASE
    spk_set_error("spk_conv_mfma: unsupported pipelined tile config MT=%d NT=%d", MT, NT);
    return -1;
}
