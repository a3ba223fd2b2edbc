// Weight packing for the matrix-core convolutions: nn.Conv2d OIHW fp32 weights -> MFMA B-fragment order, once per step
// for the forward (K = Cin) and once transposed for the data gradient (K = Cout).
//   fp32 operands : [tap][K/8][N/32][64 lanes][4 floats]      lane l: n = 32 nt + (l & 31), k = 8 g + 4 (l >> 5) + {0..3}
//   bf16 split    : [tap][K/16][term 0..2][N/32][64 lanes][8 bf16]  lane l: n as above, k = 16 g + 8 (l >> 5) + {0..7};
//                   term s of a weight is the s-th bf16 of its exact three-term split (w = t0 + t1 + t2)
//   fp16 split    : (split = 3) a 16-byte header - word 0 = float bits of max|w| over the tensor - followed by the same order
//                   with two terms: w * sigma = h0 + h1 (+ 2^-22 relative), sigma = the power of two that puts max|w| in
//                   [2^14, 2^15) (spk_sigma_from_amax_bits: the convolution kernels derive the same sigma from the header)
// spk_pack_conv_weights_batched packs every convolution of the network in ONE launch from a device-resident job table
// (the weights change every step, so this runs once per step: 70 tiny launches become one).
#include "spk_common.h"

static __device__ __forceinline__ void pack_f32_elem(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin,
                                                     int KHW, int transpose, int idx) {
    const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    int i = idx;
    const int s = i & 3; i >>= 2;
    const int lane = i & 63; i >>= 6;
    const int nt = i % (N >> 5); i /= (N >> 5);
    const int g = i % (K >> 3);
    const int t = i / (K >> 3);
    const int n = nt * 32 + (lane & 31);
    const int k = g * 8 + (lane >> 5) * 4 + s;
    const int co = transpose ? k : n, ci = transpose ? n : k;
    wpk[idx] = w[((size_t)co * Cin + ci) * KHW + t];
}

template <int NTERM>
static __device__ __forceinline__ void pack_split_elem(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout,
                                                       int Cin, int KHW, int transpose, int idx, float sigma = 1.f) {
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
    const size_t o = ((((size_t)t * (K >> 4) + g) * NTERM) * (N >> 5) + nt) * 512 + lane * 8 + e;
    if constexpr (NTERM == 2) {
        wpk += 8;                                           // past the 16-byte header
        x = fminf(fmaxf(x * sigma, -65504.f), 65504.f);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const _Float16 b = (_Float16)x;
            wpk[o + s * term] = __builtin_bit_cast(unsigned short, b);
            x -= (float)b;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const __bf16 b = (__bf16)x;
            wpk[o + s * term] = __builtin_bit_cast(unsigned short, b);
            x -= (float)b;
        }
    }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin, int KHW,
                                        int transpose, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) pack_f32_elem(w, wpk, Cout, Cin, KHW, transpose, idx);
}

__global__ void pack_conv_weight_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int Cout, int Cin,
                                              int KHW, int transpose, int total, int split) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    if (split == 3) pack_split_elem<2>(w, wpk, Cout, Cin, KHW, transpose, idx, spk_sigma_from_amax_bits(*(const unsigned*)wpk));
    else pack_split_elem<3>(w, wpk, Cout, Cin, KHW, transpose, idx);
}

// header word 0 of an fp16-split pack <- float bits of max|w| (zeroed first; the tensor is a few 10^5 values)
__global__ void pack_header_zero_kernel(unsigned* hdr) {
    if (threadIdx.x < 4) hdr[threadIdx.x] = 0u;
}
__global__ void pack_header_amax_kernel(const float* __restrict__ w, unsigned* hdr, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    spk_wave_amax_commit(idx < total ? fabsf(w[idx]) : 0.f, hdr);
}

extern "C" int spk_pack_conv_weight(const float* w, float* wpk, int Cout, int Cin, int KH, int KW, int transpose,
                                    void* stream) {
    SPK_REQUIRE(w && wpk, "spk_pack_conv_weight: null pointer");
    SPK_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "spk_pack_conv_weight: channels (%d,%d) must be multiples of 32", Cout, Cin);
    SPK_REQUIRE(KH * KW >= 1 && KH * KW <= 9, "spk_pack_conv_weight: kernel %dx%d unsupported", KH, KW);
    const int total = Cout * Cin * KH * KW;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wpk, Cout, Cin, KH * KW, transpose, total);
    SPK_LAUNCH_CHECK("spk_pack_conv_weight");
    return 0;
}

// split: 6 / 9 -> three bf16 terms (numel * 6 bytes), 3 -> 16-byte header + two fp16 terms (16 + numel * 4 bytes)
extern "C" int spk_pack_conv_weight_split(const float* w, void* wpk, int Cout, int Cin, int KH, int KW, int transpose,
                                          int split, void* stream) {
    SPK_REQUIRE(w && wpk, "spk_pack_conv_weight_split: null pointer");
    SPK_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "spk_pack_conv_weight_split: channels (%d,%d) must be multiples of 32", Cout, Cin);
    SPK_REQUIRE(KH * KW >= 1 && KH * KW <= 9, "spk_pack_conv_weight_split: kernel %dx%d unsupported", KH, KW);
    const int total = Cout * Cin * KH * KW;
    if (split == 3) {
        hipLaunchKernelGGL(pack_header_zero_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned*)wpk);
        hipLaunchKernelGGL(pack_header_amax_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                           (unsigned*)wpk, total);
    }
    hipLaunchKernelGGL(pack_conv_weight_split_kernel, dim3(spk_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)wpk, Cout, Cin, KH * KW, transpose, total, split);
    SPK_LAUNCH_CHECK("spk_pack_conv_weight_split");
    return 0;
}

// ---- all convolutions in one launch -------------------------------------------------------------------------------
// Job table entry (device memory, 48 bytes; layout mirrored by ops.py with struct.pack("<QQ8i")).
struct PackJob {
    const float* w;     // OIHW weights
    void* wpk;          // destination
    int Cout, Cin, KHW, transpose;
    int split;          // 0 = fp32 fragment order, 6 / 9 = the three-term bf16 order, 3 = the two-term fp16 order
    int total;          // Cout*Cin*KHW elements = threads of this job
    int block0;         // first block of this job (jobs are ordered; block0 of job i+1 = block0 + ceil(total/256))
    int pad;
};

static __device__ __forceinline__ int pack_job_of_block(const PackJob* __restrict__ jobs, int njobs, int b) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                       // last job whose block0 <= b (uniform per block: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// fp16-split jobs: header word 0 of the destination <- float bits of max|w| (two tiny pre-passes of the batched pack)
__global__ void pack_headers_zero_kernel(const PackJob* __restrict__ jobs, int njobs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < njobs && jobs[i].split == 3) {
        unsigned* h = (unsigned*)jobs[i].wpk;
        h[0] = h[1] = h[2] = h[3] = 0u;
    }
}
__global__ void pack_headers_amax_kernel(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ float red[4];
    const PackJob j = jobs[pack_job_of_block(jobs, njobs, blockIdx.x)];
    if (j.split != 3) return;
    const int idx = (blockIdx.x - j.block0) * 256 + threadIdx.x;
    float v = idx < j.total ? fabsf(j.w[idx]) : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {          // one look at the slot (and at most one atomic) per block
        const unsigned bits = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        unsigned* dst = (unsigned*)j.wpk;
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

__global__ void pack_conv_weights_batched_kernel(const PackJob* __restrict__ jobs, int njobs) {
    const int b = blockIdx.x;
    const PackJob j = jobs[pack_job_of_block(jobs, njobs, b)];
    const int idx = (b - j.block0) * 256 + threadIdx.x;
    if (idx >= j.total) return;
    if (j.split == 3)
        pack_split_elem<2>(j.w, (unsigned short*)j.wpk, j.Cout, j.Cin, j.KHW, j.transpose, idx,
                           spk_sigma_from_amax_bits(*(const unsigned*)j.wpk));
    else if (j.split) pack_split_elem<3>(j.w, (unsigned short*)j.wpk, j.Cout, j.Cin, j.KHW, j.transpose, idx);
    else pack_f32_elem(j.w, (float*)j.wpk, j.Cout, j.Cin, j.KHW, j.transpose, idx);
}

extern "C" int spk_pack_job_bytes(void) { return (int)sizeof(PackJob); }

// has_f16: the table holds fp16-split jobs (split = 3): their absmax headers are refreshed first (two more launches)
extern "C" int spk_pack_conv_weights_batched(const void* jobs, int njobs, int total_blocks, int has_f16, void* stream) {
    SPK_REQUIRE(jobs && njobs > 0 && total_blocks > 0, "spk_pack_conv_weights_batched: bad arguments");
    if (has_f16) {
        hipLaunchKernelGGL(pack_headers_zero_kernel, dim3(spk_ceil_div(njobs, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const PackJob*)jobs, njobs);
        hipLaunchKernelGGL(pack_headers_amax_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                           njobs);
    }
    hipLaunchKernelGGL(pack_conv_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackJob*)jobs, njobs);
    SPK_LAUNCH_CHECK("spk_pack_conv_weights_batched");
    return 0;
}
