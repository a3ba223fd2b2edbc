// In-wave pipelined weight gradient of the 3x3 convolutions, f16x3 operands (see conv_wgrad_split.hip for the operand form,
// the transposing fragment reads and the block -> (slice, channel groups) map; conv_kernel.h PIPE for why the staging has to
// ride in the same wave's MFMA shadow: on gfx950 one wave's VALU work does not overlap another wave's MFMAs).
//
// conv_wgrad_split_kernel spends as long converting a region (address arithmetic, fp16 split, LDS writes: ~500 VALU
// instructions per wave) as multiplying it (54 MFMAs), one after the other.  Here the LDS holds TWO regions in a planar image
// ([term][pixel][32 channels fp16] = 64-byte rows: half the bytes of the 192-byte pitch, and four consecutive pixel rows
// are 256 consecutive bytes, so the transposing reads stay conflict-free); while a wave runs the K loop of region i out
// of slot i & 1 it publishes region i + 1 from its prefetch registers into the other slot - one register item behind every
// tap of its first k-step - and refills the registers with region i + 2 behind the taps of its second k-step.  One barrier
// per region.  Items are branch-free (a branch would end the scheduling region): slots that do not exist read a safe address
// and write a dump row.  Same MFMA order per accumulator and the same slabs as conv_wgrad_split_kernel: bit-identical.
#include "conv_wgrad.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static __device__ __forceinline__ s16x8 tr_read8p(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ const float wgp_unit[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#ifndef WGP_VPM
#define WGP_VPM 8      // VALU instructions scheduled behind every MFMA of a tap that carries a staging item
#endif

// NJ = k-steps (16 pixels) per wave and region: the tile has exactly NJ * WK of them, so that the region body is straight-line
// code (a chain of alternatives around the unrolled k-steps makes the compiler spill the 144 accumulator registers)
template <int WK, int WN, int NX, int NJ>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pipe_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    constexpr int NTAPS = 9, KS = 3, ND = WGRAD_ND;
    static_assert(NX + ND <= 9, "one staging item per tap");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int wk = wave / WN, wn = wave % WN;
    // block -> (slice g, ci0, co0): as wgrad_block in conv_wgrad_split.hip
    int g, ci0, co0;
    {
        const int ncgi = a.Cin >> 5, M = ncgi * (a.Cout / (32 * WN));
        const int bid = blockIdx.x;
        int m;
        if ((a.nsplit & 7) == 0) {
            const int k = bid >> 3;
            m = k % M;
            g = (k / M) * 8 + (bid & 7);
        } else {
            m = bid % M;
            g = bid / M;
        }
        ci0 = (m % ncgi) * 32;
        co0 = (m / ncgi) * (32 * WN);
    }
    const int halo_pix = a.halo_h * a.halo_w;
    const int npix = a.TH * a.TW;
    const int nsteps_all = (npix + 15) >> 4;
    const int npix_pad = nsteps_all << 4;
    const int xplane = halo_pix * 64, dplane = npix_pad * 64;            // bytes of one [pixel][32 ch fp16] plane
    const int slot_bytes = 2 * xplane + 2 * WN * dplane;
    unsigned char* dump = ldsb + 2 * slot_bytes;                         // 128 bytes behind the two slots
    const int flags = a.flags;
    const float sig_x = a.x_amax ? spk_sigma_from_amax_bits(*a.x_amax) : SPK_F16_ACT_SIGMA;
    const float sig_d = a.dy_amax ? spk_sigma_from_amax_bits(*a.dy_amax) : 1.f;
    const int nmine = (a.nregions - g + a.nsplit - 1) / a.nsplit;        // regions of this block (>= 1)

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- staging items: u < NX an X halo float4, u >= NX a dY float4 ----
    constexpr int QPP = WN * 8;
    constexpr int PSTEP = 256 / QPP;
    // The items recompute their small per-thread constants (pixel, channel quad, LDS row) from `tv`, a copy of the thread
    // index that is made opaque once per region: otherwise the compiler hoists all of them out of the region loop (nine items x
    // a dozen values) and the kernel needs ~450 registers.
    int tv = tid;
    f32x4 pq[NX + ND];
    unsigned okm = 0;                                                    // bit u: item u of the registers is inside the image / tile
    const bool aff = (flags & SPK_IN_AFFINE_RELU) != 0;
    // BN scale / shift of the X operand are re-read per item (L1-resident) rather than held across the MFMAs
    // (1, 0) without the fused transform: branch-free items.  Global address space spelled out: a select between a kernel argument
    // and a device array is a generic pointer otherwise, and FLAT loads count on the LDS counter too - every LDS wait of the K
    // loop would then wait for them.
    typedef const __attribute__((address_space(1))) float* gfp;
    typedef const __attribute__((address_space(1))) f32x4* gf4p;
    const gfp scp = aff ? (gfp)(a.in_scale + ci0) : (gfp)wgp_unit;
    const gfp shp = aff ? (gfp)(a.in_shift + ci0) : (gfp)(wgp_unit + 32);
    const float floor_x = aff ? 0.f : -__builtin_inff();
    f32x4 scv = *(gf4p)(scp + (tid & 7) * 4), shv = *(gf4p)(shp + (tid & 7) * 4);
    const unsigned x_row = (unsigned)a.IW * a.Cin * 4u, x_px = (unsigned)a.Cin * 4u;
    const unsigned d_row = (unsigned)a.OW * a.Cout * 4u, d_px = (unsigned)a.Cout * 4u;
#define X_C ((unsigned)(ci0 + (tv & 7) * 4) * 4u)
#define D_C ((unsigned)(co0 + (tv % QPP) * 4) * 4u)

    // geometry of a region (scalar)
    struct Reg { const char* xb; const char* db; int oy0, ox0, iy0, ix0; unsigned x_safe, d_safe; };
    auto region_of = [&](int turn) {
        // turn i of this block = region g + i * nsplit; turns past the end re-read the last one (their items go to the dump row)
        const int i = turn < nmine ? turn : nmine - 1;
        int pt = g + i * a.nsplit;
        const int tx = pt % a.tiles_x;
        pt /= a.tiles_x;
        const int ty = pt % a.tiles_y;
        const int b = pt / a.tiles_y;
        Reg r;
        r.oy0 = ty * a.TH;
        r.ox0 = tx * a.TW;
        r.iy0 = r.oy0 * a.S - a.pad;
        r.ix0 = r.ox0 * a.S - a.pad;
        r.xb = (const char*)(a.x + (size_t)b * a.IH * a.IW * a.Cin);
        r.db = (const char*)(a.dy + (size_t)b * a.OH * a.OW * a.Cout);
        r.x_safe = (unsigned)(r.oy0 * a.S) * x_row + (unsigned)(r.ox0 * a.S) * x_px;      // (+ the thread's channel offset)
        r.d_safe = (unsigned)r.oy0 * d_row + (unsigned)r.ox0 * d_px;
        return r;
    };
    auto fetch = [&](const Reg& r, auto uc) {                            // global load of item u (branch-free)
        constexpr int u = decltype(uc)::value;
        if constexpr (u >= NX + ND) return;
        if constexpr (u < NX) {
            const int p = (tv >> 3) + 32 * u;
            const int hy = spk_div20(p, spk_m20(a.halo_w_magic));
            const int hx = p - __mul24(hy, a.halo_w);
            const int iy = r.iy0 + hy, ix = r.ix0 + hx;
            // (& not &&, and the offset formed before the select: no control flow inside an item)
            const bool ok = ((unsigned)iy < (unsigned)a.IH) & ((unsigned)ix < (unsigned)a.IW) & (p < halo_pix);
            okm = (okm & ~(1u << u)) | ((unsigned)ok << u);
            const unsigned off_in = __umul24((unsigned)iy, x_row) + __umul24((unsigned)ix, x_px);
            const unsigned off = (ok ? off_in : r.x_safe) + X_C;
            pq[u] = *(const f32x4*)(r.xb + off);
        } else {
            constexpr int v = u - NX;
            const int p = tv / QPP + PSTEP * v;
            const int ly = spk_div20(p, spk_m20(a.tw_magic));
            const int lx = p - __mul24(ly, a.TW);
            const int oy = r.oy0 + ly, ox = r.ox0 + lx;
            const bool ok = (p < npix) & (oy < a.OH) & (ox < a.OW);
            okm = (okm & ~(1u << u)) | ((unsigned)ok << u);
            const unsigned off_in = __umul24((unsigned)oy, d_row) + __umul24((unsigned)ox, d_px);
            const unsigned off = (ok ? off_in : r.d_safe) + D_C;
            pq[u] = *(const f32x4*)(r.db + off);
        }
    };
    auto publish = [&](unsigned char* slot, bool real, auto uc) {        // registers -> two fp16 terms in the planar LDS image
        constexpr int u = decltype(uc)::value;
        if constexpr (u >= NX + ND) return;
        const bool ok = (okm >> u) & 1u;
        uint2 t0, t1;
        if constexpr (u < NX) {
            const int p = (tv >> 3) + 32 * u, quad = tv & 7;
            f32x4 w = pq[u];
            w = w * scv + shv;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = ok ? fmaxf(w[k], floor_x) : 0.f;
            split2h(w, sig_x, t0, t1);
            const bool wr = real & (p < halo_pix);
            unsigned char* dst_in = slot + p * 64 + quad * 8;
            unsigned char* dst = wr ? dst_in : dump + quad * 8;
            *(uint2*)dst = t0;
            *(uint2*)(dst + (wr ? xplane : 64)) = t1;
        } else {
            constexpr int v = u - NX;
            const int p = tv / QPP + PSTEP * v, cq = tv % QPP;
            f32x4 w = pq[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = ok ? w[k] : 0.f;
            split2h(w, sig_d, t0, t1);
            const bool wr = real & (p < npix_pad);
            unsigned char* dst_in = slot + 2 * xplane + (cq >> 3) * 2 * dplane + p * 64 + (cq & 7) * 8;
            unsigned char* dst = wr ? dst_in : dump + (cq & 7) * 8;
            *(uint2*)dst = t0;
            *(uint2*)(dst + (wr ? dplane : 64)) = t1;
        }
    };

    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int col_off = (g16 & 1) * 32 + p4 * 8;

    // one k-step (16 pixels) of this wave out of `slot`; PUB: publish item t of the next region behind tap t; PRE: refill
    // item t with the region after that
    auto kstep = [&](const unsigned char* slot, int j, unsigned char* nslot, bool nreal, const Reg& r2, auto pubc, auto prec) {
        constexpr bool PUB = decltype(pubc)::value, PRE = decltype(prec)::value;
        const unsigned char* xs = slot;
        const unsigned char* dys = slot + 2 * xplane + wn * 2 * dplane;
        int xa[2], da[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int pix = j * 16 + 8 * h + 4 * blk + q;
            const int pc = pix < npix ? pix : npix - 1;                  // padding rows: any valid X address (their dY is zero)
            const int ly = spk_div20(pc, spk_m20(a.tw_magic));
            const int lx = pc - __mul24(ly, a.TW);
            xa[blk] = (__mul24(__mul24(ly, a.S), a.halo_w) + __mul24(lx, a.S)) * 64 + col_off;
            da[blk] = pix * 64 + col_off;
        }
        s16x8 bf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[s] = tr_read8p(dys + s * dplane + da[0], dys + s * dplane + da[1]);
        s16x8 a0[2], a1[2];
        auto load_a = [&](s16x8* af, int t) {
            const int toff = ((t / KS) * a.halo_w + (t % KS)) * 64;
#pragma unroll
            for (int s = 0; s < 2; ++s) af[s] = tr_read8p(xs + s * xplane + xa[0] + toff, xs + s * xplane + xa[1] + toff);
        };
        auto mma = [&](f32x16& c, const s16x8* af) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[0]), c, 0, 0, 0);
        };
        auto tap = [&](s16x8* acur, s16x8* anext, auto tc) {
            constexpr int t = decltype(tc)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < NTAPS) load_a(anext, t + 1);
            if constexpr (PUB) publish(nslot, nreal, tc);
            if constexpr (PRE) fetch(r2, tc);
            mma(acc[t], acur);
            if constexpr (PUB || PRE) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, WGP_VPM, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#define TC(n) std::integral_constant<int, n>{}
        load_a(a0, 0);
        tap(a0, a1, TC(0)); tap(a1, a0, TC(1)); tap(a0, a1, TC(2)); tap(a1, a0, TC(3)); tap(a0, a1, TC(4));
        tap(a1, a0, TC(5)); tap(a0, a1, TC(6)); tap(a1, a0, TC(7)); tap(a0, a1, TC(8));
    };
#define ALL_ITEMS(F) F(TC(0)) F(TC(1)) F(TC(2)) F(TC(3)) F(TC(4)) F(TC(5)) F(TC(6)) F(TC(7)) F(TC(8))

    // prologue: region 0 into slot 0 the plain way, region 1 into the registers
    {
        const Reg r0 = region_of(0);
#define F(c) fetch(r0, c);
        ALL_ITEMS(F)
#undef F
#define F(c) publish(ldsb, true, c);
        ALL_ITEMS(F)
#undef F
        const Reg r1 = region_of(1);
#define F(c) fetch(r1, c);
        ALL_ITEMS(F)
#undef F
    }
    __syncthreads();
    const std::true_type yes{};
    const std::false_type no{};
    const int j0 = wk, j1 = wk + WK;
    for (int i = 0; i < nmine; ++i) {
        const unsigned char* slot = ldsb + (i & 1) * slot_bytes;
        unsigned char* nslot = ldsb + ((i + 1) & 1) * slot_bytes;
        const bool nreal = i + 1 < nmine;
        const Reg r2 = region_of(i + 2);
        asm volatile("" : "+v"(tv));        // (see above)
        // BN scale / shift of the X operand: fetched once per region, early enough for the first item that uses them
        scv = *(gf4p)(scp + (tv & 7) * 4);
        shv = *(gf4p)(shp + (tv & 7) * 4);
        if constexpr (NJ == 1) {
            kstep(slot, j0, nslot, nreal, r2, yes, yes);
        } else {
            kstep(slot, j0, nslot, nreal, r2, yes, no);
            kstep(slot, j1, nslot, nreal, r2, no, yes);
        }
        __syncthreads();           // slot (i + 1) & 1 is complete, slot i & 1 is free
    }
#undef TC
#undef ALL_ITEMS

    // fold the WK pixel-splits into wk == 0 through LDS (fixed order), then one slab per block
    const int r = lane & 31;
    if (WK > 1) {
        float* red = (float*)ldsb;  // [WN][NTAPS][16][64]
#pragma unroll 1
        for (int src = 1; src < WK; ++src) {
            __syncthreads();
            if (wk == src) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((wn * NTAPS + t) * 16 + e) * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[((wn * NTAPS + t) * 16 + e) * 64 + lane];
            }
        }
    }
    if (wk == 0) {
        float* slab = a.partial + (size_t)g * NTAPS * a.Cin * a.Cout;
        const float inv_x = 1.f / sig_x, inv_d = 1.f / sig_d;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                slab[((size_t)t * a.Cin + ci0 + row) * a.Cout + co0 + wn * 32 + r] = acc[t][e] * inv_x * inv_d;
            }
    }
}

template <int WK, int WN>
static int launch_pipe(const WgradArgs& a, hipStream_t st) {
    const int npix_pad = ((a.TH * a.TW + 15) >> 4) << 4;
    size_t lds_bytes = 2 * (2 * (size_t)a.halo_h * a.halo_w * 64 + 2 * (size_t)WN * npix_pad * 64) + 128;
    const size_t red_bytes = (WK > 1) ? (size_t)WN * 9 * 16 * 64 * sizeof(float) : 0;
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    SPK_REQUIRE(lds_bytes <= 160 * 1024, "spk_conv_wgrad(pipe): tile %dx%d needs %zu B of LDS", a.TH, a.TW, lds_bytes);
    dim3 grid(a.nsplit * (a.Cin / 32) * (a.Cout / (32 * WN)));
    const int nj = (npix_pad >> 4) / WK;
    SPK_REQUIRE((npix_pad >> 4) == nj * WK && (nj == 1 || nj == 2), "spk_conv_wgrad(pipe): tile %dx%d has %d k-steps for %d wave groups (need 1 or 2 each)",
                a.TH, a.TW, npix_pad >> 4, WK);
    const bool small = a.halo_h * a.halo_w <= 32 * 4;
    if (nj == 1) {
        if (small) hipLaunchKernelGGL((conv_wgrad_pipe_kernel<WK, WN, 4, 1>), grid, dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL((conv_wgrad_pipe_kernel<WK, WN, WGRAD_NX, 1>), grid, dim3(256), lds_bytes, st, a);
    } else {
        if (small) hipLaunchKernelGGL((conv_wgrad_pipe_kernel<WK, WN, 4, 2>), grid, dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL((conv_wgrad_pipe_kernel<WK, WN, WGRAD_NX, 2>), grid, dim3(256), lds_bytes, st, a);
    }
    SPK_LAUNCH_CHECK("spk_conv_wgrad(pipe)");
    return 0;
}

int spk_launch_wgrad_pipe(const WgradArgs& a, int WN, hipStream_t st) {
    SPK_REQUIRE(a.KW == 3, "spk_conv_wgrad(pipe): 3x3 only");
    if (WN == 1) return launch_pipe<4, 1>(a, st);
    if (WN == 2) return launch_pipe<2, 2>(a, st);
    return launch_pipe<1, 4>(a, st);
}
