#!/usr/bin/env python3
"""LDS bank model of the transposed X / dY reads of conv_wgrad_wm16_kernel / conv_wgrad_c32m16_kernel (csrc/conv_wgrad_wm16.hip).

ds_read_b64_tr_b16 is banked per 32-lane half, bank = (byte address / 4) mod 64, 8 bytes per lane; every extra distinct dword on a
busy bank costs one more LDS cycle for that half (MI355X_MICROARCH.md, LDS table).  This file restates the kernel's lane -> address
formulas for both X layouts (WM16_ODD_PITCH = 0 / 1) and counts the cycles of one k-step, so that a layout can be priced on paper
before it is built: the first layout comes out at 3.8 x the conflict-free cycle count over a k-step (measured: SQ_LDS_IDX_ACTIVE /
(SQ_LDS_IDX_ACTIVE - SQ_LDS_BANK_CONFLICT) = 3.4), the odd pitch at 1.1 x for tiles 8 or 16 pixels wide (measured 1.26 over the
step's mix of tiles, which includes the 10 x 6 tiles of the 256-channel layer).

The second half does the same for conv_pipe_kernel's 16x16x32 form (A-fragment ds_read_b128, staging ds_write2_b64, epilogue slab):
instruction count, LDS-array cycles and conflict cycles per launch against the measured counters, and what a padded halo row buys.

    python tools/lds_bank_model.py            # table over the step's tiles
"""


def half_cycles(addrs):
    """LDS cycles of one 32-lane half: the deepest bank (distinct dwords per bank; a lane reads two consecutive dwords)"""
    banks = {}
    for a in addrs:
        assert a % 8 == 0
        for dw in (a // 4, a // 4 + 1):
            banks.setdefault(dw % 64, set()).add(dw)
    return max(len(v) for v in banks.values())


def pixel_of(kg, blk, q, odd):
    """pixel of the k-step that lane group kg (0..3) holds in its element q of block blk (conv_wgrad_wm16.hip, kstep)"""
    return (16 * (kg >> 1) + 8 * blk + 4 * (kg & 1) + q) if odd else (8 * kg + 4 * blk + q)


def x_read_cycles(TH, TW, stride, odd, c32, j=0, tap=(0, 0)):
    """cycles (both halves) of the two transposed reads (blk 0, 1) that fetch one A fragment of one tap in k-step j"""
    PX = (160 if c32 else 416) if odd else (192 if c32 else 384)
    halo_w = (TW - 1) * stride + 3
    npix = TH * TW
    total = 0
    for blk in range(2):
        for half in range(2):
            addrs = []
            for lane in range(32 * half, 32 * half + 32):
                kg, q, p4 = lane >> 4, (lane & 15) >> 2, lane & 3
                pix = min(j * 32 + pixel_of(kg, blk, q, odd), npix - 1)
                ly, lx = divmod(pix, TW)
                h = (ly * stride + tap[0]) * halo_w + lx * stride + tap[1]
                addrs.append(h * PX + p4 * 8)
            total += half_cycles(addrs)
    return total


def dy_read_cycles(odd, c32):
    """cycles (both halves) of the two transposed reads of one B fragment (the dY image is written by the DMA path: fixed slots)"""
    total = 0
    for blk in range(2):
        for half in range(2):
            addrs = []
            for lane in range(32 * half, 32 * half + 32):
                kg, q, p4 = lane >> 4, (lane & 15) >> 2, lane & 3
                if odd:
                    base = (2 * (kg >> 1) * 1024 + (kg & 1) * 512) if c32 else (4 * (kg >> 1) + (kg & 1)) * 1024
                    dblk = 1024 if c32 else 2048
                else:
                    base = (kg if c32 else 2 * kg) * 1024
                    dblk = 512 if c32 else 1024
                addrs.append(base + blk * dblk + p4 * 64 + q * 16)
            total += half_cycles(addrs)
    return total


def kstep_ratio(TH, TW, stride, odd, c32):
    """LDS read cycles of a tile's k-steps over the conflict-free count (9 taps x 2 terms x 2 row tiles of X, 2 x 2 fragments of dY)"""
    nsteps = -(-(TH * TW) // 32)
    cyc = ideal = 0
    for j in range(nsteps):
        for t in range(9):
            cyc += 4 * x_read_cycles(TH, TW, stride, odd, c32, j, (t // 3, t % 3))
            ideal += 4 * 4
        cyc += 4 * dy_read_cycles(odd, c32)
        ideal += 4 * 4
    return cyc / ideal


# ---------------------------------------------------------------------------------------------------------------------------------
# conv_pipe_kernel<3, 2, ., ., ., M16 = true> (csrc/conv_kernel.h, the 16x16x32 form): the A-fragment reads (ds_read_b128), the
# staging stores (two 8-byte pieces per lane, 16 bytes apart: one ds_write2_b64) and the epilogue slab (ds_write_b32 / ds_read_b128).
# Bank rules from MI355X_MICROARCH.md (LDS table): ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}
# (+32), bank = (a/4) mod 64 -> a lane covers one 16-byte granule, 16 granules per cycle; ds_write_b64 in four groups of 16 contiguous
# lanes, bank = (a/4) mod 32; ds_write_b32 in two halves, mod 32.
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
PIPE_LP4 = 5          # ConvCfg<3>::LP4: pixel pitch in 16-byte granules ([8-channel half][term][8 fp16] + 16 bytes of pad)


def pix_of_row16(row):
    """row of a 16 x 16 tile -> pixel of its 16-pixel group (conv_body, pix_of_row16)"""
    q = (row + 12) & 15
    return ((q & 7) << 1) | (q >> 3)


def _deepest(keys, nbanks):
    banks = {}
    for k in keys:
        banks.setdefault(k % nbanks, set()).add(k)
    return max(len(v) for v in banks.values())


def pipe_a_read_cycles(TH, TW, IS, i16, row_pad=0):
    """LDS cycles of one A-fragment read of row tile i16 (16 consecutive tile pixels; lanes 32-63 read another tap of the same
    pixels: a uniform shift inside their groups).  row_pad: extra granules per halo row (0 = the kernel as it is)."""
    halo_w = (TW - 1) * IS + 3
    pitch = halo_w * PIPE_LP4 + row_pad
    npix = TH * TW
    total = 0
    for grp in B128_GROUPS:
        g = []
        for l in grp:
            q = i16 * 16 + pix_of_row16(l & 15)
            ly, lx = divmod(q if q < npix else 0, TW)
            g.append(ly * IS * pitch + lx * IS * PIPE_LP4 + 2 * ((l >> 4) & 1))       # lbase[] of conv_body, M16
        total += _deepest(g, 16)
    return total


def pipe_a_read_ratio(TH, TW, IS=1, MT=3, row_pad=0):
    """A-read cycles of a block's 4 waves x 2 MT row tiles over the conflict-free count"""
    n = 4 * 2 * MT
    return sum(pipe_a_read_cycles(TH, TW, IS, i, row_pad) for i in range(n)) / (4.0 * n)


def pipe_stage_write_cycles():
    """LDS-array cycles of one staging store of a wave (ds_write2_b64 = two 8-byte accesses per lane): lane = (pixel slot, float4
    quad qd) writes term 0 at pixel * 80 + (qd >> 1) * 32 + (qd & 1) * 8 and term 1 16 bytes further (store_pair_q, M16)"""
    total = 0
    for term in range(2):
        for grp in range(4):
            dws = []
            for lane in range(16 * grp, 16 * grp + 16):
                a = (lane >> 2) * (PIPE_LP4 * 16) + ((lane & 3) >> 1) * 32 + (lane & 1) * 8 + term * 16
                dws += [a // 4, a // 4 + 1]
            total += _deepest(dws, 32)
    return total


# the 3x3 stride-1 convolutions of the ResNet-34 step that run on this kernel: (layer, launches per direction, TH, TW, Cin, blocks)
PIPE_LAUNCHES = [("64 ch 40x150", 7, 20, 19, 64, 256 * 2 * 8), ("128 ch 20x75", 11, 20, 19, 128, 256 * 1 * 4 * 2),
                 ("256 ch 10x38", 5, 10, 38, 256, 256 * 1 * 1 * 4)]


def pipe_launch_model(row_pad=0):
    """per launch, averaged over the 23 launches of a direction: (LDS wave-instructions, LDS-array cycles, conflict cycles, of which
    A reads).  Per block and 16-channel plane: 9 half-step pairs x 6 row tiles x 2 terms / 2 = 54 ds_read_b128 per wave, 8 staging
    items = 8 ds_write2_b64 per wave (the ISA of the built kernel agrees: 108 ds_read_b128 + 16 ds_write2_b64 per plane pair, llvm-objdump
    of build/conv_pipe.o); epilogue per wave: 3 m-tiles x (32 4-byte slab stores - the compiler pairs them into 48 ds_write2_b32 - and
    8 ds_read_b128), the stores two deep on both halves (rows 9 / 0 of the slab, 68-float pitch: counted, but a two-deep 4-byte
    store costs no time).  Not modelled: the 46-70 ds_bpermute_b32 per wave of the statistics / absmax reductions."""
    n = sum(k for _, k, *_ in PIPE_LAUNCHES)
    insts = cycles = conflicts = reads_conf = 0.0
    w_cyc = pipe_stage_write_cycles()
    for _, k, TH, TW, cin, blocks in PIPE_LAUNCHES:
        planes = cin // 16
        r = pipe_a_read_ratio(TH, TW, 1, 3, row_pad)
        rd = 4 * 54 * planes * blocks
        wr = 4 * 8 * planes * blocks
        ew, er = 4 * 96 * blocks, 4 * 24 * blocks
        insts += k * (rd + wr + ew + er)
        cycles += k * (rd * 4 * r + wr * w_cyc + ew * 4 + er * 4)
        conflicts += k * (rd * 4 * (r - 1) + wr * (w_cyc - 8) + ew * 2)
        reads_conf += k * rd * 4 * (r - 1)
    return insts / n, cycles / n, conflicts / n, reads_conf / n


def pipe_row_pad_consistent(TH, TW, IS, row_pad):
    """Address arithmetic of tools/patches/pipe_row_pad.patch, emulated: the staging store of (halo pixel, 8-channel half, term) and
    the A-fragment read of (tile pixel, tap, half, term) must meet on the same 16-byte granule, every granule must hold one item,
    and everything must stay inside the slot.  -> bytes of one LDS slot."""
    halo_w, halo_h = (TW - 1) * IS + 3, (TH - 1) * IS + 3
    magic = (0x100000000 + halo_w - 1) // halo_w                       # ConvArgs::halo_w_magic
    plane4 = halo_h * halo_w * PIPE_LP4 + halo_h * row_pad
    written = {}
    for p in range(halo_h * halo_w):
        hy = (p * magic) >> 32                                         # __umulhi(p, halo_w_magic)
        assert hy == p // halo_w
        for qd in range(4):
            for term in range(2):
                u2 = p * (PIPE_LP4 * 2) + (qd >> 1) * 4 + (qd & 1) + hy * (row_pad * 2) + 2 * term        # store_pair_q (uint2 units)
                assert u2 // 2 < plane4
                written.setdefault(u2 // 2, set()).add((p, qd >> 1, term))
    assert all(len(v) == 1 for v in written.values())
    for q in range(TH * TW):
        ly, lx = divmod(q, TW)
        for half in range(2):
            lbase = ((ly * IS) * halo_w + lx * IS) * PIPE_LP4 + 2 * half + ly * IS * row_pad
            for dy in range(3):
                for dx in range(3):
                    tap_off = (dy * halo_w + dx) * PIPE_LP4 + dy * row_pad                                  # launcher
                    for term in range(2):
                        assert written.get(lbase + tap_off + term) == {((ly * IS + dy) * halo_w + lx * IS + dx, half, term)}
    return plane4 * 16


STEP_TILES = [("64 ch 40x150", 8, 8, 1, False), ("128 ch 20x75", 4, 16, 1, False), ("256 ch 10x38", 10, 6, 1, False),
              ("32 ch 80x300 (32-channel layout)", 8, 16, 1, True)]

if __name__ == "__main__":
    print("%-36s %8s %8s" % ("tile", "first", "odd pitch"))
    for name, TH, TW, S, c32 in STEP_TILES:
        print("%-36s %8.2f %8.2f" % ("%s  %dx%d" % (name, TH, TW), kstep_ratio(TH, TW, S, False, c32), kstep_ratio(TH, TW, S, True, c32)))
    print()
    print("conv_pipe_kernel (16x16x32 form): A-read cycles over the conflict-free count, by extra 16-byte granules per halo row")
    for name, k, TH, TW, cin, blocks in PIPE_LAUNCHES:
        print("%-16s %dx%d  " % (name, TH, TW) + " ".join("%d:%.2f" % (p, pipe_a_read_ratio(TH, TW, 1, 3, p)) for p in (0, 2, 4, 6, 8, 14)))
    print("staging store (ds_write2_b64): %d LDS-array cycles per wave-instruction (8 conflict-free)" % pipe_stage_write_cycles())
    for pad in (0, 6):
        i, c, x, rx = pipe_launch_model(pad)
        print("row_pad %d: per launch %.3e LDS instructions, %.3e LDS cycles, %.3e conflict cycles (%.3e in the A reads)" % (pad, i, c, x, rx))
    print("measured (profiles/r04_sq_counters/final_kernels_after_odd_pitch_lds_vmem.md, forward launches): 5.288e6 / 4.007e7 / 1.654e7")
