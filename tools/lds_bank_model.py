#!/usr/bin/env python3
"""LDS bank model of the transposed X / dY reads of conv_wgrad_wm16_kernel / conv_wgrad_c32m16_kernel (csrc/conv_wgrad_wm16.hip).

ds_read_b64_tr_b16 is banked per 32-lane half, bank = (byte address / 4) mod 64, 8 bytes per lane; every extra distinct dword on a
busy bank costs one more LDS cycle for that half (MI355X_MICROARCH.md, LDS table).  This file restates the kernel's lane -> address
formulas for both X layouts (WM16_ODD_PITCH = 0 / 1) and counts the cycles of one k-step, so that a layout can be priced on paper
before it is built: the first layout comes out at 3.8 x the conflict-free cycle count over a k-step (measured: SQ_LDS_IDX_ACTIVE /
(SQ_LDS_IDX_ACTIVE - SQ_LDS_BANK_CONFLICT) = 3.4), the odd pitch at 1.1 x for tiles 8 or 16 pixels wide (measured 1.26 over the
step's mix of tiles, which includes the 10 x 6 tiles of the 256-channel layer).

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


STEP_TILES = [("64 ch 40x150", 8, 8, 1, False), ("128 ch 20x75", 4, 16, 1, False), ("256 ch 10x38", 10, 6, 1, False),
              ("32 ch 80x300 (32-channel layout)", 8, 16, 1, True)]

if __name__ == "__main__":
    print("%-36s %8s %8s" % ("tile", "first", "odd pitch"))
    for name, TH, TW, S, c32 in STEP_TILES:
        print("%-36s %8.2f %8.2f" % ("%s  %dx%d" % (name, TH, TW), kstep_ratio(TH, TW, S, False, c32), kstep_ratio(TH, TW, S, True, c32)))
