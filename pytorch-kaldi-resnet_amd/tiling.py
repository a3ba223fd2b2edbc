"""Host-side tile selection for the MFMA convolution kernels (pure Python, cached per shape).

A block of spk_conv_mfma covers a TH x TW region of logical output pixels (<= 128*MT, MT m-tiles of 32
pixels per wave, 4 waves) and NT*32 output channels; its input halo tile lives in LDS at 144 B per
pixel.  The chooser minimises a cycle model: MFMA time of the padded region plus staging time of the
halo, with penalties for configurations that cut occupancy (registers at MT*NT >= 8, LDS > 52 KiB).
"""
import os as _os
from functools import lru_cache

LDS_PIX_BYTES = 144
SPLIT_PIX_BYTES = int(_os.environ.get('SPK_SPLIT_PIX_BYTES', '112'))   # LDS bytes per pixel of the bf16-split kernels (csrc/conv_kernel.h)
LDS_SOFT = 52 * 1024      # 3 blocks / CU
LDS_HARD = 80 * 1024      # 2 blocks / CU


# prefetch-window limits of conv_wgrad_kernel (csrc/conv_wgrad.hip: WGRAD_NX = 5, WGRAD_ND = 4; also exported as
# spk_conv_wgrad_limits): the next region must fit in 9 float4 registers per thread
WGRAD_MAX_HALO = 32 * 5
WGRAD_MAX_TILE = {1: 128, 2: 64, 4: 32}


# Measured overrides (tools/conv_bench.py --sweep on MI355X): key -> (TH, TW, MT, NT) / (TH, TW, WN)
FORCE_CONV = {}
FORCE_CONV_SPLIT = {}     # the same launch shapes in the bf16-split operand mode (112 B of LDS per pixel, faster MFMA phase)
FORCE_WGRAD = {}
FORCE_CONV_WS = {}        # wave-specialised kernel (csrc/conv_ws_kernel.h): key -> (TH, TW, MT, NT, WC)
FORCE_WGRAD_C32M16 = {}   # tiles of the 32-channel-group 16x16x32 weight gradient (tests / sweeps; default: wgrad_tile_c32m16's rule)
FORCE_WGRAD_SPLIT = {}    # weight-gradient tiles of the bf16-split kernel (K = 16 pixels per MFMA: tiles of 16 k pixels pad least)


def _load_table():
    import json
    import os
    path = os.environ.get("SPK_TILE_TABLE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_table.json"))
    if os.path.exists(path):
        t = json.load(open(path))
        for k, v in t.get("conv", {}).items():
            FORCE_CONV[tuple(int(x) for x in k.split(","))] = tuple(v)
        for k, v in t.get("conv_split", {}).items():
            FORCE_CONV_SPLIT[tuple(int(x) for x in k.split(","))] = tuple(v)
        for k, v in t.get("conv_ws", {}).items():
            FORCE_CONV_WS[tuple(int(x) for x in k.split(","))] = tuple(v)
        for k, v in t.get("wgrad", {}).items():
            FORCE_WGRAD[tuple(int(x) for x in k.split(","))] = tuple(v)
        for k, v in t.get("wgrad_split", {}).items():
            FORCE_WGRAD_SPLIT[tuple(int(x) for x in k.split(","))] = tuple(v)


# Autotune on first use (the analogue of the reference's `cudnn.benchmark = True`, scripts/train_resnet.py:231):
# when enabled, ops.py times a handful of candidate tiles the first time a launch shape is seen and caches the winner.
AUTOTUNE = _os.environ.get("SPK_AUTOTUNE", "0") == "1"


def conv_candidates(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout, per_config=3, split=0):
    """Candidate (TH, TW, MT, NT) tiles: for every register configuration the best-utilised few shapes."""
    cands = []
    pix_bytes = SPLIT_PIX_BYTES if split else LDS_PIX_BYTES
    for MT in (1, 2, 3, 4):
        cap = 128 * MT
        for NT in (1, 2, 4):
            if Cout % (32 * NT) or MT * NT > 8 or (MT * NT == 8 and MT == 4):
                continue
            if split and (MT, NT) not in ((1, 1), (2, 1), (3, 1), (4, 1), (1, 2), (2, 2), (3, 2), (1, 4)):
                continue
            best = []
            for TH in range(1, min(OH, cap) + 1):
                for TW in range(1, min(OW, cap // TH) + 1):
                    halo = ((TH - 1) * IS + kspan_y) * ((TW - 1) * IS + kspan_x)
                    if halo * pix_bytes > LDS_HARD:
                        continue
                    ty, tx = -(-OH // TH), -(-OW // TW)
                    best.append((-(OH * OW) / (ty * tx * cap), halo / (TH * TW), TH, TW))
            best.sort()
            cands += [(TH, TW, MT, NT) for _, _, TH, TW in best[:per_config]]
    return cands


def wgrad_candidates(OH, OW, Cin, Cout, ksize, stride, per_config=4):
    out = []
    OWe = OW + (OW & 1)
    for WN in ((1,) if Cout == 32 else (1, 2)):
        c = []
        for TH in range(1, OH + 1):
            for TW in range(2, OWe + 1, 2):
                if TH * TW > WGRAD_MAX_TILE[WN] or ((TH - 1) * stride + ksize) * ((TW - 1) * stride + ksize) > WGRAD_MAX_HALO:
                    continue
                ty, tx = -(-OH // TH), -(-OW // TW)
                c.append((ty * tx * (TH * TW + 24.0), TH, TW))
        c.sort()
        out += [(TH, TW, WN) for _, TH, TW in c[:per_config]]
    return out


# wave layouts compiled into conv_ws.hip: (MT, NT, WC); a block covers (4 // WC) * MT * 32 pixels x WC * NT * 32 channels
WS_LAYOUTS = [(2, 1, 1), (4, 1, 1), (3, 2, 1), (3, 1, 2), (6, 1, 2), (3, 1, 4), (6, 1, 4)]
WS_LDS_BYTES = 160 * 1024
# measured in the training step (bench.py instrumented pass, MI355X): the producer/consumer kernel wins on the 64..256-channel
# layers (+5..15 %), loses on the 32-channel layer, whose launches are HBM-bound (one CU-resident workgroup hides less latency)
WS_MIN_COUT = int(_os.environ.get("SPK_WS_MIN_COUT", "64"))


def ws_lds_bytes(TH, TW, IS, kspan_y, kspan_x, NT, pix_bytes=None):
    """two ring slots of the halo tile + one epilogue slab per consumer wave (spk_conv_ws_lds_bytes)"""
    halo = ((TH - 1) * IS + kspan_y) * ((TW - 1) * IS + kspan_x)
    return 2 * halo * (pix_bytes or SPLIT_PIX_BYTES) + 4 * 32 * (NT * 32 + 4) * 4 + 128


def ws_candidates(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout, per_layout=4):
    """Candidate (TH, TW, MT, NT, WC) for the wave-specialised kernel: per compiled wave layout the best-filled tiles."""
    out = []
    for MT, NT, WC in WS_LAYOUTS:
        if Cout % (32 * NT * WC):
            continue
        cap = (4 // WC) * MT * 32
        best = []
        for TH in range(1, min(OH, cap) + 1):
            for TW in range(1, min(OW, cap // TH) + 1):
                if ws_lds_bytes(TH, TW, IS, kspan_y, kspan_x, NT) > WS_LDS_BYTES:
                    continue
                ty, tx = -(-OH // TH), -(-OW // TW)
                halo = ((TH - 1) * IS + kspan_y) * ((TW - 1) * IS + kspan_x)
                best.append((-(OH * OW) / (ty * tx * cap), halo / (TH * TW), TH, TW))
        best.sort()
        out += [(TH, TW, MT, NT, WC) for _, _, TH, TW in best[:per_layout]]
    return out


@lru_cache(maxsize=None)
def _ws_tile(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout):
    """Cost model: MFMA time of the padded tiles + weight-fragment loads (1 / MT per MFMA) + staging of the halo per
    channel group.  Returns None when the map is too small to fill even the smallest layout reasonably."""
    if Cout < WS_MIN_COUT:
        return None
    best = None
    for TH, TW, MT, NT, WC in ws_candidates(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout, per_layout=3):
        cap = (4 // WC) * MT * 32
        ty, tx = -(-OH // TH), -(-OW // TW)
        ncg = Cout // (32 * NT * WC)
        halo = ((TH - 1) * IS + kspan_y) * ((TW - 1) * IS + kspan_x)
        mfma = cap * WC * NT * ntaps                      # per 16-channel chunk, in units of 32x32 tiles x taps
        cost = ty * tx * ncg * (mfma * (1.0 + 0.35 / MT) + halo * 0.9 + 60.0)
        if best is None or cost < best[0]:
            best = (cost, (TH, TW, MT, NT, WC), (OH * OW) / (ty * tx * cap))
    if best is None or best[2] < 0.5:
        return None
    return best[1]


def ws_tile(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout):
    key = (OH, OW, IS, kspan_y, kspan_x, ntaps, Cout)
    if key in FORCE_CONV_WS:
        return FORCE_CONV_WS[key]
    return _ws_tile(*key)


# experiment knob (tile sweeps inside the training step): SPK_FUSED_TILE="OH,OW,Cout:TH,TW,MT,NT" overrides the tile of the fused
# BatchNorm-backward data gradient of that shape
_FUSED_TILE_OVERRIDE = None
if _os.environ.get("SPK_FUSED_TILE"):
    _k, _v = _os.environ["SPK_FUSED_TILE"].split(":")
    _oh, _ow, _co = (int(x) for x in _k.split(","))
    _FUSED_TILE_OVERRIDE = ((_oh, _ow, 1, 3, 3, 9, _co), tuple(int(x) for x in _v.split(",")))


# ... and SPK_PLAIN_TILE="OH,OW,ntaps,Cout:TH,TW,MT,NT" the tile of the plain launches of a shape (any stride)
# (several shapes: entries separated by ';')
_PLAIN_TILE_OVERRIDE = {}
for _e in filter(None, _os.environ.get("SPK_PLAIN_TILE", "").split(";")):
    _k, _v = _e.split(":")
    _PLAIN_TILE_OVERRIDE[tuple(int(x) for x in _k.split(","))] = tuple(int(x) for x in _v.split(","))


def conv_tile(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout, mode=0, split=0):
    """mode 1 = data gradient with the BatchNorm backward fused into its input staging (heavier staging: it may prefer
    wider channel tiles); table keys carry the mode as an 8th element and fall back to the plain entry.  split != 0
    (bf16-split operands) consults its own table first."""
    key = (OH, OW, IS, kspan_y, kspan_x, ntaps, Cout)
    if mode and _FUSED_TILE_OVERRIDE and key == _FUSED_TILE_OVERRIDE[0]:
        return _FUSED_TILE_OVERRIDE[1]
    if not mode and (OH, OW, ntaps, Cout) in _PLAIN_TILE_OVERRIDE:
        return _PLAIN_TILE_OVERRIDE[(OH, OW, ntaps, Cout)]
    if split:
        if mode and key + (mode,) in FORCE_CONV_SPLIT:
            return FORCE_CONV_SPLIT[key + (mode,)]
        if key in FORCE_CONV_SPLIT:
            return FORCE_CONV_SPLIT[key]
    if mode and key + (mode,) in FORCE_CONV:
        return FORCE_CONV[key + (mode,)]
    if key in FORCE_CONV:
        return FORCE_CONV[key]
    return _conv_tile(*key)


# ... and SPK_WGRAD_TILE="OH,OW,Cin,Cout,ksize,stride:TH,TW,WN" the weight-gradient tile of a shape (entries separated by ';')
_WGRAD_TILE_OVERRIDE = {}
for _e in filter(None, _os.environ.get("SPK_WGRAD_TILE", "").split(";")):
    _k, _v = _e.split(":")
    _WGRAD_TILE_OVERRIDE[tuple(int(x) for x in _k.split(","))] = tuple(int(x) for x in _v.split(","))


def wgrad_tile(OH, OW, Cin, Cout, ksize, stride, split=0):
    key = (OH, OW, Cin, Cout, ksize, stride)
    if key in _WGRAD_TILE_OVERRIDE:
        return _WGRAD_TILE_OVERRIDE[key]
    if split and key in FORCE_WGRAD_SPLIT:
        return FORCE_WGRAD_SPLIT[key]
    if key in FORCE_WGRAD:
        return FORCE_WGRAD[key]
    return _wgrad_tile(*key, split=1 if split else 0)


@lru_cache(maxsize=None)
def _conv_tile(OH, OW, IS, kspan_y, kspan_x, ntaps, Cout):
    """-> (TH, TW, MT, NT).  kspan = max tap offset - min tap offset + 1 per axis."""
    NT = 1 if Cout % 64 else 2
    best = None
    for MT in (2, 3, 4, 1):
        cap = 128 * MT
        for TH in range(1, min(OH, cap) + 1):
            tw_max = min(OW, cap // TH)
            if tw_max < 1:
                break
            for TW in range(1, tw_max + 1):
                # only maximal widths for a given tile count are interesting
                tx = -(-OW // TW)
                if TW > 1 and -(-OW // (TW - 1)) == tx and TW != tw_max:
                    pass
                ty = -(-OH // TH)
                halo = ((TH - 1) * IS + kspan_y) * ((TW - 1) * IS + kspan_x)
                lds = halo * LDS_PIX_BYTES
                if lds > LDS_HARD:
                    continue
                cost = ty * tx * (cap * ntaps * NT * 4.5 + halo * 8.0)
                if MT * NT >= 8:
                    cost *= 1.25
                if lds > LDS_SOFT:
                    cost *= 1.08
                if MT == 1:
                    cost *= 1.05
                key = (cost, -TH * TW)
                if best is None or key < best[0]:
                    best = (key, (TH, TW, MT, NT))
    assert best is not None, "no conv tile for %dx%d" % (OH, OW)
    return best[1]


def c32m16_tile(OH, OW, Cin, Cout, ksize, stride):
    """tile of conv_wgrad_wm16_kernel<., C32> for a shape: SPK_WGRAD_TILE override, then FORCE_WGRAD_C32M16, then the rule below"""
    key = (OH, OW, Cin, Cout, ksize, stride)
    if key in _WGRAD_TILE_OVERRIDE:
        return _WGRAD_TILE_OVERRIDE[key]
    if key in FORCE_WGRAD_C32M16:
        return FORCE_WGRAD_C32M16[key]
    return wgrad_tile_c32m16(OH, OW, ksize, stride)


@lru_cache(maxsize=None)
def wgrad_tile_c32m16(OH, OW, ksize, stride):
    """Tile (TH, TW, WN = 1) for conv_wgrad_wm16_kernel<., C32> (csrc/conv_wgrad_wm16.hip): k-steps of 32 pixels dealt to four waves, so a
    region costs ceil(steps / 4) step times; halo <= 192 pixels (six 32-pixel staging passes); X and two dY buffers of 128 B per
    pixel within 80 KB (two blocks per CU; X priced at 192 B per pixel - the kernel's odd pitch is 160 B, the bound stays on the safe side).  Ties go to the wider tile (longer contiguous runs for the DMA chunks of 8 pixels)."""
    best = None
    for TH in range(1, OH + 1):
        for TW in range(2, OW + 2, 2):
            npix = TH * TW
            if npix > 256:
                break
            halo = ((TH - 1) * stride + ksize) * ((TW - 1) * stride + ksize)
            steps = -(-npix // 32)
            if halo > 192 or halo * 192 + 2 * steps * 32 * 128 > 80 * 1024:
                continue
            ty, tx = -(-OH // TH), -(-OW // TW)
            cost = ty * tx * (-(-steps // 4) * 128 + 24.0 + 0.1 * halo)
            key = (cost, -TW)
            if best is None or key < best[0]:
                best = (key, (TH, TW, 1))
    return best[1] if best else None


@lru_cache(maxsize=None)
def _wgrad_tile(OH, OW, Cin, Cout, ksize, stride, split=0):
    """-> (TH, TW, WN).  TW even; halo <= WGRAD_MAX_HALO pixels, tile <= WGRAD_MAX_TILE[WN] pixels.  The split kernel
    consumes 16 pixels per MFMA step: its cost counts the tile padded to a multiple of 16."""
    WN = 1 if Cout == 32 else 2
    best = None
    OWe = OW + (OW & 1)
    for TH in range(1, OH + 1):
        for TW in range(2, OWe + 1, 2):
            if TH * TW > WGRAD_MAX_TILE[WN]:
                break
            halo = ((TH - 1) * stride + ksize) * ((TW - 1) * stride + ksize)
            if halo > WGRAD_MAX_HALO:
                continue
            ty, tx = -(-OH // TH), -(-OW // TW)
            # MFMA work ~ padded pixels; every region also pays two barriers and a pipeline fill (~24 pixel-equivalents)
            pix = -(-(TH * TW) // 16) * 16 if split else TH * TW
            cost = ty * tx * (pix + 24.0 + 0.1 * halo)
            key = (cost, -TH * TW)
            if best is None or key < best[0]:
                best = (key, (TH, TW, WN))
    assert best is not None
    return best[1]


WGRAD_TARGET_BLOCKS = int(_os.environ.get('SPK_WGRAD_BLOCKS', '512'))      # persistent wgrad blocks per launch = 2 per CU x 256 CUs (measured best of 512/768/1024)


def wgrad_nsplit(nregions, Cin, Cout, WN, target_blocks=None, cin_groups=1):
    per = (Cin // (32 * cin_groups)) * (Cout // (32 * WN))
    return max(1, min(nregions, (target_blocks or WGRAD_TARGET_BLOCKS) // per))


_load_table()
