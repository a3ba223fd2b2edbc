"""ctypes binding of libspkhip.so (include/spkhip.h).  Fails loudly when the library is missing:
there is no CPU or eager-PyTorch fallback anywhere in this package."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPK_LIB", os.path.join(_HERE, "libspkhip.so"))   # SPK_LIB: A/B builds of the same ABI

IN_AFFINE_RELU, EPI_AFFINE, EPI_ADD, EPI_RELU, EPI_STATS, EPI_BNBWD, IN_BNBWD, CONV_WS = 1, 2, 4, 8, 16, 32, 64, 128
CONV_PIPE = 1024
WGRAD_GROUPS = 2048        # spk_conv_wgrad flags: 1x1 f16x3 kernel with 1 << (bits 12-13) input-channel groups per block
IN_PRESPLIT, SIDE_PRESPLIT, DY_PRESPLIT = 1 << 14, 1 << 15, 1 << 16     # f16 pair tensors (include/spkhip.h)
CONV_M16 = 1 << 17      # 16x16x32 form of the pipelined convolution
WGRAD_NOSHIFT = 1 << 18  # 3x3 grouped weight gradient: plain K loop instead of the shifted-window form (A/B)
WGRAD_M16 = 1 << 19      # 3x3 grouped weight gradient on 16x16x32 with dy (a pair tensor) staged by LDS DMA
MASK_NONE, MASK_ACT, MASK_RAW, MASK_BITS = 0, 1, 2, 3

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_longlong
_F = ctypes.c_float
_D = ctypes.c_double
_IP = ctypes.POINTER(ctypes.c_int)

_SIGS = {
    "spk_pack_conv_weight": [_P, _P, _I, _I, _I, _I, _I, _P],
    "spk_pack_conv_weight_split": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "spk_pack_job_bytes": [],
    "spk_build_flags": [],
    "spk_pack_conv_weights_batched": [_P, _I, _I, _I, _P],
    "spk_conv_mfma": [_P] * 21 + [_I] * 14 + [_IP, _IP, _IP] + [_I] * 8 + [_P, _P, _P, _P],
    "spk_conv3x3_c32_stream": [_P] * 6 + [_I] * 4 + [_P, _P, _I, _P],
    "spk_conv1x1_stream": [_P] * 11 + [_L, _I, _I, _P, _P, _I, _P],
    "spk_conv1x1_stream_rows": [_I, _I],
    "spk_conv_wgrad": [_P] * 6 + [_I] * 16 + [_P, _P, _P],
    "spk_conv_wgrad_limits": [_I, _IP, _IP],
    "spk_wgrad_reduce": [_P, _P, _I, _I, _I, _I, _I, _P],
    "spk_stem_fwd_blocks": [_I, _I, _I],
    "spk_stem_conv_fwd": [_P] * 6 + [_I] * 4 + [_P, _P],
    "spk_stem_wgrad_blocks": [_I, _I, _I],
    "spk_stem_conv_wgrad": [_P] * 4 + [_I] * 4 + [_P],
    "spk_bn_stats_blocks": [_L, _I],
    "spk_bn_stats_partial": [_P, _P, _L, _I, _P],
    "spk_bn_finalize": [_P, _I, _I, _D] + [_P] * 9 + [_F, _F, _P, _P, _P, _P],
    "spk_bn_eval_coeffs": [_P] * 6 + [_I, _F, _P],
    "spk_bn_apply": [_P] * 8 + [_L, _I, _I, _P, _P],
    "spk_bn_bwd_reduce": [_P] * 8 + [_L, _I, _I, _P, _P],
    "spk_bn_bwd_finalize": [_P, _I, _I, _D] + [_P] * 5 + [_I, _P, _P, _P, _P, _P, _P, _P],
    "spk_bn_bwd_apply": [_P] * 10 + [_L, _I, _I, _P, _P, _P],
    "spk_absmax": [_P, _P, _L, _P],
    "spk_bnbwd_estimate": [_P, _P, _P, _I, _P, _P, _P, _P],
    "spk_f16_window_count": [_P, _P, _P, _L, _I, _P, _I, _P, _P],
    "spk_affine_estimate": [_P, _P, _I, _P, _P, _P],
    "spk_stats_pool_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "spk_stats_pool_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "spk_gemm_f32": [_P] * 4 + [_I] * 3 + [_L] * 5 + [_F, _I, _P, _P],
    "spk_gemm_splitk": [_I, _I, _I],
    "spk_colsum": [_P, _P, _I, _I, _I, _P],
    "spk_l2norm_fwd": [_P, _P, _P, _I, _I, _F, _P],
    "spk_l2norm_bwd": [_P, _P, _P, _P, _I, _I, _F, _I, _P],
    "spk_aam_margin_fwd": [_P, _P, _P, _I, _I, _F, _F, _P],
    "spk_aam_margin_bwd": [_P, _P, _P, _P, _I, _I, _F, _F, _P],
    "spk_softmax_ce": [_P] * 5 + [_I, _I, _F, _P],
    "spk_mean": [_P, _P, _I, _P],
    "spk_relu_bwd": [_P, _P, _P, _L, _P],
    "spk_sgd_step": [_P, _P, _P, _L, _F, _F, _F, _F, _I, _P],
    "spk_center_normalize": [_P, _P, _P, _I, _I, _F, _P],
    "spk_trial_cosine": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "spk_topk_mean_std": [_P, _P, _P, _I, _I, _I, _L, _P],
}

_lib = None


def lib():
    """Load libspkhip.so (built by build.py / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and "SPK_LIB" not in os.environ:
            # not built yet (fresh checkout): compile it now with hipcc; there is no other implementation to fall back to
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("spk_build", os.path.join(_HERE, "build.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.build(force=True, verbose=False)
            except Exception as e:
                raise RuntimeError("libspkhip.so is missing at %s and building it failed (%s). Run `python "
                                   "__graft_entry__.py build` (hipcc --offload-arch=gfx950). There is no fallback path."
                                   % (LIB_PATH, e))
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libspkhip.so is missing at %s: run `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        l.spk_last_error.restype = ctypes.c_char_p
        l.spk_version.restype = ctypes.c_int
        l.spk_conv_wgrad_workspace.restype = ctypes.c_size_t
        l.spk_conv_wgrad_workspace.argtypes = [_I, _I, _I, _I]
        l.spk_bn_finalize_workspace.restype = ctypes.c_size_t
        l.spk_bn_finalize_workspace.argtypes = [_I, _I]
        l.spk_gemm_workspace.restype = ctypes.c_size_t
        l.spk_gemm_workspace.argtypes = [_I, _I, _I]
        for name, sig in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        _lib = l
    return _lib


def has_experimental():
    """the library was built with SPK_EXPERIMENTAL=1: the measured, not-faster kernel forms are present"""
    return bool(lib().spk_build_flags() & 1)


def exported_symbols():
    return ["spk_version", "spk_last_error", "spk_conv_wgrad_workspace", "spk_bn_finalize_workspace", "spk_gemm_workspace"] + list(_SIGS)


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Requires a contiguous fp32/int64/int32 CUDA(HIP) tensor."""
    if t is None:
        return None
    assert t.is_cuda, "libspkhip takes device tensors only"
    assert t.is_contiguous(), "libspkhip takes contiguous tensors only"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, name):
    if rc != 0:
        msg = lib().spk_last_error().decode()
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, msg))


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    check(rc, name)
