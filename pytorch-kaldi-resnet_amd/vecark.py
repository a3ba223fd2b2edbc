"""Vector arks as matrices (libspkio, no torch import): the embedding file scripts/decode.py writes - text 'key [ v0 ... ]'
lines or binary FV records - parsed natively into one [n][D] float64 matrix.

The reference reads it back one Python-level parse per utterance (scripts/kaldi_io.py:238-290 through
scripts/compute_mean.py:9-33 and scripts/cosine_score.py:52-60): seconds per 100 k utterances before a score is computed.
Values are float64 as numpy parses them, so the reference's "subtract the mean in float64, then cast to float32" is kept."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libspkio.so")
        if not os.path.exists(path):
            raise RuntimeError("libspkio.so is missing at %s: run `python __graft_entry__.py build`" % path)
        l = ctypes.CDLL(path)
        l.spk_io_last_error.restype = ctypes.c_char_p
        l.spk_vec_ark_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32),
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_double)), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_int64)]
        l.spk_vec_ark_free.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        _LIB = l
    return _LIB


class EmbTable(dict):
    """utt -> float64 vector (a row view of .mat), like the dict kaldi_io.read_vec_flt_ark builds, plus the matrix itself
    (.mat [n][D] float64, .keys_list in file order, .index utt -> row): what the vectorised scoring paths work on.
    A key that occurs twice keeps its LAST vector in the dict (as a dict built from the reader does); .mat keeps every row."""

    def __init__(self, keys, mat):
        super().__init__()
        self.keys_list, self.mat = keys, mat
        self.index = {}
        for i, k in enumerate(keys):
            self.index[k] = i
            self[k] = mat[i]


def load(path, nthreads=8):
    """-> EmbTable of a text or binary vector ark"""
    l = _lib()
    n, D = ctypes.c_int64(), ctypes.c_int32()
    data, keys, kb = ctypes.POINTER(ctypes.c_double)(), ctypes.c_void_p(), ctypes.c_int64()
    rc = l.spk_vec_ark_load(os.fsencode(path), int(nthreads), ctypes.byref(n), ctypes.byref(D), ctypes.byref(data), ctypes.byref(keys),
                            ctypes.byref(kb))
    if rc != 0:
        raise RuntimeError("spk_vec_ark_load(%s) failed (rc=%d): %s" % (path, rc, l.spk_io_last_error().decode()))
    try:
        if n.value == 0:
            return EmbTable([], np.zeros((0, 0)))
        mat = np.ctypeslib.as_array(data, shape=(n.value, D.value)).copy()
        raw = ctypes.string_at(keys.value, kb.value)
        names = raw[:-1].decode().split("\0")
    finally:
        l.spk_vec_ark_free(data, keys)
    assert len(names) == n.value
    return EmbTable(names, mat)
