"""Kaldi scp/ark reader-writer for the matrix and float-vector tables the hot path touches.

Host-side ingest kept in Python per BASELINE.json's north_star; behaviour follows the reference's
scripts/kaldi_io.py (open_or_fd :41-71, read_key :110, read_mat :376-410, compressed :427-460, write_mat
:464-499, read_vec_flt(_ark) :238-290, write_vec_flt :294-326) but is written from the on-disk formats
(SURVEY.md appendix B), without the reference's import-time PATH mutation / os.popen.
Formats:  "<key> " then either binary "\\0B" + type token, or text "[ ... ]".
  matrix : 'FM ' / 'DM ' : \\x04 int32 rows \\x04 int32 cols, rows*cols floats (row-major);  'CM ' compressed
  vector : 'FV ' / 'DV ' : \\x04 int32 dim, dim floats
scp line : "<key> <path>[:<byte offset of the \\0B>]"
"""
import gzip
import re
import struct

import numpy as np


class UnknownHeader(Exception):
    pass


def open_rx(rx, mode="rb"):
    """Open 'path', 'path.gz', 'ark:path' or 'path:offset' (seeks to the offset). File objects pass through."""
    if not isinstance(rx, str):
        return rx, False
    m = re.match(r"^(ark|scp)(,[a-z,]+)?:", rx)
    if m:
        rx = rx[m.end():]
    offset = None
    m = re.search(r":(\d+)$", rx)
    if m:
        offset = int(m.group(1))
        rx = rx[:m.start()]
    if rx.endswith("|") or rx.startswith("|"):
        raise NotImplementedError("pipe rxfilenames are not supported (no Kaldi binaries in this stack)")
    fd = gzip.open(rx, mode) if rx.endswith(".gz") else open(rx, mode)
    if offset is not None:
        fd.seek(offset)
    return fd, True


def read_key(fd):
    """Next table key (bytes up to the first space); '' at end of file."""
    chars = []
    while True:
        c = fd.read(1)
        if c == b"":
            break
        if c == b" ":
            break
        chars.append(c)
    key = b"".join(chars).decode("latin1").strip()
    return key


def _read_i32(fd):
    size = fd.read(1)
    if size != b"\x04":
        raise UnknownHeader("expected int32 size marker, got %r" % size)
    return struct.unpack("<i", fd.read(4))[0]


def _read_compressed(fd):
    """Kaldi CompressedMatrix ('CM '): global (min, range, rows, cols), per-column 4 x uint16 quantiles,
    uint8 payload stored column-major; piecewise-linear decode."""
    vmin, vrange, rows, cols = struct.unpack("<ffii", fd.read(16))
    hdr = np.frombuffer(fd.read(cols * 8), dtype="<u2").reshape(cols, 4).astype(np.float32)
    hdr = vmin + vrange * 1.52590218966964e-05 * hdr
    data = np.frombuffer(fd.read(cols * rows), dtype=np.uint8).reshape(cols, rows).astype(np.float32)
    p0, p25, p75, p100 = (hdr[:, i:i + 1] for i in range(4))
    out = np.where(data <= 64, p0 + (p25 - p0) * data * (1 / 64.0),
                   np.where(data <= 192, p25 + (p75 - p25) * (data - 64) * (1 / 128.0),
                            p75 + (p100 - p75) * (data - 192) * (1 / 63.0)))
    return np.ascontiguousarray(out.T.astype(np.float32))


def _read_mat_after_flag(fd):
    tok = fd.read(3)
    if tok == b"CM ":
        return _read_compressed(fd)
    if tok == b"FM ":
        dt, sz = "<f4", 4
    elif tok == b"DM ":
        dt, sz = "<f8", 8
    else:
        raise UnknownHeader("matrix header %r" % tok)
    rows = _read_i32(fd)
    cols = _read_i32(fd)
    buf = fd.read(rows * cols * sz)
    return np.frombuffer(buf, dtype=dt).reshape(rows, cols)


def _read_text_mat(fd, first):
    rows = []
    line = first + fd.readline()
    while True:
        toks = line.decode().strip().split()
        done = False
        if toks and toks[0] == "[":
            toks = toks[1:]
        if toks and toks[-1] == "]":
            toks = toks[:-1]
            done = True
        if toks:
            rows.append(np.array(toks, dtype=np.float32))
        if done:
            break
        line = fd.readline()
        if line == b"":
            break
    return np.vstack(rows) if rows else np.zeros((0, 0), np.float32)


def read_mat(rx):
    """Matrix at an scp-style rxfilename ('path:offset') or from an open stream positioned at the flag."""
    fd, own = open_rx(rx)
    try:
        flag = fd.read(2)
        if flag == b"\0B":
            return _read_mat_after_flag(fd)
        return _read_text_mat(fd, flag)
    finally:
        if own:
            fd.close()


def read_mat_ark(rx):
    """Generator of (key, matrix) over an ark stream."""
    fd, own = open_rx(rx)
    try:
        key = read_key(fd)
        while key:
            yield key, read_mat(fd)
            key = read_key(fd)
    finally:
        if own:
            fd.close()


def read_mat_scp(scp):
    for line in open(scp):
        key, rx = line.rstrip().split(None, 1)
        yield key, read_mat(rx)


def write_mat(fd_or_path, m, key=""):
    """Binary 'FM ' / 'DM ' matrix, preceded by '<key> ' when key is given. Returns the byte offset of the
    \\0B flag (what an scp line points at)."""
    fd, own = open_rx(fd_or_path, "wb")
    try:
        if key:
            fd.write((key + " ").encode("latin1"))
        off = fd.tell()
        fd.write(b"\0B")
        if m.dtype == np.float32:
            fd.write(b"FM ")
        elif m.dtype == np.float64:
            fd.write(b"DM ")
        else:
            raise TypeError("write_mat takes float32/float64, got %s" % m.dtype)
        fd.write(b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]))
        fd.write(np.ascontiguousarray(m).tobytes())
        return off
    finally:
        if own:
            fd.close()


def read_vec_flt(rx):
    """Float vector, binary ('FV '/'DV ') or text ('[ v0 v1 ... ]' parsed as float64 like the reference)."""
    fd, own = open_rx(rx)
    try:
        flag = fd.read(2)
        if flag == b"\0B":
            tok = fd.read(3)
            if tok == b"FV ":
                dt, sz = "<f4", 4
            elif tok == b"DV ":
                dt, sz = "<f8", 8
            else:
                raise UnknownHeader("vector header %r" % tok)
            n = _read_i32(fd)
            return np.frombuffer(fd.read(n * sz), dtype=dt)
        toks = (flag + fd.readline()).decode().strip().split()
        toks = [t for t in toks if t not in ("[", "]")]
        return np.array(toks, dtype=float)
    finally:
        if own:
            fd.close()


def read_vec_flt_ark(rx):
    fd, own = open_rx(rx)
    try:
        key = read_key(fd)
        while key:
            yield key, read_vec_flt(fd)
            key = read_key(fd)
    finally:
        if own:
            fd.close()


def write_vec_flt(fd_or_path, v, key=""):
    fd, own = open_rx(fd_or_path, "wb")
    try:
        if key:
            fd.write((key + " ").encode("latin1"))
        fd.write(b"\0B")
        if v.dtype == np.float32:
            fd.write(b"FV ")
        elif v.dtype == np.float64:
            fd.write(b"DV ")
        else:
            raise TypeError("write_vec_flt takes float32/float64, got %s" % v.dtype)
        fd.write(b"\x04" + struct.pack("<i", v.shape[0]))
        fd.write(np.ascontiguousarray(v).tobytes())
    finally:
        if own:
            fd.close()
