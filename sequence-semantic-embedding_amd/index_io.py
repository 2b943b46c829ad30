"""Text <-> numbers for targetEncodingIndex.tsv, byte-identical to the reference's Python loops:
`",".join([str(n) for n in vec])` on numpy.float32 components (sse_index.py:93-95) and
`[float(f) for f in field.split(",")]` (sse_evaluator.py:87, sse_demo.py:87) -- done by the host-side C routines
of the C ABI (csrc/index_io.cpp, multi-threaded; libsse_host.so: no GPU runtime involved)."""
import ctypes as C

import numpy as np

from . import _lib


def format_rows(enc):
    """float32 [n,S] -> list of n strings 'v0,v1,...' exactly as the reference writes them."""
    lib = _lib.load_host_library()
    rows = np.ascontiguousarray(enc, dtype=np.float32)
    if rows.ndim != 2:
        raise ValueError("encodings must be [n,S]")
    n, S = rows.shape
    if n == 0:
        return []
    stride = int(lib.sse_format_rows_stride(S))
    out = np.empty(n * stride, np.uint8)
    lens = np.empty(n, np.int64)
    if lib.sse_format_rows_f32(_lib._ptr(rows), n, S, _lib._ptr(out), _lib._ptr(lens)) != 0:
        raise ValueError("sse_format_rows_f32 failed")
    buf = out.tobytes()
    return [buf[r * stride:r * stride + int(lens[r])].decode("ascii") for r in range(n)]


def parse_rows(fields, S=None):
    """list of n 'v0,v1,...' strings -> float64 [n,S] with Python float() semantics.  Raises ValueError like the
    reference's float() would on a malformed number, or when rows disagree on the number of components."""
    lib = _lib.load_host_library()
    n = len(fields)
    if n == 0:
        return np.zeros((0, S or 0), np.float64)
    if S is None:
        S = fields[0].count(",") + 1
    enc = [f.encode("utf-8") for f in fields]
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum([len(b) for b in enc], out=offsets[1:])
    text = b"".join(enc)
    out = np.empty((n, S), np.float64)
    bad = C.c_int64(-1)
    rc = lib.sse_parse_rows_f64(text, _lib._ptr(offsets), n, S, _lib._ptr(out), C.byref(bad))
    if rc == 2:
        r = int(bad.value)
        vals = [float(f) for f in fields[r].strip().split(",")]      # raises the reference's own ValueError if malformed
        raise ValueError("index row %d has %d components, expected %d" % (r, len(vals), S))
    if rc != 0:
        raise ValueError("sse_parse_rows_f64 failed")
    return out
