"""Optional independent decoders that ship with the base image (NOT part of the reference, NOT part
of this repository): system liblz4 1.9.3 and c-blosc 1.21.  Tests use them when they can be loaded
and skip otherwise; nothing in the product path touches them."""
import ctypes as C

import numpy as np


def _try(paths):
    for p in paths:
        try:
            return C.CDLL(p)
        except OSError:
            continue
    return None


_lz4 = _try(["/usr/lib/x86_64-linux-gnu/liblz4.so.1", "liblz4.so.1"])
_blosc = _try(["/opt/conda/lib/libblosc.so.1", "libblosc.so.1"])

if _lz4 is not None:
    _lz4.LZ4_decompress_safe.restype = C.c_int
    _lz4.LZ4_decompress_safe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    _lz4.LZ4_compress_default.restype = C.c_int
    _lz4.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    _lz4.LZ4_compressBound.restype = C.c_int
    _lz4.LZ4_compressBound.argtypes = [C.c_int]
if _blosc is not None:
    _blosc.blosc_decompress.restype = C.c_int
    _blosc.blosc_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _blosc.blosc_compress_ctx.restype = C.c_int
    _blosc.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    _blosc.blosc_decompress_ctx.restype = C.c_int
    _blosc.blosc_decompress_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


# how often each independent leg actually ran in this session (tests/conftest.py prints it in the terminal summary, so a
# run's log says which legs the parity tests really had — a leg that is absent is otherwise a silent `if`)
used = {"liblz4": 0, "c-blosc": 0}


def have_lz4():
    return _lz4 is not None


def have_blosc():
    return _blosc is not None


def legs():
    """-> {name: (present, what it is)} of the independent decoders the tests use when they can be loaded"""
    from tests.test_h5file import have_h5py
    return {"liblz4": (have_lz4(), "system liblz4 (LZ4_decompress_safe on every LZ4 stream)"),
            "c-blosc": (have_blosc(), "c-blosc 1.x (blosc_decompress_ctx on Blosc-1 chunks; shuffle + header + bstarts)"),
            "libhdf5": (have_h5py(), "h5py / libhdf5 under /opt/conda (the .h5 container, read_direct_chunk, filter pipeline)")}


def lz4_decompress(comp, nbytes):
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    out = np.empty(max(nbytes, 1), np.uint8)
    used["liblz4"] += 1
    n = _lz4.LZ4_decompress_safe(comp.ctypes.data, out.ctypes.data, comp.size, nbytes)
    if n < 0:
        raise RuntimeError(f"LZ4_decompress_safe rc={n}")
    return out[:n]


def lz4_compress(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    cap = _lz4.LZ4_compressBound(buf.size)
    out = np.empty(cap, np.uint8)
    n = _lz4.LZ4_compress_default(buf.ctypes.data, out.ctypes.data, buf.size, cap)
    assert n > 0
    return out[:n].copy()


def blosc1_decompress(chunk, nbytes):
    chunk = np.ascontiguousarray(chunk, dtype=np.uint8)
    out = np.empty(max(nbytes, 1), np.uint8)
    used["c-blosc"] += 1
    n = _blosc.blosc_decompress_ctx(chunk.ctypes.data, out.ctypes.data, nbytes, 1)
    if n < 0:
        raise RuntimeError(f"blosc_decompress_ctx rc={n}")
    return out[:n]


def blosc1_compress(buf, typesize, blocksize=0, clevel=5, shuffle=1, cname=b"lz4"):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    out = np.empty(buf.size + 16 + 4096, np.uint8)
    n = _blosc.blosc_compress_ctx(clevel, shuffle, typesize, buf.size, buf.ctypes.data, out.ctypes.data, out.size,
                                  cname, blocksize, 1)
    assert n > 0
    return out[:n].copy()
