"""ctypes binding of the host-side reader (include/hhgt_reader.h): .vcf / .vcf.gz (gzip or BGZF) ->
line-aligned blocks in pinned memory -> hipMemcpyAsync.  Replaces the htslib calls behind
/root/reference/cpp/vcfpp.h:1378-1385,1468 (hts_open, bcf_hdr_read, tbx_itr_next)."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib
from ._lib import check

_bound = False


def _lib_reader():
    global _bound
    L = _lib.load()
    if not _bound:
        vp, u64 = C.c_void_p, C.c_uint64
        L.hhgt_reader_open.argtypes = [C.c_char_p, u64, C.c_int, C.c_int, C.POINTER(vp)]
        L.hhgt_reader_close.argtypes = [vp]
        L.hhgt_reader_close.restype = None
        L.hhgt_reader_is_bgzf.argtypes = [vp]
        L.hhgt_reader_next.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.hhgt_reader_copy_async.argtypes = [vp, vp, u64, vp, vp]
        L.hhgt_reader_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        _bound = True
    return L


class VcfReader:
    """Iterates over blocks of whole lines.  Each block is a numpy uint8 view of the reader's pinned
    buffer, valid until the next block is requested."""

    def __init__(self, path, block_bytes=64 << 20, n_threads=0, n_blocks=3):
        self.L = _lib_reader()
        h = C.c_void_p()
        check(self.L.hhgt_reader_open(str(path).encode(), int(block_bytes), int(n_threads), int(n_blocks), C.byref(h)))
        self.h = h
        self.path = str(path)
        self._ptr = None

    @property
    def is_bgzf(self):
        return bool(self.L.hhgt_reader_is_bgzf(self.h))

    def next_block(self):
        p, n = C.c_void_p(), C.c_uint64(0)
        check(self.L.hhgt_reader_next(self.h, C.byref(p), C.byref(n)))
        if n.value == 0:
            self._ptr = None
            return None
        self._ptr = p
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    def copy_async(self, block, d_dst_ptr, stream_ptr):
        check(self.L.hhgt_reader_copy_async(self.h, C.c_void_p(block.ctypes.data), block.size,
                                            C.c_void_p(d_dst_ptr), C.c_void_p(stream_ptr)))

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self.L.hhgt_reader_stats(self.h, C.byref(a), C.byref(b)))
        return dict(file_bytes=a.value, text_bytes=b.value)

    def __iter__(self):
        while True:
            b = self.next_block()
            if b is None:
                return
            yield b

    def close(self):
        if getattr(self, "h", None):
            self.L.hhgt_reader_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_header(block):
    """sample names from the '#CHROM' line (cpp/vcfpp.h:1383 bcf_hdr_read; :369-378 sample lookup).
    -> (names, header_nbytes) ; header_nbytes = offset of the first data line inside `block`."""
    buf = bytes(block[: min(block.size, 64 << 20)])
    pos = 0
    names = None
    while pos < len(buf):
        if buf[pos:pos + 1] != b"#":
            break
        nl = buf.find(b"\n", pos)
        end = nl if nl >= 0 else len(buf)
        line = buf[pos:end].rstrip(b"\r")
        if line.startswith(b"#CHROM"):
            names = [x.decode() for x in line.split(b"\t")[9:]]
        pos = end + 1
    if names is None:
        raise _lib.HhgtError(-4, "no #CHROM header line in the first block of the VCF")
    return names, min(pos, len(buf))


def write_bgzf_native(path, data, level=6, n_threads=0):
    """the same file as write_bgzf, members deflated in parallel by libhhgt (`hhgt_synth_write_bgzf`; bench tooling).
    data: bytes-like or a uint8 numpy array (not copied)"""
    L = _lib.load()
    L.hhgt_synth_write_bgzf.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]
    a = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    check(L.hhgt_synth_write_bgzf(str(path).encode(), C.c_void_p(a.ctypes.data), a.size, int(level), int(n_threads)))


def write_bgzf(path, data, block_size=0xFF00, level=6):
    """Minimal BGZF writer (test/bench tooling: BASELINE.md §3 asks for BGZF-compressed synthetic
    shards; htslib/bgzip are not in the image)."""
    data = bytes(data)
    with open(path, "wb") as f:
        for i in list(range(0, len(data), block_size)) + [None]:
            chunk = b"" if i is None else data[i:i + block_size]
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            bsize = len(comp) + 25
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize))
            f.write(comp)
            f.write(struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
