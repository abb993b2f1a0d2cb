"""ctypes wrapper over oracle/liboracle.so — CPU ORACLE, test infrastructure only.

May be imported ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (haplohyped_varawareml_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

BLOSC1 = 1
BLOSC2 = 2


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("vcf_oracle.c", "codec_oracle.c", "hhgt_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class VcfStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_lines", "n_records", "n_kept", "n_drop_region", "n_drop_filter",
        "n_haploid_padded", "n_malformed")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u8p = C.c_void_p
        L.oracle_set_keep_multiallelic.argtypes = [C.c_int]
        L.oracle_set_keep_multiallelic.restype = None
        L.oracle_vcf_encode.restype = C.c_int64
        L.oracle_vcf_encode.argtypes = [u8p, C.c_size_t, C.c_char_p, C.c_int, C.c_size_t, u8p, u8p,
                                        u8p, u8p, u8p, u8p, C.POINTER(VcfStats)]
        L.oracle_vcf_load_sample.restype = C.c_int64
        L.oracle_vcf_load_sample.argtypes = [u8p, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_size_t,
                                             u8p, u8p, u8p, u8p, u8p, C.POINTER(VcfStats)]
        L.oracle_vcf_header_samples.restype = C.c_int
        L.oracle_vcf_header_samples.argtypes = [u8p, C.c_size_t, u8p, u8p, C.c_int]
        L.oracle_shuffle.argtypes = [u8p, u8p, C.c_size_t, C.c_int]
        L.oracle_unshuffle.argtypes = [u8p, u8p, C.c_size_t, C.c_int]
        L.oracle_lz4_bound.restype = C.c_int
        L.oracle_lz4_bound.argtypes = [C.c_int]
        L.oracle_lz4_compress.restype = C.c_int
        L.oracle_lz4_compress.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.oracle_lz4_decompress.restype = C.c_int
        L.oracle_lz4_decompress.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.oracle_blosc_bound.restype = C.c_size_t
        L.oracle_blosc_bound.argtypes = [C.c_size_t, C.c_int, C.c_int]
        L.oracle_blosc_compress.restype = C.c_int64
        L.oracle_blosc_compress.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t]
        L.oracle_blosc_decompress.restype = C.c_int64
        L.oracle_blosc_decompress.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _as_u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def header_samples(text):
    t = _as_u8(text)
    n = lib().oracle_vcf_header_samples(_ptr(t), t.size, None, None, 0)
    if n < 0:
        return None
    off = np.zeros(max(n, 1), np.uint32)
    ln = np.zeros(max(n, 1), np.uint32)
    lib().oracle_vcf_header_samples(_ptr(t), t.size, _ptr(off), _ptr(ln), n)
    tb = t.tobytes()
    return [tb[int(o):int(o) + int(l)].decode() for o, l in zip(off[:n], ln[:n])]


def vcf_encode(text, n_samples, region="", cap=None, want_chrom=False, keep_multiallelic=False):
    """-> dict(G int8 [S, n_kept, 2], start, stop, ref, alt, chrom?, stats).  keep_multiallelic: the labelled
    non-reference filter mode of include/hhgt.h (multi-allelic SNP sites kept)"""
    t = _as_u8(text)
    if cap is None:
        cap = int(np.count_nonzero(t == 10)) + 1
    cap = max(int(cap), 1)
    G = np.zeros((max(n_samples, 1), cap, 2), np.int8)
    start = np.zeros(cap, np.uint32)
    stop = np.zeros(cap, np.uint32)
    ref = np.zeros(cap, np.uint8)
    alt = np.zeros(cap, np.uint8)
    chrom = np.zeros((cap, 32), np.uint8) if want_chrom else None
    st = VcfStats()
    lib().oracle_set_keep_multiallelic(1 if keep_multiallelic else 0)
    try:
        n = lib().oracle_vcf_encode(_ptr(t), t.size, region.encode(), n_samples, cap, _ptr(G), _ptr(start),
                                    _ptr(stop), _ptr(ref), _ptr(alt), _ptr(chrom), C.byref(st))
    finally:
        lib().oracle_set_keep_multiallelic(0)
    if n < 0:
        raise RuntimeError(f"oracle_vcf_encode failed rc={n} stats={st.asdict()}")
    out = dict(G=np.ascontiguousarray(G[:n_samples, :n]), start=start[:n], stop=stop[:n], ref=ref[:n],
               alt=alt[:n], stats=st.asdict(), n_kept=int(n))
    if want_chrom:
        out["chrom"] = [bytes(r).rstrip(b"\0").decode() for r in chrom[:n]]
    return out


def vcf_load_sample(text, n_samples, sample_index, region="", cap=None):
    t = _as_u8(text)
    if cap is None:
        cap = int(np.count_nonzero(t == 10)) + 1
    phase = np.zeros((cap, 2), np.int8)
    start = np.zeros(cap, np.uint32)
    stop = np.zeros(cap, np.uint32)
    ref = np.zeros(cap, np.uint8)
    alt = np.zeros(cap, np.uint8)
    st = VcfStats()
    n = lib().oracle_vcf_load_sample(_ptr(t), t.size, region.encode(), n_samples, sample_index, cap,
                                     _ptr(phase), _ptr(start), _ptr(stop), _ptr(ref), _ptr(alt), C.byref(st))
    if n < 0:
        raise RuntimeError(f"oracle_vcf_load_sample failed rc={n}")
    return dict(phase=phase[:n], start=start[:n], stop=stop[:n], ref=ref[:n], alt=alt[:n],
                stats=st.asdict(), n_kept=int(n))


def shuffle(buf, typesize):
    s = _as_u8(buf)
    d = np.empty_like(s)
    lib().oracle_shuffle(_ptr(s), _ptr(d), s.size, typesize)
    return d


def unshuffle(buf, typesize):
    s = _as_u8(buf)
    d = np.empty_like(s)
    lib().oracle_unshuffle(_ptr(s), _ptr(d), s.size, typesize)
    return d


def lz4_compress(buf):
    s = _as_u8(buf)
    cap = lib().oracle_lz4_bound(s.size)
    d = np.empty(cap, np.uint8)
    c = lib().oracle_lz4_compress(_ptr(s), s.size, _ptr(d), cap)
    if c <= 0:
        raise RuntimeError("oracle_lz4_compress failed")
    return d[:c].copy()


def lz4_decompress(buf, nbytes):
    s = _as_u8(buf)
    d = np.empty(max(nbytes, 1), np.uint8)
    n = lib().oracle_lz4_decompress(_ptr(s), s.size, _ptr(d), nbytes)
    if n < 0:
        raise RuntimeError(f"oracle_lz4_decompress rc={n}")
    return d[:n]


def blosc_compress(buf, typesize, blocksize, fmt=BLOSC2):
    s = _as_u8(buf)
    cap = lib().oracle_blosc_bound(s.size, typesize, blocksize)
    d = np.empty(cap, np.uint8)
    c = lib().oracle_blosc_compress(_ptr(s), s.size, typesize, blocksize, fmt, _ptr(d), cap)
    if c < 0:
        raise RuntimeError(f"oracle_blosc_compress rc={c}")
    return d[:c].copy()


def blosc_decompress(chunk):
    s = _as_u8(chunk)
    nbytes = int(s[4:8].view("<u4")[0])
    d = np.empty(max(nbytes, 1), np.uint8)
    n = lib().oracle_blosc_decompress(_ptr(s), s.size, _ptr(d), nbytes)
    if n < 0:
        raise RuntimeError(f"oracle_blosc_decompress rc={n}")
    return d[:n]
