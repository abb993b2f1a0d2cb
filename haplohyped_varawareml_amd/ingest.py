"""ctypes binding of the streaming ingest engine (include/hhgt_ingest.h): files or host text in, framed genotype
chunks and variant tables out, every stage (host inflate / upload / encode / compress / download) running
concurrently inside libhhgt.  Replaces the per-(donor, chromosome) loop body of
/root/reference/src/haplohyped/vcf_to_h5.py:79-140 for all samples of a file at once."""
import ctypes as C
import gc
import threading
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import BLOSC2, check

EV_END, EV_HEADER, EV_VARIANTS, EV_COLUMNS, EV_INPUT_END = 0, 1, 2, 3, 4


class IngestOpts(C.Structure):
    _fields_ = [("sc", C.c_int32), ("vc", C.c_int32), ("typesize", C.c_int32), ("blocksize", C.c_int32),
                ("format", C.c_int32), ("sites_only", C.c_int32), ("device_inflate", C.c_int32), ("n_threads", C.c_int32),
                ("block_bytes", C.c_uint64), ("files_ahead", C.c_int32), ("expect_samples", C.c_int32)]


class IngestStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_samples", "n_lines", "n_records", "n_kept", "n_drop_region", "n_drop_filter",
                                          "n_haploid_padded", "n_general_lines", "text_bytes", "file_bytes", "raw_bytes",
                                          "compressed_bytes", "n_blocks")] + [
        ("seconds", C.c_double), ("is_bgzf", C.c_int32), ("device_inflate", C.c_int32)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class IngestEvent(C.Structure):
    _fields_ = [("kind", C.c_int32), ("input", C.c_int32),
                ("header", C.c_void_p), ("header_bytes", C.c_uint64), ("n_samples", C.c_uint64),
                ("start", C.c_void_p), ("ref", C.c_void_p), ("alt", C.c_void_p),
                ("first_variant", C.c_uint64), ("n_variants", C.c_uint64), ("n_runs", C.c_uint32), ("pad_", C.c_uint32),
                ("run_first", C.c_void_p), ("run_names", C.c_void_p),
                ("framed", C.c_void_p), ("chunk_off", C.c_void_p),
                ("framed_bytes", C.c_uint64), ("n_chunks", C.c_uint64), ("first_col", C.c_uint64), ("n_cols", C.c_uint64),
                ("raw_bytes", C.c_uint64), ("stats", IngestStats)]


Header = namedtuple("Header", "input header n_samples")
Variants = namedtuple("Variants", "input first start ref alt runs")
Columns = namedtuple("Columns", "input first_col n_cols framed chunk_off raw_bytes")
InputEnd = namedtuple("InputEnd", "input stats")

_bound = False


def _L():
    global _bound
    L = _lib.load()
    if not _bound:
        vp = C.c_void_p
        L.hhgt_ingest_open.argtypes = [vp, C.POINTER(IngestOpts), C.POINTER(vp)]
        L.hhgt_ingest_add_file.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.hhgt_ingest_add_memory.argtypes = [vp, vp, C.c_uint64, C.c_char_p]
        L.hhgt_ingest_finish.argtypes = [vp]
        L.hhgt_ingest_next.argtypes = [vp, C.POINTER(IngestEvent)]
        L.hhgt_ingest_hold.argtypes = [vp, C.POINTER(C.c_int)]
        L.hhgt_ingest_release.argtypes = [vp, C.c_int]
        L.hhgt_ingest_close.argtypes = [vp]
        L.hhgt_ingest_close.restype = None
        _bound = True
    return L


def _view(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(n) * np.dtype(dtype).itemsize,)).view(dtype)


def normalize_device_inflate(device_inflate):
    """-> False | True | "auto" from what callers and HHGT_DEVICE_INFLATE may say: booleans, 0 / 1, "auto", and the strings
    "0" / "false" / "no" / "off" / "host" / "" (host reader) and "1" / "true" / "yes" / "on" / "device".  Anything else is an
    error: `bool("host")` is True, and a misspelt policy must not silently pick the device inflater."""
    if isinstance(device_inflate, str):
        v = device_inflate.strip().lower()
        if v == "auto":
            return "auto"
        if v in ("", "0", "false", "no", "off", "host"):
            return False
        if v in ("1", "true", "yes", "on", "device"):
            return True
        raise ValueError(f"device_inflate={device_inflate!r}: expected a boolean, 'auto', 'host' or 'device'")
    return bool(device_inflate)


class Ingest:
    """One engine per (context, chunk geometry).  The context must not be used for anything else while the engine
    is open (its kernels are driven from the engine's threads)."""

    def __init__(self, ctx, sc=64, vc=8192, fmt=BLOSC2, sites_only=False, device_inflate=False, n_threads=0,
                 block_bytes=0, files_ahead=1, typesize=2, blocksize=0, expect_samples=0):
        """expect_samples > 0: the sample count of the inputs to come — the engine then makes and pins its buffers at open
        instead of inside the first input (include/hhgt_ingest.h)"""
        self.L = _L()
        self.ctx = ctx
        self._hold_lock = threading.Lock()
        self.h = None
        device_inflate = normalize_device_inflate(device_inflate)
        o = IngestOpts(int(sc), int(vc), int(typesize), int(blocksize), int(fmt), int(bool(sites_only)),
                       (2 if device_inflate == "auto" else int(device_inflate)), int(n_threads or 0), int(block_bytes or 0),
                       int(files_ahead), int(expect_samples or 0))
        h = C.c_void_p()
        check(self.L.hhgt_ingest_open(ctx.h, C.byref(o), C.byref(h)))
        self.h = h
        self._keep = []

    def add_file(self, path, region=""):
        rc = self.L.hhgt_ingest_add_file(self.h, str(path).encode(), (region or "").encode())
        if rc < 0:
            check(rc)
        return rc

    def add_memory(self, buf, region=""):
        """buf: uint8 numpy array or CPU torch tensor (pinned for full link speed); kept alive until close()"""
        self._keep.append(buf)
        ptr, n = (buf.data_ptr(), buf.numel()) if hasattr(buf, "data_ptr") else (buf.ctypes.data, buf.size)
        rc = self.L.hhgt_ingest_add_memory(self.h, C.c_void_p(ptr), int(n), (region or "").encode())
        if rc < 0:
            check(rc)
        return rc

    def finish(self):
        check(self.L.hhgt_ingest_finish(self.h))

    def events(self):
        """yields Header / Variants / Columns / InputEnd in order; the numpy views are valid until the next event"""
        ev = IngestEvent()
        # everything alive now (torch, numpy: some 10^5 container objects) moves to the permanent generation for the length
        # of the loop: a full collection while the engine's out slots wait on this thread costs 20-40 ms otherwise (measured:
        # one such pause per 3 M-variant pass, 10 % of it)
        ours = gc.get_freeze_count() == 0      # a caller that froze already keeps their freeze
        gc.freeze()
        try:
            while True:
                check(self.L.hhgt_ingest_next(self.h, C.byref(ev)))
                k = ev.kind
                if k == EV_END:
                    return
                if k == EV_HEADER:
                    yield Header(ev.input, C.string_at(ev.header, ev.header_bytes), int(ev.n_samples))
                elif k == EV_VARIANTS:
                    names = _view(ev.run_names, ev.n_runs * 32, np.uint8).reshape(-1, 32)
                    first = _view(ev.run_first, ev.n_runs, np.uint64)
                    runs = [(int(first[i]), bytes(names[i]).split(b"\0")[0].decode()) for i in range(ev.n_runs)]
                    yield Variants(ev.input, int(ev.first_variant), _view(ev.start, ev.n_variants, np.uint32),
                                   _view(ev.ref, ev.n_variants, np.uint8), _view(ev.alt, ev.n_variants, np.uint8), runs)
                elif k == EV_COLUMNS:
                    yield Columns(ev.input, int(ev.first_col), int(ev.n_cols), _view(ev.framed, ev.framed_bytes, np.uint8),
                                  _view(ev.chunk_off, ev.n_chunks + 1, np.uint64), int(ev.raw_bytes))
                elif k == EV_INPUT_END:
                    yield InputEnd(ev.input, ev.stats.asdict())
        finally:
            if ours:
                gc.unfreeze()

    def hold(self):
        """the buffers behind the event just yielded stay valid until release(token) — from any thread — instead of until
        the next event (hhgt_ingest_hold)"""
        t = C.c_int(-1)
        check(self.L.hhgt_ingest_hold(self.h, C.byref(t)))
        return int(t.value)

    def release(self, token):
        with self._hold_lock:        # (a writer thread may release while another thread closes the engine)
            if getattr(self, "h", None):
                check(self.L.hhgt_ingest_release(self.h, int(token)))

    def close(self):
        with self._hold_lock:
            if getattr(self, "h", None):
                self.L.hhgt_ingest_close(self.h)
                self.h = None
                self._keep = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
