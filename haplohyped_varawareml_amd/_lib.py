"""ctypes binding of libhhgt.so (include/hhgt.h).  Fails loudly when the library is missing:
there is no Python/CPU implementation of the path behind it."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# HHGT_LIB: load another build of the same library (development: tools/lz4_stats.py uses a -DHHGT_LZ4_STATS build)
LIB_PATH = os.environ.get("HHGT_LIB") or os.path.join(HERE, "libhhgt.so")

OK = 0
BLOSC1 = 1
BLOSC2 = 2
N_STAGES = 9
STAGE_NAMES = ["index", "fixed", "encode", "general", "lz4", "frame", "decode", "onehot", "inflate"]


class HhgtError(RuntimeError):
    """Raised for every non-zero libhhgt return code (the reference raises RuntimeError from
    cpp/parse_vcf.cpp:63-66)."""

    def __init__(self, code, msg):
        super().__init__(f"{msg} (hhgt rc={code})")
        self.code = code


class Layout(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("sc", C.c_int32), ("vc", C.c_int32), ("ring", C.c_int32),
                ("v_capacity", C.c_uint64)]


class Window(C.Structure):     # hhgt_window
    _fields_ = [("ref_ptr", C.c_uint64), ("ref_len", C.c_uint64), ("win_start", C.c_int64),
                ("var_start_ptr", C.c_uint64), ("var_ref_ptr", C.c_uint64), ("var_alt_ptr", C.c_uint64),
                ("geno_ptr", C.c_uint64), ("geno_first", C.c_uint32), ("var_lo", C.c_uint32),
                ("var_hi", C.c_uint32), ("reserved", C.c_uint32)]


class EncodeStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_lines", "n_records", "n_kept", "n_drop_region", "n_drop_filter", "n_haploid_padded",
        "n_malformed", "n_general_lines", "n_chrom_runs")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RESULT_RUNS = 16


class EncodeResultRec(C.Structure):     # hhgt_encode_result
    _fields_ = [("stats", EncodeStats), ("cursor_before", C.c_uint64), ("cursor_after", C.c_uint64),
                ("n_lines_over", C.c_uint64), ("err_density", C.c_uint64), ("run_first", C.c_uint64 * RESULT_RUNS),
                ("run_names", (C.c_char * 32) * RESULT_RUNS), ("v_capacity", C.c_uint64), ("done", C.c_uint32),
                ("reserved", C.c_uint32)]

    def chrom_runs(self):
        n = min(int(self.stats.n_chrom_runs), RESULT_RUNS)
        return [(int(self.run_first[i]), bytes(self.run_names[i]).split(b"\0")[0].decode()) for i in range(n)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: it must be in the process BEFORE libhhgt.so pulls in
    # libamdhip64, or the two runtimes disagree about the visible devices
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing. Build it with `python -m haplohyped_varawareml_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for this path.")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.hhgt_version.restype = C.c_char_p
    L.hhgt_last_error.restype = C.c_char_p
    L.hhgt_device_count.restype = i32
    L.hhgt_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.hhgt_ctx_destroy.argtypes = [vp]
    L.hhgt_ctx_destroy.restype = None
    L.hhgt_layout_bytes.restype = u64
    L.hhgt_layout_bytes.argtypes = [C.POINTER(Layout)]
    L.hhgt_layout_offset.restype = u64
    L.hhgt_layout_offset.argtypes = [C.POINTER(Layout), C.c_uint32, u64]
    L.hhgt_encode_text.argtypes = [vp, vp, u64, C.c_char_p, C.POINTER(Layout), u64, vp, vp, vp, vp, vp,
                                   C.POINTER(EncodeStats), vp]
    L.hhgt_encode_text_async.argtypes = [vp, vp, u64, C.c_char_p, C.POINTER(Layout), vp, C.c_uint32, vp, vp, vp, vp, vp,
                                         vp, vp]
    L.hhgt_planes_bytes.restype = u64
    L.hhgt_planes_bytes.argtypes = [C.POINTER(Layout)]
    L.hhgt_encode_text_planes_async.argtypes = [vp, vp, u64, C.c_char_p, C.POINTER(Layout), vp, C.c_uint32, vp, vp, vp, vp, vp, vp,
                                                vp, vp]
    L.hhgt_pad_tail_planes_cursor.argtypes = [vp, C.POINTER(Layout), vp, vp, vp]
    L.hhgt_pad_tail_planes.argtypes = [vp, C.POINTER(Layout), u64, u64, u64, vp, vp]
    L.hhgt_compress_planes.argtypes = [vp, C.POINTER(Layout), vp, vp, C.c_uint32, C.c_uint32, i32, vp, u64, vp, C.POINTER(u64), vp]
    L.hhgt_planes_expand.argtypes = [vp, C.POINTER(Layout), vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    L.hhgt_encode_result_status.argtypes = [vp]
    L.hhgt_pad_tail_cursor.argtypes = [vp, C.POINTER(Layout), vp, vp, vp]
    L.hhgt_encode_chrom_runs.argtypes = [vp, C.c_uint32, vp, vp, C.POINTER(C.c_uint32)]
    L.hhgt_pad_tail.argtypes = [vp, C.POINTER(Layout), u64, u64, u64, vp, vp]
    L.hhgt_set_clevel.argtypes = [vp, i32]
    L.hhgt_reserve.argtypes = [vp, u64, C.c_uint32, u64, u64, i32, i32]
    L.hhgt_set_keep_multiallelic.argtypes = [vp, i32]
    L.hhgt_set_index_mode.argtypes = [vp, i32]
    L.hhgt_stream_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.hhgt_stream_destroy.argtypes = [vp, vp]
    L.hhgt_set_frame_stream.argtypes = [vp, vp]
    L.hhgt_compress_bound.restype = u64
    L.hhgt_compress_bound.argtypes = [u64, u64, i32, i32]
    L.hhgt_compress_chunks.argtypes = [vp, vp, u64, u64, i32, i32, i32, vp, u64, vp, C.POINTER(u64), vp]
    L.hhgt_decompress_chunks.argtypes = [vp, vp, vp, u64, u64, i32, i32, vp, C.POINTER(u64), vp]
    L.hhgt_bgzf_scan.argtypes = [vp, u64, u64, vp, vp, vp, vp, C.POINTER(u64), C.POINTER(u64)]
    L.hhgt_inflate_members.argtypes = [vp, vp, u64, vp, vp, vp, vp, u64, vp, u64, vp, vp, C.POINTER(u64), vp]
    L.hhgt_onehot_windows.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, i32, vp, vp, vp]
    L.hhgt_onehot_bases_u8.argtypes = [vp, vp, u64, vp, i32, vp, vp]
    L.hhgt_profile_enable.argtypes = [vp, i32]
    L.hhgt_profile_reset.argtypes = [vp]
    L.hhgt_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise HhgtError(rc, load().hhgt_last_error().decode(errors="replace"))
