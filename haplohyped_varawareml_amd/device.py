"""Thin Python layer over the C ABI (include/hhgt.h): torch tensors provide device memory and the
stream; every computation happens inside libhhgt.so's HIP kernels."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from ._lib import BLOSC1, BLOSC2, EncodeResultRec, EncodeStats, HhgtError, Layout, check

# default on-disk geometry: one HDF5 chunk = 64 samples x 8192 variants x 2 haplotypes (1 MiB);
# one Blosc2 block = half a sample row of the chunk = 4096 diploid calls (8 KiB -> two 4 KiB byte
# planes: ~10 KiB of LDS per LZ4 workgroup, so ~30 waves stay resident per CU — the kernel is latency /
# issue bound, not HBM bound); typesize 2 = one diploid call
DEFAULT_SC = 64
DEFAULT_VC = 8192
DEFAULT_TYPESIZE = 2
DEFAULT_BLOCKSIZE = 8192


def make_ring_layout(n_samples, ring_cols, sc=DEFAULT_SC, vc=DEFAULT_VC):
    """ring of `ring_cols` chunk columns (streaming: kept indices wrap, see include/hhgt.h)"""
    return Layout(int(n_samples), int(sc), int(vc), int(ring_cols), int(ring_cols) * int(vc))


class PendingEncode:
    """result record of one hhgt_encode_text_async call: pinned host memory the device writes when the call's work
    has run.  .wait() synchronises on the event recorded behind the call and raises what the synchronous call would."""

    def __init__(self):
        self.buf = torch.zeros(C.sizeof(EncodeResultRec), dtype=torch.uint8).pin_memory()
        self.rec = EncodeResultRec.from_address(self.buf.data_ptr())
        self.event = torch.cuda.Event()

    def wait(self):
        self.event.synchronize()
        check(_lib.load().hhgt_encode_result_status(C.c_void_p(self.buf.data_ptr())))
        return self.rec


def make_layout(n_samples, v_capacity, sc=DEFAULT_SC, vc=DEFAULT_VC):
    """chunk-tiled layout; v_capacity is rounded up to a whole number of chunk columns
    (dense: to a multiple of 128)."""
    if vc:
        v_capacity = -(-max(int(v_capacity), 1) // vc) * vc
    else:
        v_capacity = -(-max(int(v_capacity), 1) // 128) * 128
    return Layout(int(n_samples), int(sc), int(vc), 0, int(v_capacity))


def layout_bytes(lay):
    return int(_lib.load().hhgt_layout_bytes(C.byref(lay)))


def planes_bytes(lay):
    """bytes of the bit-plane form of the matrix under `lay` (a quarter of layout_bytes; 0: the layout cannot carry
    planes — variants per chunk must be a multiple of 4096)"""
    return int(_lib.load().hhgt_planes_bytes(C.byref(lay)))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class EncodeResult:
    G: torch.Tensor            # uint8 view of the chunk-tiled int8 matrix
    layout: Layout
    start: torch.Tensor
    stop: torch.Tensor
    ref: torch.Tensor
    alt: torch.Tensor
    n_kept: int
    stats: dict
    chrom_runs: list = field(default_factory=list)   # [(first_kept_index, name)]
    P: torch.Tensor = None     # bit-plane form of the matrix (include/hhgt.h), when the encode produced that instead of G

    def dense(self):
        """int8 [S, n_kept, 2] torch tensor (device) gathered out of the chunk-tiled buffer."""
        lay = self.layout
        S, n = lay.n_samples, self.n_kept
        if lay.sc == 0 and lay.vc == 0:
            return self.G.view(torch.int8).view(max(S, 1), lay.v_capacity, 2)[:S, :n]
        Sc, Vc = lay.sc or max(S, 1), lay.vc or lay.v_capacity
        n_sc = -(-S // Sc) if lay.sc else 1
        n_vc = lay.v_capacity // Vc
        g = self.G.view(torch.int8).view(n_vc, n_sc, Sc, Vc, 2).permute(1, 2, 0, 3, 4)
        return g.reshape(n_sc * Sc, n_vc * Vc, 2)[:S, :n]


def bgzf_scan(raw):
    """Member table of a BGZF byte string (host only; `hhgt_bgzf_scan`).  -> dict(data, comp_off, comp_len, isize,
    crc32, consumed): offsets / lengths of the raw DEFLATE payloads, inflated sizes and CRC-32s from the trailers, bytes
    covered by whole members."""
    from ._lib import load
    lib = load()
    data = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray, memoryview)) else np.ascontiguousarray(raw, dtype=np.uint8)
    offs, lens, isz, crcs = [], [], [], []
    pos, cap = 0, 1 << 16
    while pos < data.size:
        co = np.zeros(cap, dtype=np.uint64)
        cl = np.zeros(cap, dtype=np.uint32)
        iz = np.zeros(cap, dtype=np.uint32)
        cr = np.zeros(cap, dtype=np.uint32)
        n, used = C.c_uint64(0), C.c_uint64(0)
        check(lib.hhgt_bgzf_scan(C.c_void_p(data.ctypes.data + pos), data.size - pos, cap, C.c_void_p(co.ctypes.data),
                                 C.c_void_p(cl.ctypes.data), C.c_void_p(iz.ctypes.data), C.c_void_p(cr.ctypes.data),
                                 C.byref(n), C.byref(used)))
        if n.value == 0:
            break
        offs.append(co[:n.value] + np.uint64(pos))
        lens.append(cl[:n.value])
        isz.append(iz[:n.value])
        crcs.append(cr[:n.value])
        pos += used.value
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
    return dict(data=data, comp_off=cat(offs, np.uint64), comp_len=cat(lens, np.uint32), isize=cat(isz, np.uint32),
                crc32=cat(crcs, np.uint32), consumed=pos)


class Context:
    """One per (process, GPU).  Wraps hhgt_ctx."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise HhgtError(-6, "no HIP device visible to torch: libhhgt has no CPU path")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        check(self.lib.hhgt_ctx_create(device, C.byref(h)))
        self.h = h
        self.clevel = 5      # hhgt_ctx's default (the reference's compression_opts[4]); set_clevel keeps it in step

    def close(self):
        if getattr(self, "h", None):
            for st in getattr(self, "_streams", []):
                self.lib.hhgt_stream_destroy(self.h, st)
            self._streams = []
            self.lib.hhgt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- encode --------------------------------------------------------------------------------
    def encode_text(self, text, n_samples, region="", layout=None, v_base=0, out=None):
        """text: uint8 CUDA tensor holding whole VCF lines.  Returns EncodeResult; pass out=
        (a previous EncodeResult) to append at v_base into the same buffers."""
        assert text.is_cuda and text.dtype == torch.uint8 and text.is_contiguous()
        nbytes = text.numel()
        with torch.cuda.device(self.device):
            if out is None:
                if layout is None:
                    # capacity from a line-count bound: a kept line has at least 16 + 2*S bytes
                    bound = nbytes // (16 + 2 * max(n_samples, 0)) + 1
                    layout = make_layout(n_samples, bound)
                G = torch.zeros(max(layout_bytes(layout), 16), dtype=torch.uint8, device=self.device)
                cap = layout.v_capacity
                start = torch.zeros(cap, dtype=torch.int32, device=self.device)
                stop = torch.zeros(cap, dtype=torch.int32, device=self.device)
                ref = torch.zeros(cap, dtype=torch.uint8, device=self.device)
                alt = torch.zeros(cap, dtype=torch.uint8, device=self.device)
                out = EncodeResult(G, layout, start, stop, ref, alt, 0, {})
            st = EncodeStats()
            rc = self.lib.hhgt_encode_text(self.h, _ptr(text), nbytes, (region or "").encode(),
                                           C.byref(out.layout), int(v_base), _ptr(out.G), _ptr(out.start),
                                           _ptr(out.stop), _ptr(out.ref), _ptr(out.alt), C.byref(st), _stream())
            check(rc)
            out.stats = st.asdict()
            out.n_kept = int(v_base) + int(st.n_kept)
            out.chrom_runs = self.chrom_runs()
        return out

    def encode_text_async(self, text, n_samples, out, cursor, max_lines=None, region="", pending=None):
        """hhgt_encode_text_async: appends at the device-resident `cursor` (int64 tensor [1]) into `out`'s buffers and
        returns without waiting.  -> PendingEncode (call .wait() once the counts are needed)"""
        assert text.is_cuda and text.dtype == torch.uint8 and text.is_contiguous()
        nbytes = text.numel()
        if max_lines is None:
            max_lines = nbytes // (16 + 2 * max(n_samples, 0)) + 64
        pending = pending or PendingEncode()
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_encode_text_async(self.h, _ptr(text), nbytes, (region or "").encode(),
                                                  C.byref(out.layout), _ptr(cursor), int(max_lines), _ptr(out.G),
                                                  _ptr(out.start), _ptr(out.stop), _ptr(out.ref), _ptr(out.alt),
                                                  C.c_void_p(pending.buf.data_ptr()), _stream()))
            pending.event.record(torch.cuda.current_stream())
        return pending

    def pad_tail_cursor(self, res, cursor):
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_pad_tail_cursor(self.h, C.byref(res.layout), _ptr(cursor), _ptr(res.G), _stream()))

    # ---- bit-plane form of the matrix (include/hhgt.h "Bit-plane form": compressor-only consumers) -------------------
    def encode_text_planes_async(self, text, n_samples, out, cursor, max_lines=None, region="", pending=None):
        """hhgt_encode_text_planes_async: like encode_text_async, but the calls land in out.P as two bits per allele;
        out.G (may be None) only receives the bytes of calls beyond 0 / 1 / missing.  -> PendingEncode
        (.rec.reserved = number of such calls)"""
        assert text.is_cuda and text.dtype == torch.uint8 and text.is_contiguous() and out.P is not None
        nbytes = text.numel()
        if max_lines is None:
            max_lines = nbytes // (16 + 2 * max(n_samples, 0)) + 64
        pending = pending or PendingEncode()
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_encode_text_planes_async(self.h, _ptr(text), nbytes, (region or "").encode(),
                                                         C.byref(out.layout), _ptr(cursor), int(max_lines), _ptr(out.P), _ptr(out.G),
                                                         _ptr(out.start), _ptr(out.stop), _ptr(out.ref), _ptr(out.alt),
                                                         C.c_void_p(pending.buf.data_ptr()), _stream()))
            pending.event.record(torch.cuda.current_stream())
        return pending

    def pad_tail_planes_cursor(self, res, cursor):
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_pad_tail_planes_cursor(self.h, C.byref(res.layout), _ptr(cursor), _ptr(res.P), _stream()))

    def pad_tail_planes(self, res, v_end, vcol_begin=0, vcol_end=None):
        lay = res.layout
        Vc = lay.vc or lay.v_capacity
        if vcol_end is None:
            vcol_end = -(-max(v_end, 1) // Vc)
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_pad_tail_planes(self.h, C.byref(lay), int(v_end), int(vcol_begin), int(vcol_end), _ptr(res.P), _stream()))

    def compress_planes(self, res, col0=0, n_cols=None, fmt=BLOSC2, dst=None, chunk_off=None, sync=True):
        """hhgt_compress_planes: the chunks of column slots [col0, col0 + n_cols) of res (res.P: the planes; res.G or None:
        the int8 matrix whose bytes back the calls beyond 0 / 1 / missing), typesize 2, 8 KiB blocks.
        -> (dst, chunk_off, total_bytes or None) like compress()"""
        lay = res.layout
        Vc, Sc = lay.vc or lay.v_capacity, lay.sc or max(lay.n_samples, 1)
        if n_cols is None:
            n_cols = lay.v_capacity // Vc - col0
        n_chunks = n_cols * (-(-lay.n_samples // Sc) if lay.sc else 1)
        chunk_nbytes = Sc * Vc * 2
        with torch.cuda.device(self.device):
            cap = int(self.lib.hhgt_compress_bound(n_chunks, chunk_nbytes, 2, 8192))
            if dst is None:
                dst = torch.empty(cap, dtype=torch.uint8, device=self.device)
            if chunk_off is None:
                chunk_off = torch.zeros(n_chunks + 1, dtype=torch.int64, device=self.device)
            total = C.c_uint64(0)
            check(self.lib.hhgt_compress_planes(self.h, C.byref(lay), _ptr(res.P), _ptr(res.G), int(col0), int(n_cols), fmt, _ptr(dst),
                                                dst.numel(), _ptr(chunk_off), C.byref(total) if sync else None, _stream()))
        return dst, chunk_off, (int(total.value) if sync else None)

    def planes_expand(self, res, col0=0, n_cols=None, out=None):
        """hhgt_planes_expand: planes (+ res.G's bytes for the calls beyond 0 / 1 / missing) of column slots [col0, col0 + n_cols)
        -> the int8 matrix bytes at their place in `out` (a buffer of the int8 layout; default: a new zeroed one)"""
        lay = res.layout
        Vc = lay.vc or lay.v_capacity
        if n_cols is None:
            n_cols = lay.v_capacity // Vc - col0
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.zeros(layout_bytes(lay), dtype=torch.uint8, device=self.device)
            check(self.lib.hhgt_planes_expand(self.h, C.byref(lay), _ptr(res.P), _ptr(res.G), int(col0), int(n_cols), _ptr(out), _stream()))
        return out

    def chrom_runs(self):
        n = C.c_uint32(0)
        check(self.lib.hhgt_encode_chrom_runs(self.h, 0, None, None, C.byref(n)))
        if n.value == 0:
            return []
        first = np.zeros(n.value, np.uint64)
        names = np.zeros((n.value, 32), np.uint8)
        check(self.lib.hhgt_encode_chrom_runs(self.h, n.value, first.ctypes.data, names.ctypes.data, C.byref(n)))
        return [(int(first[i]), bytes(names[i]).rstrip(b"\0").decode()) for i in range(n.value)]

    def pad_tail(self, res, v_end=None, vcol_begin=0, vcol_end=None):
        lay = res.layout
        Vc = lay.vc or lay.v_capacity
        v_end = res.n_kept if v_end is None else v_end
        if vcol_end is None:
            vcol_end = -(-max(v_end, 1) // Vc)
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_pad_tail(self.h, C.byref(lay), int(v_end), int(vcol_begin), int(vcol_end),
                                         _ptr(res.G), _stream()))

    def create_stream(self, kind):
        """kind "encode" | "compress": the stream pair for running compress of one block beside encode of the next
        (include/hhgt.h hhgt_stream_create: the compress stream is restricted to 3/4 of the CUs) -> torch ExternalStream,
        destroyed with the context"""
        h = C.c_void_p()
        check(self.lib.hhgt_stream_create(self.h, {"encode": 0, "compress": 1, "frame": 2}[kind], C.byref(h)))
        self._streams = getattr(self, "_streams", [])
        self._streams.append(h)
        return torch.cuda.ExternalStream(h.value, device=self.device)

    def set_frame_stream(self, stream):
        """the framing half of every later compress call runs on `stream` (a torch stream; None: back to one stream) while the
        caller's stream goes on with the next call's LZ4 kernels — include/hhgt.h hhgt_set_frame_stream"""
        check(self.lib.hhgt_set_frame_stream(self.h, C.c_void_p(stream.cuda_stream) if stream is not None else None))

    def set_keep_multiallelic(self, on=True):
        """NON-REFERENCE mode: multi-allelic SNP sites pass the record filter (the reference's isSNP drops them);
        genotypes carry allele indices > 1.  Off by default."""
        check(self.lib.hhgt_set_keep_multiallelic(self.h, 1 if on else 0))

    def set_index_mode(self, mode):
        """how the line index finds the newlines of long records: 2 the walk (default), 1 the hop by the bound, 0 the plain
        scan, -1 back to the default (include/hhgt.h hhgt_set_index_mode); results are identical"""
        check(self.lib.hhgt_set_index_mode(self.h, int(mode)))

    def set_clevel(self, clevel):
        """Blosc clevel analogue = candidates tried per position: 1-2: none (offset-1 runs only), 3-4: 1, 5-6: 2 (default 5,
        the reference's setting), 7: 4, 8: 8, 9: 12 and a one-step lazy parse (what pipeline.stream_files / the converter
        write files with)"""
        check(self.lib.hhgt_set_clevel(self.h, int(clevel)))
        self.clevel = int(clevel)

    # ---- codec ---------------------------------------------------------------------------------
    def compress(self, src, chunk_nbytes, typesize=DEFAULT_TYPESIZE, blocksize=None, fmt=BLOSC2,
                 dst=None, chunk_off=None, sync=True):
        """src: uint8 CUDA tensor of n_chunks * chunk_nbytes bytes.
        -> (dst uint8 tensor, chunk_off int64 tensor [n_chunks+1], total_bytes or None)"""
        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
        chunk_nbytes = int(chunk_nbytes)
        assert src.numel() % chunk_nbytes == 0
        n_chunks = src.numel() // chunk_nbytes
        if blocksize is None:
            blocksize = min(chunk_nbytes, DEFAULT_BLOCKSIZE)
            blocksize -= blocksize % typesize
        with torch.cuda.device(self.device):
            cap = int(self.lib.hhgt_compress_bound(n_chunks, chunk_nbytes, typesize, blocksize))
            if cap == 0:
                check(-1)
            if dst is None:
                dst = torch.empty(cap, dtype=torch.uint8, device=self.device)
            if chunk_off is None:
                chunk_off = torch.zeros(n_chunks + 1, dtype=torch.int64, device=self.device)
            total = C.c_uint64(0)
            check(self.lib.hhgt_compress_chunks(self.h, _ptr(src), n_chunks, chunk_nbytes, typesize, blocksize,
                                                fmt, _ptr(dst), dst.numel(), _ptr(chunk_off),
                                                C.byref(total) if sync else None, _stream()))
        return dst, chunk_off, (int(total.value) if sync else None)

    def decompress(self, src, chunk_off, n_chunks, chunk_nbytes, typesize=DEFAULT_TYPESIZE, blocksize=None,
                   dst=None):
        """-> (dst uint8 tensor [n_chunks*chunk_nbytes], n_bad)"""
        if blocksize is None:
            blocksize = min(int(chunk_nbytes), DEFAULT_BLOCKSIZE)
            blocksize -= blocksize % typesize
        with torch.cuda.device(self.device):
            if dst is None:
                dst = torch.empty(int(n_chunks) * int(chunk_nbytes), dtype=torch.uint8, device=self.device)
            bad = C.c_uint64(0)
            check(self.lib.hhgt_decompress_chunks(self.h, _ptr(src), _ptr(chunk_off), int(n_chunks),
                                                  int(chunk_nbytes), typesize, blocksize, _ptr(dst),
                                                  C.byref(bad), _stream()))
        return dst, int(bad.value)

    # ---- BGZF on the device -------------------------------------------------------
    def inflate_bgzf(self, raw, return_status=False, check_crc=True):
        """raw: host bytes / uint8 array holding whole BGZF members.  The host walks the member headers
        (bgzf_scan), the compressed bytes are uploaded as they are and every member is inflated by one wave.
        -> (text uint8 tensor on the device, n_bad[, status tensor])"""
        tab = bgzf_scan(raw)
        host = tab["data"]
        n = len(tab["isize"])
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(tab["isize"], dtype=np.uint64, out=out_off[1:])
        total = int(out_off[-1])
        with torch.cuda.device(self.device):
            padded = np.zeros((host.size + 3) // 4 * 4 + 4, dtype=np.uint8)
            padded[:host.size] = host
            d_src = torch.from_numpy(padded).to(self.device)
            d_off = torch.from_numpy(tab["comp_off"]).to(self.device)
            d_len = torch.from_numpy(tab["comp_len"]).to(self.device)
            d_out = torch.from_numpy(out_off[:-1].copy()).to(self.device)
            d_isz = torch.from_numpy(tab["isize"]).to(self.device)
            d_crc = torch.from_numpy(tab["crc32"]).to(self.device) if check_crc else None
            dst = torch.empty(max(total, 1), dtype=torch.uint8, device=self.device)[:total]
            status = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)[:n]
            bad = C.c_uint64(0)
            check(self.lib.hhgt_inflate_members(self.h, _ptr(d_src), padded.size, _ptr(d_off), _ptr(d_len), _ptr(d_out),
                                                _ptr(d_isz), n, _ptr(dst), total, _ptr(d_crc), _ptr(status),
                                                C.byref(bad), _stream()))
        return (dst, int(bad.value), status) if return_status else (dst, int(bad.value))

    def inflate_members(self, d_src, src_bytes, d_comp_off, d_comp_len, d_out_off, d_isize, n, dst, dst_bytes, status,
                        count_bad=True, d_crc32=None):
        """`hhgt_inflate_members` on device tensors (see include/hhgt.h) -> number of members flagged in `status`
        (count_bad=False: launch only, no synchronisation, returns None — look at `status` later)"""
        bad = C.c_uint64(0)
        with torch.cuda.device(self.device):
            check(self.lib.hhgt_inflate_members(self.h, _ptr(d_src), int(src_bytes), _ptr(d_comp_off), _ptr(d_comp_len),
                                                _ptr(d_out_off), _ptr(d_isize), int(n), _ptr(dst), int(dst_bytes),
                                                _ptr(d_crc32), _ptr(status), C.byref(bad) if count_bad else None,
                                                _stream()))
        return int(bad.value) if count_bad else None

    # ---- synthetic workloads (bench / test tooling) -----------------------------------------------
    def synth_fixed(self, contig, table, n_samples, seed, v_first=0, with_header=True, names=None):
        """Render a fixed-width synthetic shard directly in HBM.  -> (text uint8 tensor, nbytes)"""
        from . import synth
        lib = self.lib
        if not hasattr(lib, "_synth_bound"):
            lib.hhgt_synth_render_fixed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                    C.c_char_p, C.c_int, C.c_uint64, C.c_void_p]
            lib._synth_bound = True
        S = int(n_samples)
        pos = table["pos"]
        V = len(pos)
        head = synth.header_text(contig, names or synth.sample_names(S)) if with_header else b""
        ll = synth.fixed_line_lengths(contig, pos, S)
        off = np.zeros(V + 1, dtype=np.uint64)
        off[0] = len(head)
        off[1:] = len(head) + np.cumsum(ll).astype(np.uint64)
        nbytes = int(off[-1])
        with torch.cuda.device(self.device):
            text = torch.empty(nbytes + 16, dtype=torch.uint8, device=self.device)
            if head:
                text[:len(head)] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(self.device)
            d_off = torch.from_numpy(off.view(np.int64)).to(self.device)
            d_pos = torch.from_numpy(pos.view(np.int32)).to(self.device)
            d_ref = torch.from_numpy(table["ref"]).to(self.device)
            d_alt = torch.from_numpy(table["alt"]).to(self.device)
            d_thr = torch.from_numpy(table["thr"].view(np.int32)).to(self.device)
            check(lib.hhgt_synth_render_fixed(self.h, _ptr(text), nbytes, _ptr(d_off), _ptr(d_pos), _ptr(d_ref),
                                              _ptr(d_alt), _ptr(d_thr), V, int(v_first), contig.encode(), S,
                                              int(seed), _stream()))
            torch.cuda.current_stream().synchronize()
        return text[:nbytes], nbytes

    def synth_mixed(self, contig, table, n_samples, seed, v_first=0, with_header=True, names=None):
        """config-4 style shard (synth.mixed_table) rendered in HBM -> (text uint8 tensor, nbytes, line_off)"""
        from . import synth
        lib = self.lib
        if not hasattr(lib, "_synth_mixed_bound"):
            lib.hhgt_synth_render_mixed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 6 + [
                C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_uint64, C.c_void_p]
            lib._synth_mixed_bound = True
        S = int(n_samples)
        V = len(table["pos"])
        head = synth.header_text(contig, names or synth.sample_names(S)) if with_header else b""
        ll = synth.mixed_line_lengths(contig, table, S)
        off = np.zeros(V + 1, dtype=np.uint64)
        off[0] = len(head)
        off[1:] = len(head) + np.cumsum(ll).astype(np.uint64)
        nbytes = int(off[-1])
        with torch.cuda.device(self.device):
            text = torch.empty(nbytes + 16, dtype=torch.uint8, device=self.device)
            if head:
                text[:len(head)] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(self.device)
            up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(self.device)
            d_off, d_pos = up(off, np.int64), up(table["pos"], np.int32)
            d_ref8, d_alt8 = up(table["ref8"], np.int64), up(table["alt8"], np.int64)
            d_meta, d_thr = up(table["meta"], np.int32), up(table["thr"], np.int32)
            check(lib.hhgt_synth_render_mixed(self.h, _ptr(text), nbytes, _ptr(d_off), _ptr(d_pos), _ptr(d_ref8),
                                              _ptr(d_alt8), _ptr(d_meta), _ptr(d_thr), V, int(v_first),
                                              contig.encode(), S, int(seed), _stream()))
            torch.cuda.current_stream().synchronize()
        return text[:nbytes], nbytes, off

    # ---- profiling -----------------------------------------------------------------------------
    def profile(self, on=True):
        check(self.lib.hhgt_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        check(self.lib.hhgt_profile_reset(self.h))

    def profile_read(self):
        ms = (C.c_double * _lib.N_STAGES)()
        n = (C.c_uint64 * _lib.N_STAGES)()
        check(self.lib.hhgt_profile_read(self.h, ms, n))
        return {name: dict(ms=ms[i], launches=int(n[i])) for i, name in enumerate(_lib.STAGE_NAMES) if n[i]}
