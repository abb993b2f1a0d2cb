"""File -> device pipeline: host inflate (C++ reader threads) -> pinned ring -> hipMemcpyAsync on a copy
stream (double-buffered against the kernels) -> hhgt_encode_text appends into a staging window of chunk
columns -> completed columns go through hhgt_compress_chunks and leave the device framed.

Replaces the body of VCFtoHDF5Converter.genotype_vcf_to_hdf5
(/root/reference/src/haplohyped/vcf_to_h5.py:79-140): one pass per FILE for all samples instead of one
pass per (donor, chromosome)."""
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import device as dev
from .reader import VcfReader, parse_header


def _is_bgzf_file(path):
    with open(path, "rb") as f:
        h = f.read(18)
    return len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[12:16] == b"BC\x02\x00"


@dataclass
class FileStats:
    n_samples: int = 0
    n_lines: int = 0
    n_records: int = 0
    n_kept: int = 0
    n_drop_region: int = 0
    n_drop_filter: int = 0
    n_haploid_padded: int = 0
    n_general_lines: int = 0
    text_bytes: int = 0
    file_bytes: int = 0
    raw_bytes: int = 0
    compressed_bytes: int = 0
    seconds: float = 0.0
    t_setup: float = 0.0      # host seconds: open, first block, buffers
    t_source: float = 0.0     # host seconds inside the block source (launch / wait for the next text block)
    t_encode: float = 0.0     # host seconds inside hhgt_encode_text (synchronises)
    t_emit: float = 0.0       # host seconds compressing + handing out completed columns
    is_bgzf: bool = False
    samples: list = field(default_factory=list)
    chrom_runs: list = field(default_factory=list)


def _accumulate(fs, st):
    for k in ("n_lines", "n_records", "n_drop_region", "n_drop_filter", "n_haploid_padded", "n_general_lines"):
        setattr(fs, k, getattr(fs, k) + st[k])


class _HostBlocks:
    """Text blocks from the C++ reader (host inflate / pread threads -> pinned ring -> hipMemcpyAsync)."""

    def __init__(self, ctx, path, block_bytes, n_threads):
        self.rd = VcfReader(path, block_bytes=block_bytes, n_threads=n_threads)
        self.is_bgzf = self.rd.is_bgzf
        self._blk = None

    def first(self, dbuf, copy_stream):
        blk = self.rd.next_block()
        if blk is None:
            return None, 0
        self.rd.copy_async(blk, dbuf.data_ptr(), copy_stream.cuda_stream)
        copy_stream.synchronize()
        return blk, blk.size

    def start_next(self, dbuf, prev, prev_n, copy_stream):
        self._blk = self.rd.next_block()
        if self._blk is not None:
            self.rd.copy_async(self._blk, dbuf.data_ptr(), copy_stream.cuda_stream)   # overlaps the kernels of the caller

    def finish_next(self, dbuf, copy_stream):
        copy_stream.synchronize()   # the block sits in HBM before its pinned source is recycled
        return self._blk.size if self._blk is not None else 0

    def file_bytes(self):
        return self.rd.stats()["file_bytes"]

    def close(self):
        self.rd.close()


class _DeviceBgzfBlocks:
    """Text blocks of a BGZF file inflated on the device: the host walks the member headers of the
    memory-mapped file, the compressed members of one block cross PCIe as they are, `hhgt_inflate_members` writes
    their text behind the partial last line carried over from the previous block.  A block handed to the caller
    ends with a newline (except the last one of the file)."""
    is_bgzf = True

    def __init__(self, ctx, path, block_bytes):
        self.ctx = ctx
        self.raw = np.memmap(path, dtype=np.uint8, mode="r")
        tab = dev.bgzf_scan(self.raw)
        if tab["consumed"] != self.raw.size:
            raise dev.HhgtError(-4, f"{path}: {self.raw.size - tab['consumed']} bytes behind the last whole BGZF member")
        self.off, self.len, self.isz, self.crc = tab["comp_off"], tab["comp_len"], tab["isize"], tab["crc32"]
        if block_bytes is None:
            block_bytes = min(1 << 30, (int(self.isz.sum(dtype=np.uint64)) + (9 << 20)) // (1 << 20) * (1 << 20))
        self.block_bytes = block_bytes
        self.m = 0                      # next member
        self._pending = None
        # members per block: inflated sizes are known, keep one maximal line of slack for the carry
        self.room = block_bytes - (4 << 20)
        if self.room < (1 << 20):
            raise dev.HhgtError(-1, "device BGZF inflate needs block_bytes of at least 5 MiB")

    def _inflate_into(self, dbuf, carry):
        """members [m, m1) -> dbuf[carry:], returns bytes in dbuf; the stream is the current torch stream"""
        isz = self.isz
        m0, total = self.m, 0
        m1 = m0
        while m1 < len(isz) and total + int(isz[m1]) <= self.room - carry:
            total += int(isz[m1])
            m1 += 1
        if m1 == m0 and m0 < len(isz):
            raise dev.HhgtError(-4, "a line longer than the text block: raise block_bytes")
        self.m = m1
        if m1 == m0:
            return carry
        a = int(self.off[m0])
        b = int(self.off[m1 - 1]) + int(self.len[m1 - 1])
        d = self.ctx.device
        padded = np.zeros((b - a + 3) // 4 * 4 + 4, dtype=np.uint8)
        padded[:b - a] = self.raw[a:b]
        out_off = np.zeros(m1 - m0, dtype=np.uint64)
        np.cumsum(isz[m0:m1 - 1], dtype=np.uint64, out=out_off[1:])
        out_off += np.uint64(carry)
        d_src = torch.from_numpy(padded).to(d, non_blocking=True)
        d_off = torch.from_numpy(self.off[m0:m1] - np.uint64(a)).to(d, non_blocking=True)
        d_len = torch.from_numpy(np.ascontiguousarray(self.len[m0:m1])).to(d, non_blocking=True)
        d_out = torch.from_numpy(out_off).to(d, non_blocking=True)
        d_isz = torch.from_numpy(np.ascontiguousarray(isz[m0:m1])).to(d, non_blocking=True)
        d_crc = torch.from_numpy(np.ascontiguousarray(self.crc[m0:m1])).to(d, non_blocking=True)
        status = torch.zeros(m1 - m0, dtype=torch.int32, device=d)
        self.ctx.inflate_members(d_src, padded.size, d_off, d_len, d_out, d_isz, m1 - m0, dbuf, carry + total, status,
                                 count_bad=False, d_crc32=d_crc)   # no host wait here: the caller's kernels are launched next
        self._pending = (status, m0)
        return carry + total

    def _check(self):
        """raises if a member of the last launch failed (synchronises the inflate stream)"""
        if self._pending is not None:
            status, m0 = self._pending
            self._pending = None
            nz = torch.nonzero(status)
            if nz.numel():
                k = int(nz[0])
                code = int(status[k])
                what = "CRC-32 of the inflated text differs from the trailer" if code == 9 else f"DEFLATE stream is corrupt (status {code})"
                raise dev.HhgtError(-7, f"BGZF member {m0 + k}: {what}")

    def _cut(self, dbuf, n):
        """bytes of dbuf[:n] that are whole lines (everything at end of file)"""
        if self.m >= len(self.isz) or n == 0:
            return n
        lo = max(0, n - (4 << 20))
        nl = torch.nonzero(dbuf[lo:n] == 10)
        if nl.numel() == 0:
            raise dev.HhgtError(-4, "a line longer than 4 MiB")
        return lo + int(nl[-1]) + 1

    def first(self, dbuf, copy_stream):
        with torch.cuda.stream(copy_stream):
            n = self._inflate_into(dbuf, 0)
            self._check()
            cut = self._cut(dbuf, n)
            # the '#' lines only: the header ends behind the first newline that is not followed by '#' (searched on
            # the device, within the same 64 MiB bound the host reader accepts)
            lim = min(cut, 64 << 20)
            hend = lim
            if lim > 1:
                seg = dbuf[:lim]
                if int(seg[0]) != 35:
                    hend = 0
                else:
                    m = torch.nonzero((seg[:-1] == 10) & (seg[1:] != 35))
                    if m.numel():
                        hend = int(m[0]) + 1
            head = dbuf[:max(hend, min(lim, 1 << 12))].cpu().numpy()
        copy_stream.synchronize()
        self._tail = (n, cut)
        return (head if cut else None), cut

    def start_next(self, dbuf, prev, prev_n, copy_stream):
        n, cut = self._tail
        with torch.cuda.stream(copy_stream):
            carry = n - cut
            if carry:
                dbuf[:carry].copy_(prev[cut:n])
            n2 = self._inflate_into(dbuf, carry)
        self._tail = (n2, None)

    def finish_next(self, dbuf, copy_stream):
        with torch.cuda.stream(copy_stream):
            self._check()
            n2 = self._tail[0]
            cut2 = self._cut(dbuf, n2)
        copy_stream.synchronize()
        self._tail = (n2, cut2)
        return cut2

    def file_bytes(self):
        return int(self.raw.size)

    def close(self):
        self.raw = None


class RawColumns:
    """what on_columns gets as its first argument from the native engine: the size of the uncompressed columns
    (the engine hands out framed chunks only; the legacy loop passes the device tensor itself)"""

    def __init__(self, nbytes):
        self._n = int(nbytes)

    def numel(self):
        return self._n


def _want_device_inflate(device_inflate):
    """False: the host reader inflates (the north star's design, always selectable); True: BGZF files are inflated on the
    device; "auto" (the default since round 3; HHGT_DEVICE_INFLATE=0|1|auto overrides): in the ENGINE (stream_files, the
    converter: include/hhgt_ingest.h) BGZF files whose first member inflates to at least twice its size take the device —
    their compressed members are the smaller load for the host-device link, which bounds the host path at cohort widths —
    everything else the host reader; in the single-file loop stream_file "auto" means every BGZF file unless the caller's
    text blocks are below 5 MiB (too few members per launch for the device inflater).  Strings are normalised by
    ingest.normalize_device_inflate (unknown ones raise)."""
    from .ingest import normalize_device_inflate
    if device_inflate is None:
        device_inflate = os.environ.get("HHGT_DEVICE_INFLATE", "auto")
    return normalize_device_inflate(device_inflate)


FILE_CLEVEL = 9    # effort of the file-writing paths (stream_files, the converter): see stream_files


def peek_sample_count(path, limit=64 << 20):
    """number of sample columns of a .vcf / .vcf.gz file from its #CHROM line (the first `limit` bytes of text at most), or
    0 when it cannot be told cheaply — what hhgt_ingest_opts.expect_samples wants to hear before the engine opens"""
    import gzip
    try:
        with open(path, "rb") as f:
            magic = f.read(2)
        op = gzip.open if magic == b"\x1f\x8b" else open
        got = 0
        with op(path, "rb") as f:
            for line in f:
                got += len(line)
                if line.startswith(b"#CHROM"):
                    return max(len(line.rstrip(b"\r\n").split(b"\t")) - 9, 0)
                if not line.startswith(b"#") or got > limit:
                    return 0
    except OSError:
        return 0
    return 0


def stream_files(ctx, jobs, sc=dev.DEFAULT_SC, vc=dev.DEFAULT_VC, block_bytes=None, n_threads=0, sites_only=False,
                 fmt=dev.BLOSC2, device_inflate=None, on_header=None, on_variants=None, on_columns=None, on_end=None,
                 files_ahead=1, expect_samples=None, clevel=None, hold_columns=False):
    """Several inputs through ONE native ingest engine (csrc/ingest.hip): the inflate of the next file overlaps the
    encode of the current one, and nothing waits on the host between blocks.
    jobs: [(path_or_host_buffer, region)]; callbacks get the job index first:
        on_header(i, sample_names)   on_variants(i, start, ref, alt)   on_columns(i, RawColumns, n_cols, (bytes, offsets))
        on_end(i, FileStats)
    hold_columns: on_columns gets a fifth argument `release` and the (bytes, offsets) views stay valid until it is called
        (from any thread) instead of until the callback returns — a consumer that writes the chunks on another thread
        (hhgt_ingest_hold; the engine has five chunk buffers, so at most four batches can be in flight behind the current one)
    -> [FileStats] in job order"""
    from .ingest import Columns, Header, Ingest, InputEnd, Variants
    stats = [FileStats() for _ in jobs]
    # effort: files are written once and read often, and this path is bound by its input (5-11 M variants/s against > 100 M
    # for the kernels alone), so it runs at the reference's codec strength — clevel 9: twelve candidates per one and the lazy
    # rule, 6.9x on 1000G-shaped planes where the kernel-only default (clevel 5) packs 6.4x and LZ4HC level 5, what the
    # reference's compression_opts select (vcf_to_h5.py:135), 7.0x.  HHGT_FILE_CLEVEL / clevel= override; the context's own
    # level is restored afterwards.
    if clevel is None:
        clevel = int(os.environ.get("HHGT_FILE_CLEVEL", FILE_CLEVEL))
    prev_clevel = getattr(ctx, "clevel", 5)
    if hasattr(ctx, "set_clevel"):
        ctx.set_clevel(clevel)
    try:
        return _stream_files(ctx, jobs, stats, sc, vc, block_bytes, n_threads, sites_only, fmt, device_inflate, on_header, on_variants,
                             on_columns, on_end, files_ahead, expect_samples, hold_columns)
    finally:
        if hasattr(ctx, "set_clevel"):
            ctx.set_clevel(prev_clevel)


def _stream_files(ctx, jobs, stats, sc, vc, block_bytes, n_threads, sites_only, fmt, device_inflate, on_header, on_variants, on_columns,
                  on_end, files_ahead, expect_samples, hold_columns=False):
    from .ingest import Columns, Header, Ingest, InputEnd, Variants
    if expect_samples is None:     # the first file's header says how wide the cohort is: the engine sizes and pins at open
        first = next((src for src, _ in jobs if isinstance(src, (str, os.PathLike))), None)
        expect_samples = peek_sample_count(first) if first is not None else 0
    with Ingest(ctx, sc=sc, vc=vc, fmt=fmt, sites_only=sites_only, device_inflate=_want_device_inflate(device_inflate),
                n_threads=n_threads, block_bytes=block_bytes or 0, files_ahead=files_ahead, expect_samples=expect_samples) as ing:
        for src, region in jobs:
            if isinstance(src, (str, os.PathLike)):
                ing.add_file(src, region)
            else:
                ing.add_memory(src, region)
        ing.finish()
        for ev in ing.events():
            fs = stats[ev.input]
            if isinstance(ev, Header):
                names, _ = parse_header(np.frombuffer(ev.header, dtype=np.uint8))
                fs.samples = names
                if on_header:
                    on_header(ev.input, names)
            elif isinstance(ev, Variants):
                fs.chrom_runs.extend(ev.runs)
                if on_variants and len(ev.start):
                    on_variants(ev.input, ev.start, ev.ref, ev.alt)
            elif isinstance(ev, Columns):
                if on_columns and hold_columns:
                    token = ing.hold()
                    on_columns(ev.input, RawColumns(ev.raw_bytes), ev.n_cols, (ev.framed, ev.chunk_off),
                               (lambda t=token: ing.release(t)))
                elif on_columns:
                    on_columns(ev.input, RawColumns(ev.raw_bytes), ev.n_cols, (ev.framed, ev.chunk_off))
            elif isinstance(ev, InputEnd):
                for k, v in ev.stats.items():
                    if hasattr(fs, k):
                        setattr(fs, k, v)
                fs.is_bgzf = bool(ev.stats["is_bgzf"])
                if on_end:
                    on_end(ev.input, fs)
    return stats


def stream_file(ctx, path, region="", sc=dev.DEFAULT_SC, vc=dev.DEFAULT_VC, block_bytes=None, n_threads=0,
                sites_only=False, on_columns=None, on_variants=None, on_header=None, compress=True, fmt=dev.BLOSC2,
                device_inflate=None):
    """Streams one VCF through the device path.  With compress=True and a chunk-tiled layout (the converter's case)
    the native engine runs it (stream_files); otherwise the Python loop below (dense / uncompressed columns handed
    out as device tensors: the parse_vcf facade and tests).
    on_header(samples)                         once
    on_variants(start, ref, alt)               numpy arrays for each text block's kept records
    on_columns(G_cols, n_cols, framed)         for every batch of completed chunk columns:
        G_cols: uint8 CUDA tensor [n_cols * column_bytes] (valid during the call),
        framed: (host bytes, offsets uint64[n_chunks+1]) when compress=True else None
    device_inflate: BGZF members are inflated on the device instead of by the host reader threads (None: the
        HHGT_DEVICE_INFLATE environment variable, default off — the north star keeps BGZF on the host)
    -> FileStats"""
    if compress and sc and not sites_only:
        return stream_files(ctx, [(path, region)], sc=sc, vc=vc, block_bytes=block_bytes, n_threads=n_threads, fmt=fmt,
                            device_inflate=device_inflate,
                            on_header=(lambda i, names: on_header(names)) if on_header else None,
                            on_variants=(lambda i, a, b, c: on_variants(a, b, c)) if on_variants else None,
                            on_columns=(lambda i, g, n, f: on_columns(g, n, f)) if on_columns else None)[0]
    t_start = time.perf_counter()
    fs = FileStats()
    d = ctx.device
    device_inflate = _want_device_inflate(device_inflate)
    if device_inflate == "auto":   # this single-file form: every BGZF file, unless the caller's text blocks are too small for the device path
        device_inflate = block_bytes is None or block_bytes >= (5 << 20)
    if device_inflate and _is_bgzf_file(path):
        # one wave per member: a launch wants >= 10 k members (64 KiB of text each) to fill the chip, so the text
        # block is 1 GiB unless the file is smaller (measured: 64 MiB blocks 1.4 M variants/s, 1 GiB 4.0 M)
        rd = _DeviceBgzfBlocks(ctx, path, block_bytes)
        block_bytes = rd.block_bytes
    else:
        if block_bytes is None:
            block_bytes = 64 << 20
        rd = _HostBlocks(ctx, path, block_bytes, n_threads)
    try:
        fs.is_bgzf = rd.is_bgzf
        dbuf = [torch.empty(block_bytes + 64, dtype=torch.uint8, device=d) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=d)
        main = torch.cuda.current_stream(d)
        blk, blk_n = rd.first(dbuf[0], copy_stream)
        if blk is None:
            raise dev.HhgtError(-4, f"{path}: empty file (no VCF header)")
        names, _ = parse_header(blk)
        fs.samples = names
        S = 0 if sites_only else len(names)
        fs.n_samples = S
        if on_header:
            on_header(names)
        max_lines = block_bytes // (2 * S + 16) + 2
        W = -(-max_lines // vc) + 1
        lay = dev.make_layout(S, W * vc, sc=sc, vc=vc)
        col_bytes = dev.layout_bytes(lay) // W if S else 0
        chunk_nbytes = sc * vc * 2
        cap = lay.v_capacity

        def new_stage():
            return dev.EncodeResult(torch.zeros(max(dev.layout_bytes(lay), 16), dtype=torch.uint8, device=d), lay,
                                    torch.zeros(cap, dtype=torch.int32, device=d),
                                    torch.zeros(cap, dtype=torch.int32, device=d),
                                    torch.zeros(cap, dtype=torch.uint8, device=d),
                                    torch.zeros(cap, dtype=torch.uint8, device=d), 0, {})

        stage, spare = new_stage(), new_stage()
        fs.t_setup = time.perf_counter() - t_start
        cur_n, i, fill, v_global = blk_n, 0, 0, 0
        last_run = None
        while cur_n:
            t0 = time.perf_counter()
            rd.start_next(dbuf[1 - i], dbuf[i], cur_n, copy_stream)   # overlaps the kernels below
            t1 = time.perf_counter()
            ctx.encode_text(dbuf[i][:cur_n], S, region=region, v_base=fill, out=stage)
            t2 = time.perf_counter()
            fs.t_source += t1 - t0
            fs.t_encode += t2 - t1
            st = stage.stats
            _accumulate(fs, st)
            fs.text_bytes += cur_n
            k = st["n_kept"]
            for first, name in stage.chrom_runs:
                if name != last_run:
                    fs.chrom_runs.append((v_global + first, name))
                    last_run = name
            if on_variants and k:
                on_variants(stage.start[fill:fill + k].cpu().numpy().view(np.uint32),
                            stage.ref[fill:fill + k].cpu().numpy(), stage.alt[fill:fill + k].cpu().numpy())
            total = fill + k
            done = total // vc
            if done and S:
                t0 = time.perf_counter()
                _emit(ctx, stage, done, col_bytes, chunk_nbytes, on_columns, compress, fmt, fs)
                fs.t_emit += time.perf_counter() - t0
            rem = total - done * vc
            if done:
                # carry the partial column into column 0 of the other staging window
                if rem and S:
                    spare.G[:col_bytes].copy_(stage.G[done * col_bytes:(done + 1) * col_bytes])
                stage, spare = spare, stage
            fill = rem
            v_global += k
            t0 = time.perf_counter()
            nxt_n = rd.finish_next(dbuf[1 - i], copy_stream)
            fs.t_source += time.perf_counter() - t0
            main.synchronize()
            cur_n = nxt_n
            i ^= 1
        if fill and S:
            stage.n_kept = fill
            ctx.pad_tail(stage, v_end=fill, vcol_begin=0, vcol_end=1)
            _emit(ctx, stage, 1, col_bytes, chunk_nbytes, on_columns, compress, fmt, fs)
        elif S:
            pass
        fs.n_kept = v_global
        fs.file_bytes = rd.file_bytes()
    finally:
        rd.close()
    fs.seconds = time.perf_counter() - t_start
    return fs


def _pinned(ctx, nbytes):
    buf = getattr(ctx, "_pinned_out", None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
        ctx._pinned_out = buf
    return buf


def _emit(ctx, stage, n_cols, col_bytes, chunk_nbytes, on_columns, compress, fmt, fs):
    lay = stage.layout
    # sample-padding rows of the completed columns must be zero before they are framed
    if lay.sc and lay.n_samples % lay.sc:
        ctx.pad_tail(stage, v_end=n_cols * lay.vc, vcol_begin=0, vcol_end=n_cols)
    cols = stage.G[: n_cols * col_bytes]
    framed = None
    fs.raw_bytes += cols.numel()
    if compress:
        dst, off, total = ctx.compress(cols, chunk_nbytes, typesize=dev.DEFAULT_TYPESIZE, blocksize=min(lay.vc * 2, dev.DEFAULT_BLOCKSIZE), fmt=fmt)
        # framed bytes leave the device through a pinned staging buffer (pageable D2H runs at a fraction of
        # the link rate); the numpy view handed to on_columns is valid during the callback only
        host = _pinned(ctx, total)
        host[:total].copy_(dst[:total], non_blocking=True)
        offs = off.cpu().numpy().astype(np.uint64)      # synchronises: the copy above is complete
        framed = (host[:total].numpy(), offs)
        fs.compressed_bytes += total
    if on_columns:
        on_columns(cols, n_cols, framed)


def encode_file_resident(ctx, path, region="", sites_only=False, block_bytes=64 << 20, n_threads=0):
    """Whole file -> dense int8 G [S, V, 2] on the device plus tables on the host (used by the
    parse_vcf facade and by tests; large cohorts should stream with stream_file instead)."""
    cols, tabs = [], []

    def on_columns(G_cols, n_cols, framed):
        cols.append(G_cols.clone())

    def on_variants(start, ref, alt):
        tabs.append((start.copy(), ref.copy(), alt.copy()))

    vc = 1024
    fs = stream_file(ctx, path, region=region, sc=0, vc=vc, block_bytes=block_bytes, n_threads=n_threads,
                     sites_only=sites_only, on_columns=on_columns, on_variants=on_variants, compress=False)
    S, V = fs.n_samples, fs.n_kept
    if S and cols:
        # dense layout per column: [1][1][S][vc][2] -> concatenate columns along the variant axis
        parts = []
        for c in cols:
            n = c.numel() // (S * vc * 2)
            parts.append(c.view(torch.int8).view(n, S, vc, 2).permute(1, 0, 2, 3).reshape(S, n * vc, 2))
        G = torch.cat(parts, dim=1)[:, :V].contiguous()
    else:
        G = torch.zeros((S, V, 2), dtype=torch.int8, device=ctx.device)
    start = np.concatenate([t[0] for t in tabs]) if tabs else np.zeros(0, np.uint32)
    ref = np.concatenate([t[1] for t in tabs]) if tabs else np.zeros(0, np.uint8)
    alt = np.concatenate([t[2] for t in tabs]) if tabs else np.zeros(0, np.uint8)
    return G, start, ref, alt, fs
