"""TEST INFRASTRUCTURE ONLY: an oracle-backed stand-in for pipeline.stream_file, so that the CPU suite (no GPU in
the container) can drive the converter's rank / plan / merge logic through the real VCFtoHDF5Converter code.  The
product never imports this module (tests/test_abi.py greps for that); on a GPU the same tests run with the real
stream_file (tests/test_gpu_pipeline.py::test_converter_two_workers_one_gpu)."""
import gzip
import time
from types import SimpleNamespace

import numpy as np

from oracle import oracle


class FakeCtx:
    def __init__(self, device=0):
        self.device = device

    def close(self):
        pass


def _read_text(path):
    raw = open(path, "rb").read()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw      # BGZF is multi-member gzip


def stream_files(ctx, jobs, sc=64, vc=8192, block_bytes=None, n_threads=0, sites_only=False, fmt=oracle.BLOSC2,
                 device_inflate=None, on_header=None, on_variants=None, on_columns=None, on_end=None, files_ahead=1):
    """same interface as pipeline.stream_files"""
    out = []
    for i, (path, region) in enumerate(jobs):
        fs = stream_file(ctx, path, region=region, sc=sc, vc=vc, fmt=fmt,
                         on_header=(lambda names, i=i: on_header(i, names)) if on_header else None,
                         on_variants=(lambda a, b, c, i=i: on_variants(i, a, b, c)) if on_variants else None,
                         on_columns=(lambda g, n, f, i=i: on_columns(i, g, n, f)) if on_columns else None)
        if on_end:
            on_end(i, fs)
        out.append(fs)
    return out


def stream_file(ctx, path, region="", sc=64, vc=8192, block_bytes=None, n_threads=0, sites_only=False, on_columns=None,
                on_variants=None, on_header=None, compress=True, fmt=oracle.BLOSC2, device_inflate=None):
    t0 = time.perf_counter()
    text = _read_text(path)
    names = oracle.header_samples(text)
    S = len(names)
    if on_header:
        on_header(names)
    o = oracle.vcf_encode(text, S, region=region, want_chrom=True)
    V = o["n_kept"]
    if on_variants and V:
        on_variants(o["start"], o["ref"], o["alt"])
    n_vcol, n_scol = -(-V // vc), -(-max(S, 1) // sc)
    tiled = np.zeros((n_vcol, n_scol, sc, vc, 2), np.int8)       # the chunk-tiled layout of include/hhgt.h
    G = o["G"]
    for v in range(n_vcol):
        w = min(vc, V - v * vc)
        for s in range(n_scol):
            h = min(sc, S - s * sc)
            tiled[v, s, :h, :w] = G[s * sc:s * sc + h, v * vc:v * vc + w]
    raw_bytes = comp_bytes = 0
    chunk_nbytes = sc * vc * 2
    for v in range(n_vcol):
        frames = [oracle.blosc_compress(tiled[v, s].reshape(-1).view(np.uint8), 2, min(vc * 2, 8192), fmt) for s in range(n_scol)]
        offs = np.zeros(n_scol + 1, np.uint64)
        offs[1:] = np.cumsum([f.size for f in frames])
        raw_bytes += n_scol * chunk_nbytes
        comp_bytes += int(offs[-1])
        if on_columns:
            on_columns(SimpleNamespace(numel=lambda: n_scol * chunk_nbytes), 1, (np.concatenate(frames), offs))
    runs, last = [], None
    for i, c in enumerate(o["chrom"]):
        if c != last:
            runs.append((i, c))
            last = c
    return SimpleNamespace(n_kept=V, n_samples=S, n_lines=o["stats"]["n_lines"], seconds=time.perf_counter() - t0,
                           raw_bytes=raw_bytes, compressed_bytes=comp_bytes, chrom_runs=runs, samples=names)
