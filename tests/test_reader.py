"""CPU: the host reader (C++: mmap + zlib, BGZF blocks inflated in parallel) delivers exactly the
file's text, cut at line boundaries, for plain / gzip / multi-member gzip / BGZF inputs."""
import gzip
import os

import numpy as np
import pytest

from haplohyped_varawareml_amd import synth
from haplohyped_varawareml_amd.reader import VcfReader, parse_header, write_bgzf


def collect(path, **kw):
    out = []
    with VcfReader(path, **kw) as r:
        bg = r.is_bgzf
        for b in r:
            assert b.size > 0
            out.append(bytes(b))
    return out, bg


@pytest.fixture(scope="module")
def text():
    tab = synth.variant_table(3, 3000, 200)
    t, _ = synth.render_fixed_numpy("chr3", tab, 200, seed=3)
    return t      # ~2.5 MB


@pytest.mark.parametrize("kind", ["plain", "gzip", "gzip2", "bgzf"])
@pytest.mark.parametrize("block", [1 << 20, 8 << 20])
def test_roundtrip(tmp_path, text, kind, block):
    p = str(tmp_path / f"x.{kind}")
    if kind == "plain":
        open(p, "wb").write(text)
    elif kind == "gzip":
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(text)
    elif kind == "gzip2":   # concatenated members (bgzip-less tools do this)
        cut = len(text) // 3
        with open(p, "wb") as f:
            f.write(gzip.compress(text[:cut], 1))
            f.write(gzip.compress(text[cut:], 1))
    else:
        write_bgzf(p, text)
    blocks, bg = collect(p, block_bytes=block, n_threads=4)
    assert bg == (kind == "bgzf")
    assert b"".join(blocks) == text
    assert all(b.endswith(b"\n") for b in blocks)
    if block == 1 << 20:
        assert len(blocks) >= 2


def test_no_trailing_newline_and_empty(tmp_path):
    p = str(tmp_path / "a.vcf")
    open(p, "wb").write(b"#x\nline1\nline2")
    blocks, _ = collect(p)
    assert b"".join(blocks) == b"#x\nline1\nline2"
    open(p, "wb").write(b"")
    assert collect(p)[0] == []


def test_reference_fixture_header(golden_dir, fixture_golden):
    with VcfReader(os.path.join(golden_dir, "chr22.filtered.vcf.gz")) as r:
        assert not r.is_bgzf           # the reference's fixture is plain gzip
        b = r.next_block()
        names, hdr = parse_header(b)
        assert names == fixture_golden["samples"]
        assert bytes(b[hdr:hdr + 5]) == b"chr22" and b.size == fixture_golden["text_bytes"]
        assert r.next_block() is None
        assert r.stats()["text_bytes"] == fixture_golden["text_bytes"]


def test_errors(tmp_path):
    from haplohyped_varawareml_amd._lib import HhgtError
    with pytest.raises(HhgtError):
        VcfReader(str(tmp_path / "missing.vcf.gz"))
    p = str(tmp_path / "trunc.vcf.gz")
    open(p, "wb").write(gzip.compress(b"abc\n" * 100000)[:-200])
    with pytest.raises(HhgtError, match="truncated|inflate"):
        collect(p)
    q = str(tmp_path / "long.vcf")
    open(q, "wb").write(b"x" * (3 << 20))
    with pytest.raises(HhgtError, match="longer than"):
        collect(q, block_bytes=1 << 20)


def test_bgzf_crc_mismatch_is_an_error(tmp_path, monkeypatch):
    """htslib's bgzf.c fails a member whose text does not hash to the trailer CRC32; so does the host reader
    (HHGT_BGZF_NO_CRC=1 skips the check)"""
    from haplohyped_varawareml_amd._lib import HhgtError
    text = b"".join(b"chr1\t%d\t.\tA\tG\t.\t.\t.\tGT\t0|1\n" % i for i in range(20000))
    p = str(tmp_path / "c.vcf.gz")
    write_bgzf(p, text, block_size=20000)
    raw = bytearray(open(p, "rb").read())
    assert b"".join(collect(p)[0]) == text and collect(p)[1]
    bsize = raw[16] | (raw[17] << 8)
    raw[bsize + 1 - 8] ^= 0x40                 # CRC32 field of the first member
    open(p, "wb").write(bytes(raw))
    with pytest.raises(HhgtError, match="CRC32"):
        collect(p)
    monkeypatch.setenv("HHGT_BGZF_NO_CRC", "1")
    assert b"".join(collect(p)[0]) == text


def test_crc32_matches_zlib():
    """hhgt_crc32 (PCLMULQDQ folding + table tail) against zlib's crc32 at every alignment / length class"""
    import ctypes as C
    import zlib
    from haplohyped_varawareml_amd import _lib
    L = _lib.load()
    L.hhgt_crc32.restype = C.c_uint32
    L.hhgt_crc32.argtypes = [C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, 70000, dtype=np.uint8)
    for n in list(range(0, 200)) + [255, 256, 257, 1023, 4096, 65280, 65535, 65536, 69999]:
        for off in (0, 1, 7):
            a = np.ascontiguousarray(big[off:off + n])
            assert L.hhgt_crc32(a.ctypes.data, a.size) == (zlib.crc32(a.tobytes()) & 0xFFFFFFFF), (n, off)
    z = np.zeros(65536, np.uint8)
    assert L.hhgt_crc32(z.ctypes.data, z.size) == (zlib.crc32(z.tobytes()) & 0xFFFFFFFF)


@pytest.mark.parametrize("threads,blocks,block_kb", [(1, 2, 1024), (8, 3, 1024), (16, 6, 2048)])
def test_bgzf_many_blocks_in_flight(tmp_path, threads, blocks, block_kb):
    """the barrier-free BGZF path: many ring blocks open to the workers at once, held blocks released out of
    order, every byte and every line boundary intact"""
    import ctypes as C
    from haplohyped_varawareml_amd import reader as rd
    rng = np.random.default_rng(11)
    lines = [b"chr1\t%d\t.\tA\tC\t" % i + bytes(rng.integers(48, 50, int(rng.integers(10, 9000)), dtype=np.uint8)) + b"\n"
             for i in range(6000)]
    text = b"".join(lines)
    p = str(tmp_path / "x.vcf.gz")
    rd.write_bgzf(p, text, level=1)
    L = rd._lib_reader()
    L.hhgt_reader_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hhgt_reader_release.argtypes = [C.c_void_p, C.c_int]
    h = C.c_void_p()
    rd.check(L.hhgt_reader_open(p.encode(), block_kb << 10, threads, blocks, C.byref(h)))
    got, held, saw_last = [], [], False
    while True:
        ptr, n, tok, last = C.c_void_p(), C.c_uint64(0), C.c_int(-1), C.c_int(0)
        rd.check(L.hhgt_reader_acquire(h, C.byref(ptr), C.byref(n), C.byref(tok), C.byref(last)))
        if n.value == 0:
            break
        blk = C.string_at(ptr, n.value)
        assert not saw_last
        saw_last = bool(last.value)
        assert blk.endswith(b"\n")
        got.append(blk)
        held.append(tok.value)
        if len(held) >= blocks - 1:                     # keep several blocks, give the newest back first
            rd.check(L.hhgt_reader_release(h, held.pop()))
    for t in held:
        rd.check(L.hhgt_reader_release(h, t))
    L.hhgt_reader_close(h)
    assert saw_last and b"".join(got) == text and len(got) > 3


def test_bgzf_crc_mismatch_is_reported(tmp_path):
    from haplohyped_varawareml_amd import reader as rd
    text = b"".join(b"chr1\t%d\t.\tA\tC\tzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz\n" % i for i in range(20000))
    p = str(tmp_path / "y.vcf.gz")
    rd.write_bgzf(p, text, level=6)
    raw = bytearray(open(p, "rb").read())
    bsize = raw[16] | (raw[17] << 8)
    raw[bsize + 1 - 8] ^= 0x55                           # CRC field of the first member
    open(p, "wb").write(raw)
    with pytest.raises(Exception, match="CRC32 checksum mismatch"):
        with rd.VcfReader(p, block_bytes=1 << 20, n_threads=4) as r:
            for _ in r:
                pass


def test_peek_sample_count(tmp_path):
    """pipeline.peek_sample_count: the #CHROM line of a plain or gzip / BGZF file -> number of sample columns (what the ingest
    engine is told as expect_samples before it opens)"""
    import gzip
    from haplohyped_varawareml_amd.pipeline import peek_sample_count
    hdr = b"##fileformat=VCFv4.2\n##contig=<ID=chr1>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + \
        b"\t".join(b"s%d" % i for i in range(37)) + b"\nchr1\t5\t.\tA\tC\t.\t.\t.\tGT\t" + b"\t".join([b"0|1"] * 37) + b"\n"
    p = tmp_path / "a.vcf"
    p.write_bytes(hdr)
    assert peek_sample_count(str(p)) == 37
    q = tmp_path / "a.vcf.gz"
    with gzip.open(q, "wb") as f:
        f.write(hdr)
    assert peek_sample_count(str(q)) == 37
    sites = tmp_path / "sites.vcf"
    sites.write_bytes(b"##x\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\nchr1\t5\t.\tA\tC\t.\t.\t.\n")
    assert peek_sample_count(str(sites)) == 0
    assert peek_sample_count(str(tmp_path / "missing.vcf")) == 0
    nohdr = tmp_path / "n.vcf"
    nohdr.write_bytes(b"chr1\t5\t.\tA\tC\t.\t.\t.\n")
    assert peek_sample_count(str(nohdr)) == 0
