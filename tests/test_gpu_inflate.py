"""-m gpu: BGZF members inflated on the device (`hhgt_inflate_members`) against zlib.
Bar: byte-exact text for every member; stored, fixed-Huffman and dynamic-Huffman blocks, empty members,
64 KiB members, overlapping matches of every short distance; corrupt members are flagged, not crashed on."""
import struct
import zlib

import numpy as np
import pytest
import torch

from haplohyped_varawareml_amd import device as dev, synth
from haplohyped_varawareml_amd.reader import write_bgzf

pytestmark = pytest.mark.gpu


def bgzf(chunks, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    out = []
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        comp = co.compress(c) + co.flush()
        assert len(comp) + 26 <= 65536
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                   struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
    return b"".join(out)


def split(data, n):
    return [data[i:i + n] for i in range(0, len(data), n)]


def check(ctx, raw, want):
    text, bad, status = ctx.inflate_bgzf(raw, return_status=True)
    assert bad == 0, status.cpu().numpy()[status.cpu().numpy() != 0][:8]
    got = text.cpu().numpy().tobytes()
    assert len(got) == len(want)
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        i = int(np.flatnonzero(a != b)[0])
        raise AssertionError(f"first difference at byte {i} of {len(want)}: got {got[i-8:i+8]!r} want {want[i-8:i+8]!r}")


def vcf_like(n_lines, S, seed):
    rng = np.random.default_rng(seed)
    lines = [b"##fileformat=VCFv4.2\n", b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%d" % i for i in range(S)) + b"\n"]
    calls = np.array([b"0|0", b"0|1", b"1|0", b"1|1", b".|."])
    for i in range(n_lines):
        g = calls[rng.choice(5, size=S, p=[0.9, 0.04, 0.04, 0.015, 0.005])]
        lines.append(b"chr1\t%d\trs%d\tA\tG\t.\tPASS\t.\tGT\t" % (1000 + 37 * i, i) + b"\t".join(g) + b"\n")
    return b"".join(lines)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_vcf_text_members(ctx, level):
    text = vcf_like(400, 500, 5)
    check(ctx, bgzf(split(text, 0xFF00), level), text)


def test_fixed_huffman_and_stored_blocks(ctx):
    text = vcf_like(60, 300, 6)
    check(ctx, bgzf(split(text, 20000), 6, zlib.Z_FIXED), text)             # BTYPE 1
    check(ctx, bgzf(split(text, 20000), 0), text)                          # BTYPE 0 (level 0: stored blocks only)
    rnd = bytes(np.random.default_rng(1).integers(0, 256, size=150_000, dtype=np.uint8))
    check(ctx, bgzf(split(rnd, 60000), 6), rnd)                             # zlib falls back to stored blocks
    check(ctx, bgzf(split(rnd, 60000), 6, zlib.Z_HUFFMAN_ONLY), rnd)        # literals only, 8/9-bit codes


def test_overlapping_matches_every_distance(ctx):
    rng = np.random.default_rng(8)
    parts = []
    for d in list(range(1, 70)) + [127, 128, 129, 255, 256, 1000, 4097, 32768]:
        unit = bytes(rng.integers(65, 91, size=d, dtype=np.uint8))
        parts.append((unit * (700 // d + 3))[:700 + d])
        parts.append(bytes(rng.integers(0, 256, size=13, dtype=np.uint8)))
    text = b"".join(parts)
    check(ctx, bgzf(split(text, 65000), 9), text)


def test_member_shapes(ctx):
    chunks = [b"", b"x", b"ab" * 3, b"\n" * 65536, bytes(range(256)) * 256, b"", vcf_like(5, 20, 2)]
    assert len(chunks[3]) == 65536 and len(chunks[4]) == 65536
    check(ctx, bgzf(chunks), b"".join(chunks))
    many = [b"line %d\n" % i for i in range(3000)]                          # more members than one launch wave-set
    check(ctx, bgzf(many, 6), b"".join(many))


def test_writer_file_roundtrip(ctx, tmp_path):
    tab = synth.variant_table(11, 2000, 64)
    text_dev, _ = ctx.synth_fixed("chr1", tab, 64, seed=11)
    text = text_dev.cpu().numpy().tobytes()
    path = tmp_path / "s.vcf.gz"
    write_bgzf(str(path), text)
    check(ctx, path.read_bytes(), text)


def test_corrupt_members_are_flagged(ctx):
    text = vcf_like(300, 400, 9)
    raw = bytearray(bgzf(split(text, 0xFF00), 6))
    tab = dev.bgzf_scan(bytes(raw))
    n = len(tab["isize"])
    assert n >= 3
    rng = np.random.default_rng(4)
    for m in (0, n // 2):                                                    # garble two payloads, keep the framing
        o, l = int(tab["comp_off"][m]), int(tab["comp_len"][m])
        raw[o + 20:o + l] = bytes(rng.integers(0, 256, size=l - 20, dtype=np.uint8))
    out, bad, status = ctx.inflate_bgzf(bytes(raw), return_status=True)
    st = status.cpu().numpy()
    assert bad >= 1 and set(np.flatnonzero(st)) <= {0, n // 2}
    # the untouched members are still right
    off = np.concatenate([[0], np.cumsum(tab["isize"])]).astype(np.int64)
    got = out.cpu().numpy().tobytes()
    for m in range(n):
        if st[m] == 0 and m not in (0, n // 2):
            assert got[off[m]:off[m + 1]] == text[off[m]:off[m + 1]]


def test_stream_file_device_inflate_matches_host_inflate(ctx, tmp_path):
    """the converter's streaming path with BGZF inflated on the device == the same file through the host reader
    (several text blocks, lines carried across block and member boundaries)"""
    from haplohyped_varawareml_amd import pipeline
    S = 300
    tab = synth.variant_table(21, 9000, S)
    text_dev, _ = ctx.synth_fixed("chr1", tab, S, seed=21)
    path = tmp_path / "c.vcf.gz"
    write_bgzf(str(path), text_dev.cpu().numpy().tobytes())
    res = {}
    for mode in (False, True):
        cols, tabs = [], []
        fs = pipeline.stream_file(ctx, str(path), sc=64, vc=512, block_bytes=6 << 20, compress=False, device_inflate=mode,
                                  on_columns=lambda G, n, framed: cols.append(G.cpu().numpy().copy()),
                                  on_variants=lambda st, r, a: tabs.append((st.copy(), r.copy(), a.copy())))
        res[mode] = (fs, np.concatenate(cols), np.concatenate([t[0] for t in tabs]))
    (fh, Gh, sh), (fd, Gd, sd) = res[False], res[True]
    assert fd.is_bgzf and fd.n_kept == fh.n_kept == 9000 and fd.n_lines == fh.n_lines and fd.text_bytes == fh.text_bytes
    assert np.array_equal(Gh, Gd) and np.array_equal(sh, sd)


def test_stream_file_device_inflate_accepts_a_long_header(ctx, tmp_path):
    """a header of more than 8 MiB (tens of thousands of ##contig lines): the device-inflate path finds its end on the
    device and hands all of it to the header parser, as the host path does (up to 64 MiB)"""
    from haplohyped_varawareml_amd import pipeline
    S = 40
    tab = synth.variant_table(3, 600, S)
    text, _ = synth.render_fixed_numpy("chr3", tab, S, seed=3)
    text = bytes(text)
    first_nl = text.index(b"\n") + 1
    contigs = b"".join(b"##contig=<ID=scaffold_%07d,length=%d,assembly=an_assembly_with_a_long_name_%060d>\n" % (i, 1000 + i, i)
                       for i in range(80000))
    assert len(contigs) > (9 << 20)
    big = text[:first_nl] + contigs + text[first_nl:]
    path = tmp_path / "h.vcf.gz"
    write_bgzf(str(path), big, level=1)
    out = {}
    for mode in (False, True):
        cols = []
        fs = pipeline.stream_file(ctx, str(path), sc=64, vc=512, block_bytes=16 << 20, compress=False, device_inflate=mode,
                                  on_columns=lambda G, n, framed: cols.append(G.cpu().numpy().copy()))
        out[mode] = (fs.n_kept, np.concatenate(cols))
    assert out[True][0] == out[False][0] == 600
    assert np.array_equal(out[True][1], out[False][1])
    # the ingest engine (compressed output) takes the same file with either inflater
    for mode in (False, True):
        fs = pipeline.stream_file(ctx, str(path), sc=64, vc=512, compress=True, device_inflate=mode)
        assert fs.n_kept == 600 and fs.n_samples == S


def test_fuzzed_members_terminate_and_stay_in_bounds(ctx):
    """400 members, each garbled differently (bit flips, random runs, truncated tables): every wave must finish, flag or
    decode its member, and never write outside the member's slice (guard bytes around the output stay intact)."""
    rng = np.random.default_rng(2024)
    chunks = [vcf_like(12, 100 + (i % 7) * 30, 100 + i) for i in range(400)]
    raw = bytearray(bgzf(chunks, 6))
    tab = dev.bgzf_scan(bytes(raw))
    n = len(tab["isize"])
    touched = set()
    for m in range(0, n, 2):
        o, l = int(tab["comp_off"][m]), int(tab["comp_len"][m])
        kind = m % 6
        if kind == 0:
            for _ in range(3):
                raw[o + int(rng.integers(0, l))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:
            a = int(rng.integers(0, l - 4))
            raw[o + a:o + l] = bytes(rng.integers(0, 256, size=l - a, dtype=np.uint8))
        else:
            raw[o:o + 12] = bytes(rng.integers(0, 256, size=12, dtype=np.uint8))      # block header + code tables
        touched.add(m)
    out, bad, status = ctx.inflate_bgzf(bytes(raw), return_status=True)
    st = status.cpu().numpy()
    assert set(np.flatnonzero(st).tolist()) <= touched and bad == int((st != 0).sum()) and bad > len(touched) // 2
    got = out.cpu().numpy().tobytes()
    off = np.concatenate([[0], np.cumsum(tab["isize"])]).astype(np.int64)
    for m in range(1, n, 2):                                                        # neighbours of garbled members
        assert got[off[m]:off[m + 1]] == chunks[m], m


def test_converter_with_device_inflate_writes_the_same_h5(ctx, tmp_path, golden_dir, monkeypatch):
    """vcf_to_h5 with HHGT_DEVICE_INFLATE=1: BGZF shards go through the device inflater, the plain-gzip fixture keeps the
    host reader; the cohort file is byte-identical to the one written with the host inflater"""
    import os
    import shutil
    from haplohyped_varawareml_amd.vcf_to_h5 import VCFtoHDF5Converter
    vcf_dir = tmp_path / "vcf"
    vcf_dir.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr22.filtered.vcf.gz"), vcf_dir / "chr22.filtered.vcf.gz")
    names = [l.strip() for l in open(os.path.join(golden_dir, "ipscs_samples_test.txt")) if l.strip()]
    write_bgzf(str(vcf_dir / "chr4.filtered.vcf.gz"), synth.render_mixed("chr4", 5000, len(names), seed=4, names=names))
    files = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("HHGT_DEVICE_INFLATE", mode)
        conv = VCFtoHDF5Converter("c", str(vcf_dir), str(tmp_path / f"out{mode}"),
                                  os.path.join(golden_dir, "ipscs_samples_test.txt"), cores=2, cxx_threads=1)
        conv.run()
        assert conv.stats["chr_4"].is_bgzf and conv.stats["chr_4"].n_kept > 1000
        files[mode] = open(conv.h5_path, "rb").read()
    assert files["0"] == files["1"]


def test_crc32_of_the_text_is_checked(ctx):
    """htslib rejects a member whose inflated bytes do not match the trailer CRC-32 (bgzf.c); so does the device path:
    a flipped trailer, and a flipped byte inside a stored block (inflates fine, to different text), are status 9"""
    rng = np.random.default_rng(12)
    chunks = [vcf_like(30, 200, 40 + i) for i in range(6)] + [b"", bytes(rng.integers(0, 256, size=60000, dtype=np.uint8)), b"z", b"q" * 4096, b"r" * 4097, b"s" * 8191, bytes(range(256)) * 17]
    raw = bytearray(bgzf(chunks[:4], 6) + bgzf(chunks[4:6], 0) + bgzf(chunks[6:], 6))
    tab = dev.bgzf_scan(bytes(raw))
    assert [int(c) for c in tab["crc32"]] == [zlib.crc32(c) for c in chunks]
    text, bad = ctx.inflate_bgzf(bytes(raw))
    assert bad == 0 and text.cpu().numpy().tobytes() == b"".join(chunks)
    o1 = int(tab["comp_off"][1]) + int(tab["comp_len"][1])                 # member 1: its CRC field
    raw[o1] ^= 0x10
    o5 = int(tab["comp_off"][5]) + 5 + 1000                                # member 5: a byte of the stored block's data
    raw[o5] ^= 0x01
    text, bad, status = ctx.inflate_bgzf(bytes(raw), return_status=True)
    st = status.cpu().numpy().tolist()
    assert bad == 2 and st[1] == 9 and st[5] == 9 and sum(1 for x in st if x) == 2
    _, bad = ctx.inflate_bgzf(bytes(raw), check_crc=False)
    assert bad == 0
