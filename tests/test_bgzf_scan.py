"""BGZF member table on the host (`hhgt_bgzf_scan`): no GPU needed.  Members are made with
the package's minimal BGZF writer and with hand-built headers (extra subfields before "BC")."""
import struct
import zlib

import numpy as np
import pytest

from haplohyped_varawareml_amd import device as dev
from haplohyped_varawareml_amd._lib import HhgtError
from haplohyped_varawareml_amd.reader import write_bgzf


def member(payload, extra_before=b""):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(payload) + co.flush()
    xlen = len(extra_before) + 6
    bsize = 12 + xlen + len(comp) + 8 - 1
    head = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra_before + b"BC\x02\0" + struct.pack("<H", bsize)
    return head + comp + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)), comp


def test_scan_matches_writer(tmp_path):
    rng = np.random.default_rng(3)
    text = bytes(rng.integers(48, 58, size=300_000, dtype=np.uint8))
    path = tmp_path / "a.vcf.gz"
    write_bgzf(str(path), text, block_size=0xFF00)
    raw = path.read_bytes()
    tab = dev.bgzf_scan(raw)
    assert tab["consumed"] == len(raw)
    assert int(tab["isize"].sum()) == len(text)
    assert tab["isize"][-1] == 0 and tab["comp_len"][-1] == 2          # the EOF marker member
    assert int(tab["crc32"][0]) == zlib.crc32(text[:0xFF00]) and int(tab["crc32"][-1]) == 0
    out = b"".join(zlib.decompress(raw[int(o):int(o) + int(l)], -15) for o, l in zip(tab["comp_off"], tab["comp_len"]))
    assert out == text


def test_scan_extra_subfields_and_cut_member():
    a, ca = member(b"hello hello hello\n" * 50, extra_before=b"XY\x03\0abc")
    b, cb = member(b"second member\n")
    raw = a + b
    tab = dev.bgzf_scan(raw)
    assert list(tab["isize"]) == [900, 14] and tab["consumed"] == len(raw)
    assert raw[int(tab["comp_off"][0]):][:len(ca)] == ca and int(tab["comp_len"][1]) == len(cb)
    cut = dev.bgzf_scan(raw[:-5])
    assert list(cut["isize"]) == [900] and cut["consumed"] == len(a)     # the cut member waits for more input
    assert len(dev.bgzf_scan(b"")["isize"]) == 0


def test_scan_rejects_non_bgzf():
    import gzip
    with pytest.raises(HhgtError):
        dev.bgzf_scan(gzip.compress(b"plain gzip has no BC subfield" * 10))
    with pytest.raises(HhgtError):
        dev.bgzf_scan(b"not gzip at all, but long enough to hold a header")
