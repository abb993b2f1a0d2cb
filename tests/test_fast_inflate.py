"""CPU: the host DEFLATE decoder of the BGZF reader (csrc/fast_inflate.h) against zlib — every kind of stream zlib
writes, corrupted and truncated streams, under AddressSanitizer / UBSan (tests/fast_inflate_fuzz.cpp), and through the
library's own entry point."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz")
    for flags in (["-mssse3"], []):      # the pshufb copy path the library is built with, and the baseline x86-64 one
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"] + flags +
                              ["-o", exe, os.path.join(ROOT, "tests", "fast_inflate_fuzz.cpp"), "-lz"])
        r = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_library_entry_point_and_reader_use_it(tmp_path):
    from haplohyped_varawareml_amd import _lib, synth
    from haplohyped_varawareml_amd.reader import VcfReader, write_bgzf
    L = _lib.load()
    L.hhgt_fast_inflate.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    tab = synth.variant_table(3, 40, 2504)
    text, _ = synth.render_fixed_numpy("chr3", tab, 2504, seed=3)
    for level in (1, 6, 9):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        piece = text[:65280]
        comp = co.compress(piece) + co.flush()
        src = np.frombuffer(comp, dtype=np.uint8)
        out = np.zeros(len(piece), np.uint8)
        assert L.hhgt_fast_inflate(src.ctypes.data, src.size, out.ctypes.data, out.size) == 0
        assert out.tobytes() == piece
        assert L.hhgt_fast_inflate(src.ctypes.data, src.size - 3, out.ctypes.data, out.size) != 0
    # the reader gives the same text with either inflater (HHGT_ZLIB_INFLATE is read once per process: a child each)
    p = str(tmp_path / "a.vcf.gz")
    write_bgzf(p, text, level=6)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from haplohyped_varawareml_amd.reader import VcfReader\n"
            "import hashlib\n"
            "h = hashlib.sha256()\n"
            "with VcfReader(%r, block_bytes=1 << 20, n_threads=3) as r:\n"
            "    for b in r: h.update(bytes(b))\n"
            "print(h.hexdigest())\n") % (ROOT, p)
    import hashlib
    import sys
    want = hashlib.sha256(text).hexdigest()
    for z in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HHGT_ZLIB_INFLATE=z), capture_output=True, text=True, timeout=300)
        assert out.stdout.strip() == want, out.stderr
