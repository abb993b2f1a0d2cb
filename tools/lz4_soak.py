#!/usr/bin/env python3
"""Soak of the bit-plane LZ4 encoder: thousands of random planes (densities from 0.01x to 8x the bench's, block-periodic
and clustered ones among them) at every effort level; each stream must equal tools/sim/gapenc_ref.c byte for byte where
the reference takes the plane, and every chunk must decode back.  usage: python tools/lz4_soak.py [--exc] [planes_per_case]
(--exc: planes with missing calls from the bit-plane form, through the exception-aware instantiation)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 4096


def planes_to_tile_major(planes):
    """byte planes [2 rows][4096] (0 / 1 / 0xF7) -> (tile-major bit-plane bytes of one chunk column, the int8 bytes they stand
    for): include/hhgt.h "Bit-plane form" (the same packing tests/test_gpu_lz4_bitplanes.py uses)"""
    rows = planes.shape[0] // 2
    P = np.zeros((16, 4, rows, 32), np.uint8)
    for h in range(2):
        pl = planes[h::2]
        P[:, h] = np.packbits((pl != 0).reshape(rows, 16, 256), axis=2, bitorder="little").transpose(1, 0, 2)
        P[:, 2 + h] = np.packbits((pl == 0xF7).reshape(rows, 16, 256), axis=2, bitorder="little").transpose(1, 0, 2)
    raw = np.empty((rows, N, 2), np.uint8)
    raw[:, :, 0] = planes[0::2]
    raw[:, :, 1] = planes[1::2]
    return P.reshape(-1), raw.reshape(-1)


def soak_exc(per):
    """the exception-aware instantiation: planes with missing calls from the bit-plane form, nonzero counts from a handful up
    to beyond what its list holds (540), long runs of missing calls, at the default effort and at clevel 9 (lazy)"""
    import torch
    from haplohyped_varawareml_amd import device as dev
    from oracle import oracle
    so = os.path.join(tempfile.mkdtemp(), "libgapenc.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "sim", "gapenc_ref.c")])
    L = C.CDLL(so)
    L.gapenc_ref.restype = C.c_int
    L.gapenc_ref.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    ctx = dev.Context(0)
    rng = np.random.default_rng(4040)
    rows = 1
    while rows * 2 < per:
        rows *= 2
    total = same = handed = 0
    for clevel, depth in ((5, 2), (9, 12 | 0x100), (3, 1)):
        ctx.set_clevel(clevel)
        for scale in (0.05, 0.5, 1.0, 1.5, 1.9, 2.2, 3.0):
            for frac in (0.0005, 0.01, 0.05, 0.3):
                p = np.minimum((1.0 / 5008) * (0.5 * 5008) ** rng.random((2 * rows, N)) * scale, 0.5)
                planes = (rng.random((2 * rows, N)) < p).astype(np.uint8)
                miss = rng.random((2 * rows, N)) < frac * rng.random((2 * rows, 1))
                planes[miss] = 0xF7
                for k in range(0, 2 * rows, 9):      # runs of missing calls, a sample missing for a stretch
                    a = int(rng.integers(0, N - 600))
                    planes[k, a:a + int(rng.integers(5, 600))] = 0xF7
                P, raw = planes_to_tile_major(planes)
                lay = dev.make_layout(rows, N, sc=rows, vc=N)
                res = dev.EncodeResult(None, lay, None, None, None, None, 0, {}, [], torch.from_numpy(P).cuda())
                dst, off, tot = ctx.compress_planes(res, fmt=dev.BLOSC1)
                chunk = dst[:tot].cpu().numpy()
                assert np.array_equal(oracle.blosc_decompress(chunk), raw), (clevel, scale, frac)
                bst = chunk[16:16 + 4 * rows].view("<u4")
                for b in range(rows):
                    q = int(bst[b])
                    for h in range(2):
                        cs = int(chunk[q:q + 4].view("<u4")[0])
                        stream = chunk[q + 4:q + 4 + cs]
                        q += 4 + cs
                        plane = np.ascontiguousarray(planes[2 * b + h])
                        out = np.zeros(N + 64, np.uint8)
                        n = L.gapenc_ref(plane.ctypes.data, N, out.ctypes.data, depth)
                        total += 1
                        if n < 0 or n >= N:
                            handed += 1
                            continue
                        assert cs == n and np.array_equal(stream, out[:n]), (clevel, scale, frac, b, h, cs, n)
                        same += 1
            print(f"exc clevel {clevel} scale {scale}: ok", flush=True)
    print(f"{total} planes with missing calls: {same} byte-identical to the reference, {handed} handed to the byte-wise coder or stored; all chunks decode")


def main():
    if "--exc" in sys.argv:
        sys.argv.remove("--exc")
        return soak_exc(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
    import torch
    from haplohyped_varawareml_amd import device as dev
    from oracle import oracle
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    so = os.path.join(tempfile.mkdtemp(), "libgapenc.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "sim", "gapenc_ref.c")])
    L = C.CDLL(so)
    L.gapenc_ref.restype = C.c_int
    L.gapenc_ref.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    ctx = dev.Context(0)
    rng = np.random.default_rng(2026)
    lo = 1.0 / 5008
    total = same = handed = 0
    for clevel, depth in ((1, 0), (3, 1), (5, 2), (7, 4), (8, 8), (9, 12 | 0x100)):   # (clevel 9: twelve candidates + the lazy rule)
        for scale in (0.01, 0.1, 0.5, 1.0, 2.0, 4.0, 8.0):
            p = np.minimum(lo * (0.5 / lo) ** rng.random((per, N)) * scale, 0.5)
            planes = (rng.random((per, N)) < p).astype(np.uint8)
            # structure: some planes periodic, some with a repeated motif, some with a dense cluster
            for k in range(0, per, 7):
                period = int(rng.integers(2, 200))
                planes[k] = 0
                planes[k, ::period] = 1
            for k in range(3, per, 11):
                motif = (rng.random(int(rng.integers(20, 400))) < 0.1).astype(np.uint8)
                reps = N // len(motif)
                planes[k, :reps * len(motif)] = np.tile(motif, reps)
            for k in range(5, per, 13):
                a = int(rng.integers(0, N - 300))
                planes[k, a:a + 300] = (rng.random(300) < 0.5)
            # two planes per 8 KiB block (byte-shuffled: plane 0 = even bytes), 128 blocks per chunk
            blocks = np.stack([planes[0::2], planes[1::2]], axis=2).reshape(-1, 8192)    # [per/2, 8192] interleaved
            nblk = blocks.shape[0]
            pad = (-nblk) % 128
            data = np.concatenate([blocks, np.zeros((pad, 8192), np.uint8)]).reshape(-1)
            chunk = 128 * 8192
            ctx.set_clevel(clevel)
            src = torch.from_numpy(data).cuda()
            dst, off, tot = ctx.compress(src, chunk, typesize=2, blocksize=8192, fmt=dev.BLOSC1)
            back, bad = ctx.decompress(dst, off, data.size // chunk, chunk, typesize=2, blocksize=8192)
            assert bad == 0 and bool((back == src).all()), (clevel, scale)
            d = dst[:tot].cpu().numpy()
            offs = off.cpu().numpy()
            for c in range(len(offs) - 1):
                ch = d[int(offs[c]):int(offs[c + 1])]
                bst = ch[16:16 + 4 * 128].view("<u4")
                for b in range(128):
                    gb = c * 128 + b
                    if gb >= nblk:
                        break
                    q = int(bst[b])
                    for pl in range(2):
                        cs = int(ch[q:q + 4].view("<i4")[0])
                        stream = ch[q + 4:q + 4 + cs]
                        q += 4 + cs
                        plane = planes[2 * gb + pl]
                        out = np.zeros(N + 64, np.uint8)
                        n = L.gapenc_ref(plane.ctypes.data, N, out.ctypes.data, depth)
                        total += 1
                        if n < 0 or n >= N:
                            handed += 1
                            continue
                        assert cs == n and np.array_equal(stream, out[:n]), (clevel, scale, gb, pl, cs, n)
                        same += 1
            print(f"clevel {clevel} scale {scale}: ok ({data.size / tot:.2f}x)", flush=True)
    print(f"{total} planes: {same} byte-identical to the reference, {handed} handed to the byte-wise coder or stored; all chunks decode")


if __name__ == "__main__":
    main()
