"""-m gpu: HIP shuffle + LZ4 + Blosc framing (through the C ABI) against the CPU oracle's decoder,
the image's independent decoders (liblz4 / c-blosc 1.21 when loadable) and the GPU decoder.
Bar: every chunk decompresses to the identical bytes (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from haplohyped_varawareml_amd import device as dev
from tests import extlibs
from tests.gpu_util import split_chunks, to_dev
from tests.test_oracle_codec import BLOSC_CASES, cases, genotype_like

pytestmark = pytest.mark.gpu


def roundtrip(ctx, data, chunk_nbytes, typesize, blocksize, fmt):
    src = to_dev(data)
    n_chunks = data.size // chunk_nbytes
    dst, off, total = ctx.compress(src, chunk_nbytes, typesize=typesize, blocksize=blocksize, fmt=fmt)
    chunks = split_chunks(dst, off, total)
    assert len(chunks) == n_chunks
    hl = 32 if fmt == dev.BLOSC2 else 16
    for i, ck in enumerate(chunks):
        want = data[i * chunk_nbytes:(i + 1) * chunk_nbytes]
        assert ck.size <= chunk_nbytes + hl
        assert int(ck[4:8].view("<u4")[0]) == chunk_nbytes and int(ck[12:16].view("<u4")[0]) == ck.size
        assert np.array_equal(oracle.blosc_decompress(ck), want), f"chunk {i}: oracle decode differs"
        if fmt == dev.BLOSC1 and extlibs.have_blosc():
            assert np.array_equal(extlibs.blosc1_decompress(ck, chunk_nbytes), want), f"chunk {i}: c-blosc decode differs"
    back, bad = ctx.decompress(dst, off, n_chunks, chunk_nbytes, typesize=typesize, blocksize=blocksize)
    assert bad == 0
    assert np.array_equal(back.cpu().numpy(), data[:n_chunks * chunk_nbytes])
    return chunks


@pytest.mark.parametrize("typesize,blocksize,nbytes", BLOSC_CASES)
@pytest.mark.parametrize("fmt", [dev.BLOSC1, dev.BLOSC2])
def test_chunk_shapes(ctx, typesize, blocksize, nbytes, fmt):
    n_chunks = 3
    data = genotype_like(1, n_chunks * nbytes // 2 + 1, nbytes).reshape(-1).view(np.uint8)[:n_chunks * nbytes].copy()
    roundtrip(ctx, data, nbytes, typesize, blocksize, fmt)


@pytest.mark.parametrize("name,data", [c for c in cases() if c[1].size >= 16])
def test_lz4_streams_via_liblz4(ctx, name, data):
    """typesize 1, one block: the chunk's single stream is a bare LZ4 block -> decode it with liblz4"""
    n = data.size
    blocksize = min(n, 65536)
    chunks = roundtrip(ctx, data[: n - n % 1], n, 1, blocksize, dev.BLOSC1)
    ck = chunks[0]
    if ck[2] & 0x2:      # memcpyed
        return
    nblocks = -(-n // blocksize)
    bst = ck[16:16 + 4 * nblocks].view("<u4")
    for b in range(nblocks):
        bs = min(blocksize, n - b * blocksize)
        cs = int(ck[bst[b]:bst[b] + 4].view("<u4")[0])
        stream = ck[bst[b] + 4: bst[b] + 4 + cs]
        want = data[b * blocksize: b * blocksize + bs]
        if cs == bs:
            assert np.array_equal(stream, want)
        else:
            assert np.array_equal(oracle.lz4_decompress(stream, bs), want)
            if extlibs.have_lz4():
                assert np.array_equal(extlibs.lz4_decompress(stream, bs), want)


def test_incompressible_is_memcpyed(ctx):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 2 * 65536, dtype=np.uint8)
    for fmt, hl in ((dev.BLOSC1, 16), (dev.BLOSC2, 32)):
        chunks = roundtrip(ctx, data, 65536, 2, 32768, fmt)
        assert all(c.size == 65536 + hl and (c[2] & 0x2) for c in chunks)


def test_genotype_chunk_geometry_and_ratio(ctx):
    """the production geometry: chunk = 64 samples x 16384 variants x 2 (2 MiB), block = one sample
    row (32 KiB), typesize 2 -> two 16 KiB planes per block"""
    G = genotype_like(64 * 3, 16384, 11)
    data = np.ascontiguousarray(G.reshape(3, 64, 16384, 2)).reshape(-1).view(np.uint8)
    chunks = roundtrip(ctx, data, 64 * 16384 * 2, 2, 32768, dev.BLOSC2)
    ck = chunks[0]
    assert ck[0] == 5 and ck[1] == 1 and ck[2] == (0x1 | 0x4 | 0x20) and ck[3] == 2
    assert bytes(ck[16:22]) == b"\0\0\0\0\0\x01" and bytes(ck[22:32]) == b"\0" * 10
    bst = ck[32:32 + 4 * 64].view("<u4")
    assert bst[0] == 32 + 4 * 64 and np.all(np.diff(bst.astype(np.int64)) > 0)
    ratio = data.size / sum(c.size for c in chunks)
    cpu_ratio = data.size / sum(oracle.blosc_compress(data[i * (1 << 21):(i + 1) * (1 << 21)], 2, 32768).size for i in range(3))
    assert ratio > 2.0 and ratio > 0.8 * cpu_ratio, (ratio, cpu_ratio)


def test_gpu_decoder_reads_oracle_chunks(ctx):
    G = genotype_like(8, 4096, 5)
    data = G.reshape(-1).view(np.uint8)
    for fmt in (oracle.BLOSC1, oracle.BLOSC2):
        ck = oracle.blosc_compress(data, 2, 8192, fmt)
        src = to_dev(ck)
        off = torch.tensor([0, ck.size], dtype=torch.int64, device="cuda")
        back, bad = ctx.decompress(src, off, 1, data.size, typesize=2, blocksize=8192)
        assert bad == 0 and np.array_equal(back.cpu().numpy(), data)
    if extlibs.have_blosc():
        ck = extlibs.blosc1_compress(data, 2, 8192)
        real_bs = int(ck[8:12].view("<u4")[0])     # c-blosc picks its own blocksize; the header says which
        src = to_dev(ck)
        off = torch.tensor([0, ck.size], dtype=torch.int64, device="cuda")
        back, bad = ctx.decompress(src, off, 1, data.size, typesize=2, blocksize=real_bs)
        assert bad == 0 and np.array_equal(back.cpu().numpy(), data)


@pytest.mark.parametrize("typesize,blocksize,tail", [(2, 8192, 0), (2, 8192, 3000), (1, 8192, 0), (4, 16384, 1234)])
def test_gpu_decoder_in_place_margin(ctx, typesize, blocksize, tail):
    """The decoder works in place (compressed bytes at the end of the plane's LDS buffer).  Worst case for that
    scheme: streams that barely compress — a short run, then literals to the end with every length-extension byte —
    so the read head leads the write head by the smallest margin LZ4 allows.  Streams are the oracle's (a foreign
    encoder), in both orders (run first / run last), plus a short last block decoded as one stream."""
    rng = np.random.default_rng(77)
    nblk = 6
    for run_first in (True, False):
        for run in (48, 64, 100, 255, 700):
            planes = rng.integers(0, 256, size=(nblk, typesize, blocksize // typesize), dtype=np.uint8)
            if run_first:
                planes[:, :, :run] = 7
            else:
                planes[:, :, -run:] = 7
            body = np.ascontiguousarray(planes.transpose(0, 2, 1)).reshape(-1)      # un-shuffled element order
            data = np.concatenate([body, rng.integers(0, 4, size=tail, dtype=np.uint8)])
            for fmt in (oracle.BLOSC1, oracle.BLOSC2):
                ck = oracle.blosc_compress(data, typesize, blocksize, fmt)
                assert ck.size < data.size + 64
                off = torch.tensor([0, ck.size], dtype=torch.int64, device="cuda")
                back, bad = ctx.decompress(to_dev(ck), off, 1, data.size, typesize=typesize, blocksize=blocksize)
                assert bad == 0 and np.array_equal(back.cpu().numpy(), data), (run_first, run, fmt)


def test_gpu_decoder_flags_corruption(ctx):
    data = genotype_like(4, 4096, 9).reshape(-1).view(np.uint8)
    ck = oracle.blosc_compress(data, 2, 8192, oracle.BLOSC2).copy()
    ck[40] ^= 0xFF   # first bstart
    off = torch.tensor([0, ck.size], dtype=torch.int64, device="cuda")
    _, bad = ctx.decompress(to_dev(ck), off, 1, data.size, typesize=2, blocksize=8192)
    assert bad >= 1


def test_fixture_chunk_matches_oracle_decode(ctx, golden_dir):
    import os
    G = np.load(os.path.join(golden_dir, "fixture_G.npy"))
    roundtrip(ctx, G.reshape(-1).view(np.uint8).copy(), 6000, 2, 2000, dev.BLOSC2)


def test_clevel_fast_mode_roundtrip(ctx):
    """clevel 1-2 = run candidate only (LZ4 'acceleration'); still a valid stream for every decoder"""
    from haplohyped_varawareml_amd._lib import HhgtError
    G = genotype_like(64 * 2, 8192, 13)
    data = np.ascontiguousarray(G.reshape(2, 64, 8192, 2)).reshape(-1).view(np.uint8)
    try:
        ctx.set_clevel(1)
        fast = roundtrip(ctx, data, 64 * 8192 * 2, 2, 8192, dev.BLOSC1)
        for name, arr in cases():
            if arr.size >= 64:
                roundtrip(ctx, arr, arr.size, 1, min(arr.size, 65536), dev.BLOSC2)
    finally:
        ctx.set_clevel(5)
    full = roundtrip(ctx, data, 64 * 8192 * 2, 2, 8192, dev.BLOSC1)
    assert sum(c.size for c in full) < sum(c.size for c in fast) < data.size / 2
    with pytest.raises(HhgtError):
        ctx.set_clevel(0)
