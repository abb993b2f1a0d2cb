"""CPU: pins oracle/codec_oracle.c (shuffle, LZ4 block format, Blosc chunk layout) against the
independent decoders of the base image (liblz4 1.9.3, c-blosc 1.21) and against committed golden
chunk bytes.  The reference's own codec tests assert round-trip equality only
(/root/reference/tests/test_compression.py:38-70,86-108); these do the same and more."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests import extlibs


def sparse_bits(n, p, seed):
    rng = np.random.default_rng(seed)
    return (rng.random(n) < p).astype(np.uint8)


def cases():
    rng = np.random.default_rng(7)
    yield "empty", np.zeros(0, np.uint8)
    yield "one", np.array([5], np.uint8)
    yield "short12", np.arange(12, dtype=np.uint8)
    yield "short13", np.zeros(13, np.uint8)
    yield "zeros", np.zeros(70000, np.uint8)
    yield "ones", np.full(4099, 1, np.uint8)
    yield "random", rng.integers(0, 256, 50000, dtype=np.uint8)
    yield "sparse6", sparse_bits(65536, 0.06, 1)
    yield "sparse40", sparse_bits(30001, 0.4, 2)
    yield "text", np.frombuffer((b"0|1\t1|0\t0|0\t" * 3000), dtype=np.uint8)
    yield "longrun", np.concatenate([rng.integers(0, 256, 300, dtype=np.uint8), np.zeros(40000, np.uint8),
                                     rng.integers(0, 4, 500, dtype=np.uint8)])


@pytest.mark.parametrize("name,data", list(cases()))
def test_lz4_roundtrip_and_liblz4(name, data):
    if data.size == 0:
        return
    comp = oracle.lz4_compress(data)
    assert np.array_equal(oracle.lz4_decompress(comp, data.size), data)
    if extlibs.have_lz4():
        # our stream through the real decoder (checks the end-of-block rules too) ...
        assert np.array_equal(extlibs.lz4_decompress(comp, data.size), data)
        # ... and the real encoder's stream through our decoder
        assert np.array_equal(oracle.lz4_decompress(extlibs.lz4_compress(data), data.size), data)


def test_lz4_decoder_rejects_garbage():
    with pytest.raises(RuntimeError):
        oracle.lz4_decompress(np.array([0xF0], np.uint8), 100)          # literal run past the input
    with pytest.raises(RuntimeError):
        oracle.lz4_decompress(np.array([0x10, 65, 9, 0], np.uint8), 100)  # offset beyond output start


@pytest.mark.parametrize("typesize", [1, 2, 3, 4, 8, 35])
def test_shuffle_matches_definition(typesize):
    rng = np.random.default_rng(typesize)
    for n in (0, 1, typesize, 10 * typesize + 3, 4096):
        src = rng.integers(0, 256, n, dtype=np.uint8)
        sh = oracle.shuffle(src, typesize)
        ne = n // typesize
        exp = np.concatenate([src[:ne * typesize].reshape(ne, typesize).T.reshape(-1), src[ne * typesize:]]) if typesize > 1 else src
        assert np.array_equal(sh, exp)
        assert np.array_equal(oracle.unshuffle(sh, typesize), src)


def genotype_like(n_samples, n_var, seed):
    rng = np.random.default_rng(seed)
    af = np.exp(rng.uniform(np.log(1 / 5000), np.log(0.5), n_var))
    g = (rng.random((n_samples, n_var, 2)) < af[None, :, None]).astype(np.int8)
    g[rng.random((n_samples, n_var)) < 0.001] = -9
    return g


BLOSC_CASES = [
    # (typesize, blocksize, nbytes)
    (2, 32768, 32768 * 4),        # the genotype chunk shape: split into 2 planes
    (2, 32768, 32768 * 3 + 1000), # leftover block (not split)
    (2, 4096, 4096 * 5),
    (1, 32768, 100000),
    (4, 8192, 8192 * 3 + 12),
    (35, 32760, 35 * 5000),       # the reference's 35-byte compound record (vcf_to_h5.py:119-127): no split
    (35, 32760, 35 * 100),
    (16, 16384, 16384 * 2),
    (17, 1700, 1700 * 3 + 5),
    (2, 64, 64),                  # blocksize/typesize < 128: no split
]


@pytest.mark.parametrize("typesize,blocksize,nbytes", BLOSC_CASES)
@pytest.mark.parametrize("fmt", [oracle.BLOSC1, oracle.BLOSC2])
def test_blosc_roundtrip(typesize, blocksize, nbytes, fmt):
    data = genotype_like(1, nbytes // 2 + 1, nbytes).reshape(-1).view(np.uint8)[:nbytes].copy()
    chunk = oracle.blosc_compress(data, typesize, blocksize, fmt)
    assert np.array_equal(oracle.blosc_decompress(chunk), data)
    hl = 32 if fmt == oracle.BLOSC2 else 16
    assert chunk[0] == (5 if fmt == oracle.BLOSC2 else 2) and chunk[3] == typesize
    assert int(chunk[4:8].view("<u4")[0]) == nbytes and int(chunk[12:16].view("<u4")[0]) == chunk.size
    if fmt == oracle.BLOSC1 and extlibs.have_blosc():
        # real c-blosc 1.21 decodes the oracle's chunk
        assert np.array_equal(extlibs.blosc1_decompress(chunk, nbytes), data)
    assert chunk.size <= nbytes + hl


@pytest.mark.parametrize("typesize,blocksize,nbytes", BLOSC_CASES)
def test_blosc_oracle_decodes_real_cblosc(typesize, blocksize, nbytes):
    if not extlibs.have_blosc():
        pytest.skip("c-blosc not in this image")
    data = genotype_like(1, nbytes // 2 + 1, nbytes + 1).reshape(-1).view(np.uint8)[:nbytes].copy()
    real = extlibs.blosc1_compress(data, typesize, blocksize)
    assert np.array_equal(oracle.blosc_decompress(real), data)


def test_blosc_incompressible_is_memcpyed():
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 70000, dtype=np.uint8)
    for fmt, hl in ((oracle.BLOSC1, 16), (oracle.BLOSC2, 32)):
        chunk = oracle.blosc_compress(data, 2, 32768, fmt)
        assert chunk.size == data.size + hl and chunk[2] & 0x2
        assert np.array_equal(oracle.blosc_decompress(chunk), data)
        if fmt == oracle.BLOSC1 and extlibs.have_blosc():
            assert np.array_equal(extlibs.blosc1_decompress(chunk, data.size), data)


def test_blosc2_header_fields():
    data = genotype_like(4, 16384, 5).reshape(-1).view(np.uint8)
    chunk = oracle.blosc_compress(data, 2, 32768, oracle.BLOSC2)
    # version 5, LZ4 format version 1, flags: extended (0x1|0x4) + LZ4 format (1<<5), split allowed
    assert chunk[0] == 5 and chunk[1] == 1 and chunk[2] == (0x1 | 0x4 | 0x20) and chunk[3] == 2
    assert bytes(chunk[16:22]) == b"\0\0\0\0\0\x01"          # filter pipeline: shuffle in the last slot
    assert bytes(chunk[22:32]) == b"\0" * 10
    nblocks = data.size // 32768
    bstarts = chunk[32:32 + 4 * nblocks].view("<u4")
    assert bstarts[0] == 32 + 4 * nblocks and np.all(np.diff(bstarts.astype(np.int64)) > 0)


def test_golden_chunk_bytes(golden_dir):
    """committed known-answer: sha256 of the oracle's chunk for the fixture genotypes (regenerate
    with tests/golden/make_golden_chunk.py); guards the oracle itself against drift."""
    import hashlib, json
    gold = json.load(open(os.path.join(golden_dir, "fixture_chunk_golden.json")))
    G = np.load(os.path.join(golden_dir, "fixture_G.npy"))
    for key, fmt in (("blosc1", oracle.BLOSC1), ("blosc2", oracle.BLOSC2)):
        chunk = oracle.blosc_compress(G.reshape(-1).view(np.uint8), 2, 2000, fmt)
        assert hashlib.sha256(chunk.tobytes()).hexdigest() == gold[key]["sha256"]
        assert chunk.size == gold[key]["cbytes"]
