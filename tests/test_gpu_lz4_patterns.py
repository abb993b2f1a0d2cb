"""-m gpu: structured streams aimed at the window-parallel LZ4 encoder's seams (csrc/lz4.hip):
the 64-position window, the 20-byte per-lane match cap, the offset-1 run candidate, the run-dominance rule,
the sequence queue / flush, and LZ4's end-of-block rules (last 5 bytes literal, no match start in the last 12).
Every stream must decode to the identical bytes with the oracle's decoder, liblz4 (when loadable) and the GPU
decoder; on sparse planes the compressed size must stay within 10 % of the CPU LZ4 oracle's."""
import numpy as np
import pytest
import torch

from oracle import oracle
from haplohyped_varawareml_amd import device as dev
from tests import extlibs
from tests.gpu_util import split_chunks, to_dev

pytestmark = pytest.mark.gpu


def compress_streams(ctx, data, n):
    """data: uint8 [k * n] -> list of (raw stream bytes or None if stored/memcpyed, original) per n-byte chunk"""
    k = data.size // n
    bs = max(n, 16)      # the ABI wants blocksize >= 16; a block never exceeds its chunk
    dst, off, total = ctx.compress(to_dev(data), n, typesize=1, blocksize=bs, fmt=dev.BLOSC1)
    out = []
    for i, ck in enumerate(split_chunks(dst, off, total)):
        want = data[i * n:(i + 1) * n]
        assert np.array_equal(oracle.blosc_decompress(ck), want), f"chunk {i} (n={n}): oracle decode differs"
        if ck[2] & 0x2:
            out.append((None, want))
            continue
        b0 = int(ck[16:20].view("<u4")[0])
        cs = int(ck[b0:b0 + 4].view("<u4")[0])
        stream = ck[b0 + 4:b0 + 4 + cs]
        if cs != n:
            assert np.array_equal(oracle.lz4_decompress(stream, n), want)
            if extlibs.have_lz4():
                assert np.array_equal(extlibs.lz4_decompress(stream, n), want), f"chunk {i} (n={n}): liblz4 decode differs"
        out.append((stream if cs != n else None, want))
    back, bad = ctx.decompress(dst, off, k, n, typesize=1, blocksize=bs)
    assert bad == 0 and np.array_equal(back.cpu().numpy(), data[:k * n])
    return out


def sparse(rng, n, p, missing=0.0):
    a = (rng.random(n) < p).astype(np.uint8)
    if missing:
        a[rng.random(n) < missing] = 0xF7
    return a


@pytest.mark.parametrize("n", list(range(1, 40)) + [63, 64, 65, 75, 76, 77, 127, 128, 129, 191, 192, 193, 255, 256, 257])
def test_short_streams(ctx, n):
    """streams shorter than / around one window and around MFLIMIT (12) / LASTLITERALS (5)"""
    rng = np.random.default_rng(n)
    rows = [np.zeros(n, np.uint8), np.ones(n, np.uint8), sparse(rng, n, 0.1), sparse(rng, n, 0.5),
            np.tile(np.array([1, 0, 0], np.uint8), n // 3 + 1)[:n], rng.integers(0, 256, n, dtype=np.uint8)]
    compress_streams(ctx, np.concatenate(rows), n)


@pytest.mark.parametrize("gap", [1, 2, 3, 4, 5, 6, 15, 17, 18, 19, 20, 21, 22, 23, 34, 35, 36, 37, 38, 62, 63, 64, 65, 66, 127, 128, 129, 300])
def test_ones_at_fixed_gap(ctx, gap):
    """'1' every `gap` bytes: hash matches of length gap (periodic) against runs of gap - 1 zeros; the caps and the
    run-dominance rule sit at gaps 18-23 and 34-38, the window at 62-66"""
    n = 4096
    for phase in (0, 1, gap // 2):
        a = np.zeros(n, np.uint8)
        a[phase::gap] = 1
        compress_streams(ctx, a, n)


def test_random_gaps_ratio_close_to_cpu_lz4(ctx):
    """sparse genotype-like planes at several densities: valid streams and a size within 10 % of the CPU oracle's LZ4"""
    n = 4096
    for seed, p in enumerate((0.002, 0.01, 0.03, 0.1, 0.3)):
        rng = np.random.default_rng(100 + seed)
        data = np.concatenate([sparse(rng, n, p, missing=0.002) for _ in range(64)])
        res = compress_streams(ctx, data, n)
        gpu = sum(s.size if s is not None else n for s, _ in res)
        cpu = sum(min(oracle.lz4_compress(w).size, n) for _, w in res)
        assert gpu <= 1.10 * cpu + 64, (p, gpu, cpu)


@pytest.mark.parametrize("period", [1, 2, 3, 5, 7, 16, 19, 20, 21, 33, 63, 64, 65, 100, 257])
def test_periodic_motifs(ctx, period):
    """a random motif repeated: one long hash match (offset = period) that the cooperative extension must carry
    across many windows; then the same with a literal break every 1000 bytes"""
    rng = np.random.default_rng(period)
    motif = rng.integers(0, 4, period, dtype=np.uint8)
    n = 8192
    a = np.tile(motif, n // period + 1)[:n].copy()
    compress_streams(ctx, a, n)
    a[::1000] ^= 0x55
    compress_streams(ctx, a, n)


@pytest.mark.parametrize("tail", range(0, 20))
def test_match_reaching_the_end(ctx, tail):
    """a run / a repeat that would continue into the last `tail` bytes: the last 5 bytes must stay literals and no
    match may start within the last 12"""
    n = 1000
    a = np.zeros(n, np.uint8)
    if tail:
        a[n - tail:] = np.arange(1, tail + 1, dtype=np.uint8)
    b = np.tile(np.array([3, 1, 4, 1, 5, 9, 2, 6], np.uint8), n // 8 + 1)[:n].copy()
    if tail:
        b[n - tail] ^= 0xFF
    compress_streams(ctx, np.concatenate([a, b]), n)


def test_run_boundaries_around_windows(ctx):
    """runs that end exactly at, one before and one after each of the first window boundaries, of several byte values"""
    n = 2048
    rows = []
    for edge in (64, 128, 192, 256):
        for d in (-2, -1, 0, 1, 2):
            for v in (0, 1, 0xF7):
                a = np.full(n, v, np.uint8)
                a[edge + d] = v ^ 1
                a[edge + d + 40] = v ^ 1
                rows.append(a)
    compress_streams(ctx, np.concatenate(rows), n)


def test_many_short_sequences_fill_the_queue(ctx):
    """alternating 4-byte matches and single literals: up to 16 sequences per window, so the per-wave queue flushes
    every third window; then literal stretches longer than 32 bytes (the wave-cooperative literal copy)"""
    rng = np.random.default_rng(7)
    n = 16384
    a = np.zeros(n, np.uint8)
    a[::5] = rng.integers(1, 256, a[::5].size, dtype=np.uint8)
    b = np.zeros(n, np.uint8)
    for s0 in range(0, n - 200, 200):
        b[s0:s0 + 40 + (s0 // 200) % 60] = rng.integers(0, 256, 40 + (s0 // 200) % 60, dtype=np.uint8)
    compress_streams(ctx, np.concatenate([a, b]), n)


@pytest.mark.parametrize("clevel", [1, 9])
def test_other_effort_levels_same_patterns(ctx, clevel):
    """clevel 1 (run candidate only) and clevel 9 (plus the long-run source candidate) over the sparse and the
    boundary patterns; the higher effort must not compress worse than the default on sparse planes"""
    n = 4096
    rng = np.random.default_rng(5)
    data = np.concatenate([sparse(rng, n, p, missing=0.003) for p in (0.001, 0.01, 0.03, 0.1, 0.5)] * 4 + [np.zeros(n, np.uint8)] +
                          [np.where(np.arange(n) % g == 0, 1, 0).astype(np.uint8) for g in (7, 19, 33, 70, 200)])
    base = sum(s.size if s is not None else n for s, _ in compress_streams(ctx, data, n))
    try:
        ctx.set_clevel(clevel)
        got = sum(s.size if s is not None else n for s, _ in compress_streams(ctx, data, n))
    finally:
        ctx.set_clevel(5)
    if clevel == 9:
        assert got <= base, (got, base)
