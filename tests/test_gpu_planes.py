"""-m gpu: the bit-plane form of the genotype matrix (include/hhgt.h "Bit-plane form": hhgt_encode_text_planes_async,
hhgt_pad_tail_planes*, hhgt_compress_planes, hhgt_planes_expand) — the intermediate between encode and compress when
the compressor is the only consumer.  The contract does not change with it: the planes expand to the int8 values the
oracle's restatement of cpp/parse_vcf.cpp:46-61 / cpp/vcfpp.h:546-588 gives, and the chunks compressed FROM the planes
decode (oracle decoder) to those bytes — on fixed-width text (C2 shape), on the config-4 mixture (missing calls, '/',
GT:DP lines, dropped records) and on the reference's own fixture (C1: every line on the variable-width path)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.gpu_util import split_chunks, to_dev
from haplohyped_varawareml_amd import device as dev, synth

pytestmark = pytest.mark.gpu


def _blocks(text, n):
    """n line-aligned pieces of a VCF text"""
    a = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
    nl = np.flatnonzero(a == 10) + 1
    cuts = [0] + [int(nl[(len(nl) * (k + 1)) // n - 1]) for k in range(n)]
    return [bytes(a[cuts[k]:cuts[k + 1]]) for k in range(n) if cuts[k + 1] > cuts[k]]


def _new_result(ctx, lay, with_g=True, poison=False):
    d = ctx.device
    cap = lay.v_capacity
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=d)
    nb = dev.planes_bytes(lay)
    assert nb == dev.layout_bytes(lay) // 4 and nb > 0
    P = torch.full((nb,), 0xA5, dtype=torch.uint8, device=d) if poison else z(nb, torch.uint8)   # poison: nothing may rely on zeroed planes
    G = z(max(dev.layout_bytes(lay), 16), torch.uint8) if with_g else None
    return dev.EncodeResult(G, lay, z(cap, torch.int32), z(cap, torch.int32), z(cap, torch.uint8), z(cap, torch.uint8), 0, {}, [], P)


def _dense_from_bytes(raw, lay, n):
    """int8 [S, n, 2] out of the chunk-tiled bytes"""
    r = dev.EncodeResult(raw, lay, None, None, None, None, n, {})
    return r.dense().cpu().numpy()


def _encode_planes(ctx, text, S, lay, region, nblk=1, with_g=True, max_lines=None):
    res = _new_result(ctx, lay, with_g=with_g, poison=True)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    pend, keep = [], []
    for blk in _blocks(text, nblk):
        keep.append(to_dev(blk))
        pend.append(ctx.encode_text_planes_async(keep[-1], S, res, cursor, region=region,
                                                 max_lines=max_lines(keep[-1]) if max_lines else None))
    recs = [p.wait() for p in pend]
    n = int(cursor.item())
    Vc = lay.vc or lay.v_capacity
    # everything behind the cursor up to the end of the touched columns, and the sample padding rows
    ctx.pad_tail_planes(res, n, 0, max(-(-n // Vc), 1))
    return res, n, recs


@pytest.mark.parametrize("S,V,sc,vc,nblk", [(300, 9000, 64, 4096, 1), (1000, 5000, 64, 8192, 3), (2504, 1500, 64, 4096, 4),
                                            (257, 4100, 0, 0, 2), (5, 700, 64, 4096, 5), (64, 12289, 64, 4096, 7)])
def test_fixed_width_planes_vs_oracle(ctx, S, V, sc, vc, nblk):
    tab = synth.variant_table(5, V, S)
    text, _ = synth.render_fixed_numpy("chr5", tab, S, seed=5)
    o = oracle.vcf_encode(text, S, region="chr5")
    cap = -(-V // 4096) * 4096
    lay = dev.make_layout(S, cap, sc=sc, vc=vc) if vc else dev.Layout(S, 0, 0, 0, cap)
    res, n, recs = _encode_planes(ctx, text, S, lay, "chr5", nblk, with_g=False)
    assert n == V == o["n_kept"] and all(r.reserved == 0 for r in recs)
    back = ctx.planes_expand(res)
    assert np.array_equal(_dense_from_bytes(back, lay, V), o["G"])
    # nothing but zeros behind the last variant and in the sample padding rows
    full = dev.EncodeResult(back, lay, None, None, None, None, lay.v_capacity, {}).dense()
    Vc = lay.vc or lay.v_capacity
    assert not full[:, V:-(-V // Vc) * Vc].any()
    # the same text through the int8 path gives the same bytes wherever a chunk column was touched
    res8 = _new_result(ctx, lay)
    c8 = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    t = to_dev(text)
    ctx.encode_text_async(t, S, res8, c8, region="chr5").wait()
    ctx.pad_tail_cursor(res8, c8)
    res8.n_kept = V
    assert np.array_equal(res8.dense().cpu().numpy(), o["G"])
    assert np.array_equal(res.start[:V].cpu().numpy().view(np.uint32), o["start"])
    assert np.array_equal(res.ref[:V].cpu().numpy(), o["ref"]) and np.array_equal(res.alt[:V].cpu().numpy(), o["alt"])


def test_mixed_c4_planes_vs_oracle(ctx):
    """config-4 mixture: ./. and .|1 calls, '/' separators, GT:DP lines (variable-width path), dropped records"""
    S, V = 700, 9000
    t = synth.mixed_table(4, V, S)
    text, n, _ = ctx.synth_mixed("chr4", t, S, seed=4)
    host = text.cpu().numpy()
    o = oracle.vcf_encode(host, S, region="chr4", cap=V)
    lay = dev.make_layout(S, -(-o["n_kept"] // 4096) * 4096, sc=64, vc=4096)
    res, nk, recs = _encode_planes(ctx, host, S, lay, "chr4", 3, with_g=False, max_lines=lambda t_: t_.numel() // 16 + 8)
    assert nk == o["n_kept"] and sum(r.stats.n_drop_filter for r in recs) == o["stats"]["n_drop_filter"]
    assert all(r.reserved == 0 for r in recs)                 # under the reference's filter: 0, 1 and missing only
    # (round 4: the GT:DP records of this text have columns of one width — "a|b:dd" — and are decoded by the tile kernel at that
    # stride; the variable-width kernel only sees a record whose last columns sit within 8 bytes of the end of a text block)
    assert sum(r.stats.n_general_lines for r in recs) <= 3 < int(t["with_dp"][np.nonzero(t["kept"])[0]].sum())
    back = ctx.planes_expand(res)
    G = _dense_from_bytes(back, lay, nk)
    assert (o["G"] == -9).any()
    assert np.array_equal(G, o["G"])


def test_fixture_planes_vs_golden(ctx, fixture_text, golden_dir):
    """the reference's own fixture (GT:GQ:DP columns: every line takes the variable-width kernel and sets its bits by
    atomic or) against the committed golden matrix"""
    want = np.load(f"{golden_dir}/fixture_G.npy")
    lay = dev.make_layout(3, 4096, sc=64, vc=4096)
    res, n, recs = _encode_planes(ctx, fixture_text, 3, lay, "chr22", 2, max_lines=lambda t_: t_.numel() // 16 + 8)
    assert n == 1000
    G = _dense_from_bytes(ctx.planes_expand(res), lay, n)
    assert np.array_equal(G, want)


def test_keep_multiallelic_other_calls(ctx):
    """non-reference mode: allele indices >= 2 are 'other' calls — EXC without ONE, byte in G"""
    S, V = 130, 5000
    t = synth.mixed_table(41, V, S)
    text, n, _ = ctx.synth_mixed("chr4", t, S, seed=41)
    host = text.cpu().numpy()
    o = oracle.vcf_encode(host, S, region="chr4", cap=V, keep_multiallelic=True)
    assert o["G"].max() >= 2
    ctx.set_keep_multiallelic(True)
    try:
        lay = dev.make_layout(S, -(-o["n_kept"] // 4096) * 4096, sc=64, vc=4096)
        res, nk, recs = _encode_planes(ctx, host, S, lay, "chr4", 2, max_lines=lambda t_: t_.numel() // 16 + 8)
    finally:
        ctx.set_keep_multiallelic(False)
    assert nk == o["n_kept"]
    n_other = int(((o["G"] != 0) & (o["G"] != 1) & (o["G"] != -9)).sum())
    assert sum(r.reserved for r in recs) == n_other > 0      # exact: every line is decoded by the variable-width kernel at most once
    G = _dense_from_bytes(ctx.planes_expand(res), lay, nk)
    assert np.array_equal(G, o["G"])
    # compressed from the planes (+ G for the other calls): every chunk decodes to the oracle's bytes
    want = np.zeros((-(-S // 64) * 64, lay.v_capacity, 2), np.int8)
    want[:S, :nk] = o["G"]
    dst, off, total = ctx.compress_planes(res, fmt=dev.BLOSC1)
    chunks = split_chunks(dst, off, total)
    k = 0
    for vcol in range(lay.v_capacity // 4096):
        for scol in range(-(-S // 64)):
            back = oracle.blosc_decompress(chunks[k]).view(np.int8).reshape(64, 4096, 2)
            assert np.array_equal(back, want[scol * 64:(scol + 1) * 64, vcol * 4096:(vcol + 1) * 4096]), (vcol, scol)
            k += 1


@pytest.mark.parametrize("shape", ["fixed", "mixed"])
def test_compress_planes_decodes_to_oracle_matrix(ctx, shape):
    """encode -> planes -> hhgt_compress_planes: every chunk through the ORACLE decoder equals the oracle's int8 matrix
    (zero padded), and is byte-identical to the chunk hhgt_compress_chunks makes from the int8 matrix"""
    S, V = (1000, 10000) if shape == "fixed" else (400, 12000)
    if shape == "fixed":
        text, _ = synth.render_fixed_numpy("chr2", synth.variant_table(2, V, S), S, seed=2)
        ml = None
    else:
        text = ctx.synth_mixed("chr2", synth.mixed_table(2, V, S), S, seed=2)[0].cpu().numpy()
        ml = lambda t_: t_.numel() // 16 + 8
    o = oracle.vcf_encode(text, S, region="chr2", cap=V)
    nk = o["n_kept"]
    lay = dev.make_layout(S, nk, sc=64, vc=8192)
    res, n, _ = _encode_planes(ctx, text, S, lay, "chr2", 2, with_g=False, max_lines=ml)
    assert n == nk
    chunk_nbytes = 64 * 8192 * 2
    for fmt in (dev.BLOSC1, dev.BLOSC2):
        dst, off, total = ctx.compress_planes(res, fmt=fmt)
        chunks = split_chunks(dst, off, total)
        want = np.zeros((-(-S // 64) * 64, lay.v_capacity, 2), np.int8)
        want[:S, :nk] = o["G"]
        k = 0
        for vcol in range(lay.v_capacity // 8192):
            for scol in range(-(-S // 64)):
                back = oracle.blosc_decompress(chunks[k]).view(np.int8).reshape(64, 8192, 2)
                assert np.array_equal(back, want[scol * 64:(scol + 1) * 64, vcol * 8192:(vcol + 1) * 8192]), (vcol, scol)
                k += 1
        # the int8 path on the expanded matrix: same chunks, byte for byte (the bit-plane coder sees the same bit maps).
        # Only where every stream is the bit-plane coder's: the byte-wise kernel that codes planes with missing calls
        # emits valid streams whose bytes are not reproducible from run to run, from either input.
        raw = ctx.planes_expand(res)
        if shape == "fixed":
            dst8, off8, total8 = ctx.compress(raw, chunk_nbytes, typesize=2, blocksize=8192, fmt=fmt)
            assert total8 == total and torch.equal(off8, off) and torch.equal(dst8[:total8], dst[:total])
        # and the GPU decoder agrees
        back, bad = ctx.decompress(dst, off, len(chunks), chunk_nbytes, typesize=2, blocksize=8192)
        assert bad == 0 and torch.equal(back, raw)


def test_ring_of_plane_columns(ctx):
    """the planes as a ring of chunk columns (streaming): completed columns are read before later blocks overwrite them"""
    S, V, vc, ring = 100, 30000, 4096, 3
    text, _ = synth.render_fixed_numpy("chr9", synth.variant_table(9, V, S), S, seed=9)
    o = oracle.vcf_encode(text, S, region="chr9")
    lay = dev.make_ring_layout(S, ring, sc=64, vc=vc)
    res = _new_result(ctx, lay, with_g=False, poison=True)
    ctx.pad_tail_planes(res, lay.v_capacity, 0, ring)          # sample padding rows of every ring column, once
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    n_sc = 2
    col_bytes = n_sc * 64 * vc * 2
    ncol = -(-V // vc)
    got = np.zeros((n_sc * 64, ncol * vc, 2), np.int8)
    done = 0
    scratch = torch.zeros(dev.layout_bytes(lay), dtype=torch.uint8, device=ctx.device)

    def take(col):
        slot = col % ring
        ctx.planes_expand(res, col0=slot, n_cols=1, out=scratch)
        raw = scratch[slot * col_bytes:(slot + 1) * col_bytes]
        got[:, col * vc:(col + 1) * vc] = raw.view(torch.int8).view(n_sc, 64, vc, 2).reshape(n_sc * 64, vc, 2).cpu().numpy()

    for blk in _blocks(text, 23):                      # ~1300 variants per block: less than a column, odd phases
        t = to_dev(blk)
        rec = ctx.encode_text_planes_async(t, S, res, cursor, region="chr9").wait()
        for col in range(done, rec.cursor_after // vc):
            take(col)
        done = rec.cursor_after // vc
    assert int(cursor.item()) == V
    ctx.pad_tail_planes_cursor(res, cursor)
    take(done)
    assert np.array_equal(got[:S, :V], o["G"]) and not got[S:].any() and not got[:, V:].any()
    # one ring slot compressed on its own: its two chunks decode to the column the slot holds now
    slot = done % ring
    dst, off, total = ctx.compress_planes(res, col0=slot, n_cols=1, fmt=dev.BLOSC1)
    for scol, ck in enumerate(split_chunks(dst, off, total)):
        back = oracle.blosc_decompress(ck).view(np.int8).reshape(64, vc, 2)
        assert np.array_equal(back, got[scol * 64:(scol + 1) * 64, done * vc:(done + 1) * vc])


def test_incompressible_planes_are_stored(ctx):
    """every call an 'other' call (EXC without ONE) whose byte in G is random: nothing compresses, Blosc stores the chunk
    verbatim (memcpyed) and the framing generates the bytes from planes + G; the oracle decodes them to G's bytes.
    The planes are written by hand here, which also pins the documented tile-major layout (include/hhgt.h)."""
    rng = np.random.default_rng(7)
    S, vc, ncol = 16, 4096, 4
    lay = dev.make_layout(S, ncol * vc, sc=16, vc=vc)
    assert dev.planes_bytes(lay) == ncol * 16 * 4 * 16 * 32
    P = np.zeros((ncol, vc // 256, 4, 16, 32), np.uint8)      # [column][tile][kind-plane][sample row][32 B]
    P[:, :, 2:] = 0xFF
    G = rng.integers(0, 256, dev.layout_bytes(lay), dtype=np.uint8)
    res = dev.EncodeResult(to_dev(G), lay, None, None, None, None, 0, {}, [], to_dev(P.reshape(-1)))
    assert np.array_equal(ctx.planes_expand(res).cpu().numpy(), G)
    # a sparse (sample row, 4096-variant block) in between still expands from its bits alone: column 1, row 5
    P2 = P.copy()
    P2[1, :, :, 5, :] = 0
    P2[1, 0, 0, 5, 17] = 0x81            # haplotype 0: allele 1 at variants 136 and 143
    P2[1, 6, 1, 5, 8] = 0x10             # haplotype 1: ONE ...
    P2[1, 6, 3, 5, 8] = 0x10             # ... and EXC at variant 6 * 256 + 68: a missing call
    res2 = dev.EncodeResult(res.G, lay, None, None, None, None, 0, {}, [], to_dev(P2.reshape(-1)))
    raw2 = ctx.planes_expand(res2).cpu().numpy().view(np.int8).reshape(ncol, 16, vc, 2)      # [column][row][variant][h]
    want5 = np.zeros((vc, 2), np.int8)
    want5[136, 0] = want5[143, 0] = 1
    want5[6 * 256 + 68, 1] = -9
    assert np.array_equal(raw2[1, 5], want5)
    assert np.array_equal(raw2[1, 6].reshape(-1).view(np.uint8), G.reshape(ncol, 16, vc * 2)[1, 6])
    chunk_nbytes = 16 * vc * 2
    for fmt in (dev.BLOSC1, dev.BLOSC2):
        dst, off, total = ctx.compress_planes(res, fmt=fmt)
        chunks = split_chunks(dst, off, total)
        assert len(chunks) == ncol
        for i, ch in enumerate(chunks):
            assert ch[2] & 0x2, "chunk should be stored verbatim"
            assert np.array_equal(oracle.blosc_decompress(ch), G[i * chunk_nbytes:(i + 1) * chunk_nbytes])
    # the hand-made sparse block through the compressor as well (its chunk is compressible no more than the others:
    # still stored; what matters is that it decodes to the expansion)
    dst, off, total = ctx.compress_planes(res2, col0=1, n_cols=1, fmt=dev.BLOSC1)
    ck = split_chunks(dst, off, total)[0]
    assert np.array_equal(oracle.blosc_decompress(ck).view(np.int8).reshape(16, vc, 2), raw2[1])


def test_sample_columns_beyond_2_gib(ctx):
    """offsets into the text are 32-bit UNSIGNED: data lines that start behind 2 GiB of comment lines (a device kernel that
    sign-extends an offset reads 4 GiB off — found the hard way with the aligned loads' scalar tail load)"""
    S, V = 600, 700                                     # three bands of 256 samples: two of them take the aligned-load path
    body, _ = synth.render_fixed_numpy("chr7", synth.variant_table(7, V, S), S, seed=7)
    o = oracle.vcf_encode(body, S, region="chr7")
    pad_line = torch.full((1 << 20,), ord("x"), dtype=torch.uint8, device=ctx.device)
    pad_line[:2] = ord("#")
    pad_line[-1] = 10
    n_pad = 2100                                        # 2.2 GB of '#' lines of 1 MiB
    text = torch.empty(n_pad * (1 << 20) + len(body) + 16, dtype=torch.uint8, device=ctx.device)
    text[:n_pad << 20].view(n_pad, 1 << 20)[:] = pad_line
    text[n_pad << 20:(n_pad << 20) + len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8).to(ctx.device)
    text = text[:(n_pad << 20) + len(body)]
    lay = dev.make_layout(S, 4096, sc=64, vc=4096)
    res = _new_result(ctx, lay, with_g=False, poison=True)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    rec = ctx.encode_text_planes_async(text, S, res, cursor, region="chr7", max_lines=V + n_pad + 64).wait()
    assert rec.stats.n_kept == V == o["n_kept"] and rec.stats.n_general_lines == 0
    ctx.pad_tail_planes(res, V, 0, 1)
    assert np.array_equal(_dense_from_bytes(ctx.planes_expand(res), lay, V), o["G"])
    del text


def test_bad_layout_is_refused(ctx):
    lay = dev.make_layout(10, 1024, sc=64, vc=1024)
    assert dev.planes_bytes(lay) == 0
    res = dev.EncodeResult(None, lay, None, None, None, None, 0, {}, [], torch.zeros(4096, dtype=torch.uint8, device=ctx.device))
    with pytest.raises(dev.HhgtError):
        ctx.encode_text_planes_async(to_dev(b"#x\n"), 10, res, torch.zeros(1, dtype=torch.int64, device=ctx.device))


@pytest.mark.parametrize("S", [5, 257, 700])
def test_columns_of_one_width_go_through_the_tile_kernel(ctx, S):
    """Round 4: a kept record whose FORMAT starts with GT and whose sample columns are all of one width of 5 .. 8 bytes ("a|b:dd",
    config 4's GT:DP records) is decoded by k_encode_planes at that stride — every column checked where it stands — instead of
    by the variable-width kernel.  Widths 5 .. 9, missing and unphased calls, and the records that only LOOK regular (the total
    length fits, two columns trade a byte; a tab inside a sub-field's place; a third allele; CRLF): all equal to the oracle, and
    only the irregular ones reach the variable-width kernel."""
    rng = np.random.default_rng(S)
    hdr = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"s%d" % i for i in range(S)) + b"\n"
    gts = [b"0|0", b"0|1", b"1|0", b"1|1", b"./.", b".|1", b"0/1", b"0|."]
    lines, irregular = [], 0
    for v in range(400):
        kind = v % 10
        g = [gts[i] for i in rng.integers(0, len(gts), S)]
        fmt = b"GT:DP"
        if kind == 0:
            cols = [x + b":%d" % rng.integers(0, 10) for x in g]                     # width 6
        elif kind in (1, 2, 3):
            cols = [x + b":%02d" % rng.integers(10, 100) for x in g]                  # width 7 (config 4's shape)
        elif kind == 4:
            cols = [x + b":%03d" % rng.integers(100, 1000) for x in g]                # width 8
        elif kind == 5:
            cols = [x + b":" for x in g]                                              # width 5: an empty second sub-field
        elif kind == 6:
            fmt = b"GT:GQ:DP"
            cols = [x + b":9:%02d" % rng.integers(10, 100) for x in g]                # width 9: not this path
            irregular += 1
        elif kind == 7 and S >= 2:
            cols = [x + b":%02d" % rng.integers(10, 100) for x in g]                  # the length fits, two columns trade a byte
            i, k = sorted(rng.choice(S, 2, replace=False))
            cols[i] = g[i] + b":100"
            cols[k] = g[k] + b":7"
            irregular += 1
        elif kind == 8:
            cols = [x + b":%02d" % rng.integers(10, 100) for x in g]
            cols[int(rng.integers(0, S))] = b"2|1:33"                                  # a third allele: same width, not this alphabet
            irregular += 1
        else:
            cols = [x + b":%02d" % rng.integers(10, 100) for x in g]
            cols[int(rng.integers(0, S))] = b"10|1:" if S > 1 else cols[0]            # a two-digit allele in a column of the same width
            irregular += S > 1
        lines.append(b"chr3\t%d\t.\tA\tC\t.\tPASS\t.\t" % (100 + 7 * v) + fmt + b"\t" + b"\t".join(cols))
    for text in (hdr + b"\n".join(lines) + b"\n", hdr + b"\r\n".join(lines[:40]) + b"\r\n"):
        crlf = b"\r\n" in text[len(hdr):]
        o = oracle.vcf_encode(text, S, region="chr3")
        lay = dev.make_layout(S, -(-o["n_kept"] // 4096) * 4096, sc=64, vc=4096)
        res, nk, recs = _encode_planes(ctx, text, S, lay, "chr3", 3, max_lines=lambda t_: t_.numel() // 16 + 8)
        assert nk == o["n_kept"] == (40 if crlf else 400)
        G = _dense_from_bytes(ctx.planes_expand(res), lay, nk)
        assert np.array_equal(G, o["G"])
        n_gen = sum(r.stats.n_general_lines for r in recs)
        if not crlf:
            assert irregular <= n_gen <= irregular + 6, (n_gen, irregular)   # (+ a record whose last columns sit at the end of a text block)
        assert sum(r.stats.n_haploid_padded for r in recs) == o["stats"]["n_haploid_padded"]
