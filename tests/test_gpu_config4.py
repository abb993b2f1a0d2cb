"""-m gpu: BASELINE config 4 — multiallelic + missing-call VCF (./., .|1, 0/1, GT:DP columns, indel / '*' /
'<DEL>' / lower-case ALTs).  Under the reference's semantics every non-biallelic-SNP record is dropped
(cpp/vcfpp.h:990-1000), missing alleles become -9 (cpp/vcfpp.h:567-573).  Small and medium sizes are compared
with the oracle; the full 500 000 x 5000 size is checked through size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import oracle
from haplohyped_varawareml_amd import device as dev, synth
from tests.gpu_util import assert_same_as_oracle, gpu_encode

pytestmark = pytest.mark.gpu


def test_device_renderer_matches_numpy_mirror(ctx):
    for S, V in ((9, 400), (130, 257)):
        t = synth.mixed_table(4, V, S)
        ref_text, off = synth.render_mixed_numpy("chr4", t, S, 4)
        text, n, doff = ctx.synth_mixed("chr4", t, S, seed=4)
        assert n == len(ref_text) and np.array_equal(doff, off)
        assert bytes(text.cpu().numpy()) == ref_text


def test_medium_vs_oracle(ctx):
    S, V = 5000, 6000
    t = synth.mixed_table(4, V, S)
    text, n, _ = ctx.synth_mixed("chr4", t, S, seed=4)
    g = gpu_encode(ctx, text, S, region="chr4")
    o = oracle.vcf_encode(text.cpu().numpy(), S, region="chr4", cap=V)
    assert_same_as_oracle(g, o)
    kept = np.nonzero(t["kept"])[0]
    assert g["n_kept"] == len(kept) and g["stats"]["n_drop_filter"] == V - len(kept)
    assert g["stats"]["n_general_lines"] == int(t["with_dp"][kept].sum())     # only GT:DP lines leave the tile kernel
    assert (g["G"] == -9).any() and g["G"].max() == 1


def test_full_size_500k_x_5000_properties(ctx):
    S, V = 5000, 500_000
    seed = 4
    t = synth.mixed_table(seed, V, S)
    kept = np.nonzero(t["kept"])[0]
    lay = dev.make_layout(S, len(kept))
    cap = lay.v_capacity
    d = ctx.device
    res = dev.EncodeResult(torch.zeros(dev.layout_bytes(lay), dtype=torch.uint8, device=d), lay,
                           torch.zeros(cap, dtype=torch.int32, device=d), torch.zeros(cap, dtype=torch.int32, device=d),
                           torch.zeros(cap, dtype=torch.uint8, device=d), torch.zeros(cap, dtype=torch.uint8, device=d), 0, {})
    # the text is ~11.5 GB: render and encode it in pieces below 4 GiB (whole lines), appending at v_base
    piece = 100_000
    v_base, tot = 0, dict(n_records=0, n_drop_filter=0, n_general_lines=0, n_haploid_padded=0)
    for a in range(0, V, piece):
        sub = {k: (v[a:a + piece] if isinstance(v, np.ndarray) and len(v) == V else v) for k, v in t.items()}
        text, n, _ = ctx.synth_mixed("chr4", sub, S, seed=seed, v_first=a, with_header=(a == 0))
        assert n < (1 << 32)
        res = ctx.encode_text(text, S, region="chr4", v_base=v_base, out=res)
        for k in tot:
            tot[k] += res.stats[k]
        v_base = res.n_kept
        del text
    assert res.n_kept == len(kept) and tot["n_records"] == V and tot["n_drop_filter"] == V - len(kept)
    assert tot["n_general_lines"] == int(t["with_dp"][kept].sum()) and tot["n_haploid_padded"] == 0
    assert np.array_equal(res.start[:len(kept)].cpu().numpy().view(np.uint32) + 1, t["pos"][kept])
    # sampled columns against the generator's own call rule (numpy mirror of the hash)
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(len(kept), 1500, replace=False))
    exp = synth.mixed_expected_G(seed, t, S, kept[pick])
    Sc, Vc = lay.sc, lay.vc
    n_sc, n_vc = -(-S // Sc), cap // Vc
    Gv = res.G.view(torch.int8).view(n_vc, n_sc, Sc, Vc, 2)
    pk = torch.from_numpy(pick).to(d)
    got = Gv[pk // Vc, :, :, pk % Vc, :]                       # [npick, n_sc, Sc, 2]
    got = got.reshape(len(pick), n_sc * Sc, 2)[:, :S].permute(1, 0, 2).cpu().numpy()
    assert np.array_equal(got, exp)
    # value set and global counts
    g8 = res.G.view(torch.int8)
    assert int(g8.min()) == -9 and int(g8.max()) == 1
    assert not bool(((g8 != 0) & (g8 != 1) & (g8 != -9)).any())
    del g8
    # encode -> pad -> compress -> GPU decode round trip over the whole matrix
    ctx.pad_tail(res)
    chunk_nbytes = Sc * Vc * 2
    dst, off, total = ctx.compress(res.G, chunk_nbytes)
    back, bad = ctx.decompress(dst, off, res.G.numel() // chunk_nbytes, chunk_nbytes)
    assert bad == 0 and torch.equal(back, res.G)
    assert res.G.numel() / total > 2.0
    # a few chunks through the CPU oracle's decoder as well
    offs = off.cpu().numpy()
    for i in (0, len(offs) // 2, len(offs) - 2):
        ck = dst[int(offs[i]):int(offs[i + 1])].cpu().numpy()
        assert np.array_equal(oracle.blosc_decompress(ck), res.G[i * chunk_nbytes:(i + 1) * chunk_nbytes].cpu().numpy())


def test_full_size_500k_x_5000_plane_path(ctx):
    """The path bench.py's C4 leg times, at full size: bit planes appended piece by piece at a device cursor (k_encode_planes +
    k_encode_general for the GT:DP lines, planes with missing calls), hhgt_pad_tail_planes_cursor, hhgt_compress_planes (the
    exception-aware bit-plane coder).  Checks: counts, POS table, 1500 sampled columns against the generator's call rule,
    value set, every chunk GPU-decoded == expanded planes, three chunks through the CPU oracle's decoder — and the first
    6 000 variants (one piece of their own) against oracle.vcf_encode."""
    S, V = 5000, 500_000
    seed = 4
    t = synth.mixed_table(seed, V, S)
    kept = np.nonzero(t["kept"])[0]
    d = ctx.device
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=d)

    def new_result(lay):
        cap = lay.v_capacity
        r = dev.EncodeResult(None, lay, z(cap, torch.int32), z(cap, torch.int32), z(cap, torch.uint8), z(cap, torch.uint8), 0, {}, [],
                             torch.full((dev.planes_bytes(lay),), 0x3C, dtype=torch.uint8, device=d))
        ctx.pad_tail_planes(r, cap, 0, cap // lay.vc)      # the sample padding rows (S .. round_up(S, 64)) are nobody's to write: zeroed once
        return r

    # (a) a slice against the oracle
    nsl = 6000
    sub = {k: (v[:nsl] if isinstance(v, np.ndarray) and len(v) == V else v) for k, v in t.items()}
    text, n, _ = ctx.synth_mixed("chr4", sub, S, seed=seed, v_first=0, with_header=True)
    o = oracle.vcf_encode(text.cpu().numpy(), S, region="chr4", cap=nsl)
    lay0 = dev.make_layout(S, max(o["n_kept"], 1))
    r0 = new_result(lay0)
    c0 = z(1, torch.int64)
    rec = ctx.encode_text_planes_async(text, S, r0, c0, max_lines=nsl + 64, region="chr4").wait()
    ctx.pad_tail_planes_cursor(r0, c0)
    assert rec.stats.n_kept == o["n_kept"] and rec.reserved == 0
    G0 = dev.EncodeResult(ctx.planes_expand(r0), lay0, None, None, None, None, o["n_kept"], {}).dense().cpu().numpy()
    assert np.array_equal(G0, o["G"])
    assert np.array_equal(r0.start[:o["n_kept"]].cpu().numpy().view(np.uint32), o["start"])
    del r0, G0, text
    # (b) the full size
    lay = dev.make_layout(S, len(kept))
    cap = lay.v_capacity
    res = new_result(lay)
    cursor = z(1, torch.int64)
    piece = 100_000
    tot = dict(n_records=0, n_drop_filter=0, n_general_lines=0, n_haploid_padded=0, n_kept=0)
    for a in range(0, V, piece):
        sub = {k: (v[a:a + piece] if isinstance(v, np.ndarray) and len(v) == V else v) for k, v in t.items()}
        text, n, _ = ctx.synth_mixed("chr4", sub, S, seed=seed, v_first=a, with_header=(a == 0))
        assert n < (1 << 32)
        rec = ctx.encode_text_planes_async(text, S, res, cursor, max_lines=piece + 64, region="chr4").wait()
        assert rec.reserved == 0
        for k in tot:
            tot[k] += getattr(rec.stats, k)
        del text
    ctx.pad_tail_planes_cursor(res, cursor)
    assert int(cursor.item()) == len(kept) == tot["n_kept"] and tot["n_records"] == V and tot["n_drop_filter"] == V - len(kept)
    assert tot["n_general_lines"] <= 5 and tot["n_haploid_padded"] == 0     # (the GT:DP records go through the tile kernel at their stride)
    assert np.array_equal(res.start[:len(kept)].cpu().numpy().view(np.uint32) + 1, t["pos"][kept])
    dst, off, total = ctx.compress_planes(res, fmt=dev.BLOSC2)
    G = ctx.planes_expand(res)
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(len(kept), 1500, replace=False))
    exp = synth.mixed_expected_G(seed, t, S, kept[pick])
    Sc, Vc = lay.sc, lay.vc
    n_sc, n_vc = -(-S // Sc), cap // Vc
    Gv = G.view(torch.int8).view(n_vc, n_sc, Sc, Vc, 2)
    pk = torch.from_numpy(pick).to(d)
    got = Gv[pk // Vc, :, :, pk % Vc, :].reshape(len(pick), n_sc * Sc, 2)[:, :S].permute(1, 0, 2).cpu().numpy()
    assert np.array_equal(got, exp)
    g8 = G.view(torch.int8)
    assert int(g8.min()) == -9 and int(g8.max()) == 1 and not bool(((g8 != 0) & (g8 != 1) & (g8 != -9)).any())
    del g8
    chunk_nbytes = Sc * Vc * 2
    n_chunks = G.numel() // chunk_nbytes
    back, bad = ctx.decompress(dst, off, n_chunks, chunk_nbytes, typesize=2, blocksize=8192)
    assert bad == 0 and torch.equal(back, G)
    assert G.numel() / total > 4.0
    offs = off.cpu().numpy()
    for i in (0, n_chunks // 2, n_chunks - 1):
        ck = dst[int(offs[i]):int(offs[i + 1])].cpu().numpy()
        assert np.array_equal(oracle.blosc_decompress(ck), G[i * chunk_nbytes:(i + 1) * chunk_nbytes].cpu().numpy())
