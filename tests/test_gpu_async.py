"""-m gpu: the asynchronous encode chain (hhgt_encode_text_async: device-resident append cursor, caller's bound on
the line count, pinned result record) and the ring layout the streaming ingest is built on — against the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.gpu_util import to_dev
from haplohyped_varawareml_amd import device as dev, synth

pytestmark = pytest.mark.gpu


def _blocks(text, n):
    """n line-aligned pieces of a VCF text"""
    nl = [i + 1 for i, b in enumerate(text) if b == 10]
    cuts = [0] + [nl[(len(nl) * (k + 1)) // n - 1] for k in range(n)]
    return [text[cuts[k]:cuts[k + 1]] for k in range(n) if cuts[k + 1] > cuts[k]]


def _new_result(ctx, lay):
    d = ctx.device
    cap = lay.v_capacity
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=d)
    return dev.EncodeResult(z(max(dev.layout_bytes(lay), 16), torch.uint8), lay, z(cap, torch.int32), z(cap, torch.int32),
                            z(cap, torch.uint8), z(cap, torch.uint8), 0, {})


@pytest.mark.parametrize("S,V,nblk", [(300, 5000, 3), (64, 900, 5), (2504, 700, 2)])
def test_chain_appends_like_one_call(ctx, S, V, nblk):
    tab = synth.variant_table(3, V, S)
    text, _ = synth.render_fixed_numpy("chr3", tab, S, seed=3)
    o = oracle.vcf_encode(text, S, region="chr3")
    lay = dev.make_layout(S, V, sc=64, vc=1024)
    res = _new_result(ctx, lay)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    pend = []
    keep = []
    for blk in _blocks(text, nblk):
        t = to_dev(blk)
        keep.append(t)
        pend.append(ctx.encode_text_async(t, S, res, cursor, region="chr3"))      # no host wait between the calls
    ctx.pad_tail_cursor(res, cursor)
    recs = [p.wait() for p in pend]
    assert int(cursor.item()) == V == recs[-1].cursor_after
    assert [r.cursor_before for r in recs] == [0] + [r.cursor_after for r in recs[:-1]]
    assert sum(r.stats.n_kept for r in recs) == V and sum(r.stats.n_records for r in recs) == o["stats"]["n_records"]
    assert recs[0].chrom_runs() == [(0, "chr3")] and all(r.stats.n_chrom_runs == 1 for r in recs)
    res.n_kept = V
    assert np.array_equal(res.dense().cpu().numpy(), o["G"])
    assert np.array_equal(res.start[:V].cpu().numpy().view(np.uint32), o["start"])
    assert np.array_equal(res.ref[:V].cpu().numpy(), o["ref"]) and np.array_equal(res.alt[:V].cpu().numpy(), o["alt"])
    # padding behind the cursor is zero up to the end of its chunk column
    vcol_end = -(-V // 1024) * 1024
    full = dev.EncodeResult(res.G, lay, res.start, res.stop, res.ref, res.alt, vcol_end, {})
    assert not full.dense()[:, V:].any()


def test_chain_mixed_lines_and_region(ctx):
    """variable-width lines (general path), dropped records, a region with a range: the chain equals the oracle"""
    text = synth.render_mixed("chr4", 1500, 40, seed=4)
    o = oracle.vcf_encode(text, 40, region="chr4:1-400000")
    lay = dev.make_layout(40, 1500, sc=0, vc=0)
    res = _new_result(ctx, lay)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    pend, keep = [], []
    for blk in _blocks(text, 4):
        keep.append(to_dev(blk))
        pend.append(ctx.encode_text_async(keep[-1], 40, res, cursor, region="chr4:1-400000", max_lines=keep[-1].numel() // 16 + 8))
    recs = [p.wait() for p in pend]
    n = int(cursor.item())
    assert n == o["n_kept"] and sum(r.stats.n_drop_filter for r in recs) == o["stats"]["n_drop_filter"]
    assert sum(r.stats.n_drop_region for r in recs) == o["stats"]["n_drop_region"]
    assert sum(r.stats.n_haploid_padded for r in recs) == o["stats"]["n_haploid_padded"]
    res.n_kept = n
    assert np.array_equal(res.dense().cpu().numpy(), o["G"])
    assert np.array_equal(res.start[:n].cpu().numpy().view(np.uint32), o["start"])


def test_ring_wraps(ctx):
    """G and the tables as a ring of 3 chunk columns; each block's completed columns are read before later blocks
    overwrite them — together they are the oracle's matrix"""
    S, V, vc = 100, 4000, 256
    tab = synth.variant_table(9, V, S)
    text, _ = synth.render_fixed_numpy("chr9", tab, S, seed=9)
    o = oracle.vcf_encode(text, S, region="chr9")
    ring = 4
    lay = dev.make_ring_layout(S, ring, sc=64, vc=vc)
    res = _new_result(ctx, lay)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    n_sc = 2
    col_bytes = n_sc * 64 * vc * 2
    got = np.zeros((n_sc * 64, -(-V // vc) * vc, 2), np.int8)
    starts = np.zeros(-(-V // vc) * vc, np.uint32)
    done = 0
    for blk in _blocks(text, 20):                      # ~200 variants per block: less than a column
        t = to_dev(blk)
        rec = ctx.encode_text_async(t, S, res, cursor, region="chr9").wait()
        assert rec.v_capacity == 0
        now = rec.cursor_after // vc
        for col in range(done, now):
            slot = col % ring
            g = res.G[slot * col_bytes:(slot + 1) * col_bytes].view(torch.int8).view(n_sc, 64, vc, 2).reshape(n_sc * 64, vc, 2)
            got[:, col * vc:(col + 1) * vc] = g.cpu().numpy()
            starts[col * vc:(col + 1) * vc] = res.start[slot * vc:(slot + 1) * vc].cpu().numpy().view(np.uint32)
        done = now
    assert int(cursor.item()) == V and done == V // vc
    ctx.pad_tail_cursor(res, cursor)
    slot = done % ring
    g = res.G[slot * col_bytes:(slot + 1) * col_bytes].view(torch.int8).view(n_sc, 64, vc, 2).reshape(n_sc * 64, vc, 2)
    got[:, done * vc:(done + 1) * vc] = g.cpu().numpy()
    starts[done * vc:(done + 1) * vc] = res.start[slot * vc:(slot + 1) * vc].cpu().numpy().view(np.uint32)
    assert np.array_equal(got[:S, :V], o["G"]) and not got[S:].any() and not got[:, V:].any()
    assert np.array_equal(starts[:V], o["start"])


def test_bound_too_small_is_reported(ctx):
    S, V = 50, 400
    tab = synth.variant_table(2, V, S)
    text, _ = synth.render_fixed_numpy("chr2", tab, S, seed=2)
    lay = dev.make_layout(S, V, sc=0, vc=0)
    res = _new_result(ctx, lay)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    p = ctx.encode_text_async(to_dev(text), S, res, cursor, region="chr2", max_lines=100)
    with pytest.raises(dev.HhgtError, match="beyond max_lines") as e:
        p.wait()
    assert e.value.code == -3 and p.rec.n_lines_over == V + 7 - 100      # 7 header lines


def test_async_malformed_and_capacity(ctx):
    bad = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\nchr1\tx\t.\tA\tC\t.\t.\t.\tGT\t0|1\n"
    lay = dev.make_layout(1, 128, sc=0, vc=0)
    res = _new_result(ctx, lay)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    with pytest.raises(dev.HhgtError, match="Error parsing VCF file"):
        ctx.encode_text_async(to_dev(bad), 1, res, cursor).wait()
    # kept records beyond a linear layout's capacity
    S, V = 4, 600
    tab = synth.variant_table(2, V, S)
    text, _ = synth.render_fixed_numpy("chr2", tab, S, seed=2)
    small = dev.make_layout(S, 256, sc=0, vc=0)
    res = _new_result(ctx, small)
    cursor = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    with pytest.raises(dev.HhgtError, match="exceed v_capacity"):
        ctx.encode_text_async(to_dev(text), S, res, cursor).wait()


def test_compress_async_needs_the_bound(ctx):
    src = torch.zeros(4 * 65536, dtype=torch.uint8, device=ctx.device)
    small = torch.empty(4000, dtype=torch.uint8, device=ctx.device)
    with pytest.raises(dev.HhgtError, match="hhgt_compress_bound"):
        ctx.compress(src, 65536, dst=small, sync=False)
    dst, off, total = ctx.compress(src, 65536, dst=small, sync=True)      # the synchronous form sizes exactly
    assert total <= 4000


def test_stream_pair_helpers(ctx):
    """hhgt_stream_create: the encode / compress stream pair (the compress stream carries a CU mask) runs the same
    kernels to the same bytes as the default stream"""
    from haplohyped_varawareml_amd import synth
    S, V = 200, 9000
    tab = synth.variant_table(9, V, S)
    text, _ = ctx.synth_fixed("chr9", tab, S, seed=9)
    lay = dev.make_layout(S, V)
    ref = ctx.encode_text(text, S, region="chr9", layout=lay)
    ctx.pad_tail(ref)
    chunk = lay.sc * lay.vc * 2
    d0, o0, t0 = ctx.compress(ref.G, chunk)
    s_enc, s_cmp = ctx.create_stream("encode"), ctx.create_stream("compress")
    with torch.cuda.stream(s_enc):
        res = ctx.encode_text(text, S, region="chr9", layout=lay)
        ctx.pad_tail(res)
        done = s_enc.record_event()
    with torch.cuda.stream(s_cmp):
        s_cmp.wait_event(done)
        d1, o1, t1 = ctx.compress(res.G, chunk)
    torch.cuda.synchronize()
    assert t1 == t0 and torch.equal(o0, o1) and torch.equal(d0[:t0], d1[:t1]) and torch.equal(ref.G, res.G)
