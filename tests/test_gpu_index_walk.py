"""-m gpu: the walking line index (csrc/index.hip, k_index_hop<U, true>; include/hhgt.h hhgt_set_index_mode) — the head of
each record says where its sample columns start, and with FORMAT == "GT" where its newline must be — against the plain scan
of every byte, the hop by the bound, and the oracle: wide random text (every FORMAT / GT shape of tests/test_gpu_fuzz.py at
cohort widths), heads that do not fit the 1 KiB window, records near the end of the text, and the one input shape on which
the walk trusts a newline that is not the record's own."""
import numpy as np
import pytest

from oracle import oracle
from haplohyped_varawareml_amd import synth
from haplohyped_varawareml_amd._lib import HhgtError
from tests.gpu_util import assert_same_as_oracle, gpu_encode
from tests.test_gpu_fuzz import make_text

pytestmark = pytest.mark.gpu


def encode_modes(ctx, text, S, region, modes=(0, 1, 2)):
    out = {}
    try:
        for m in modes:
            ctx.set_index_mode(m)
            out[m] = gpu_encode(ctx, text, S, region=region)
    finally:
        ctx.set_index_mode(-1)
    return out


def same(a, b):
    assert a["n_kept"] == b["n_kept"] and a["stats"] == b["stats"]
    for k in ("G", "start", "stop", "ref", "alt"):
        assert np.array_equal(a[k], b[k]), k
    assert a["res"].chrom_runs == b["res"].chrom_runs


@pytest.mark.parametrize("seed", range(8))
def test_wide_random_text_every_mode(ctx, seed):
    rng = np.random.default_rng(4000 + seed)
    S = int(rng.choice([760, 761, 800, 1023, 1100, 2049]))
    text = make_text(rng, S, int(rng.integers(40, 120)), fixed_share=float(rng.choice([0.0, 0.5, 0.9, 1.0])))
    for region in ("", "chrA"):
        o = oracle.vcf_encode(text, S, region=region)
        g = encode_modes(ctx, text, S, region)
        for m in g:
            assert_same_as_oracle(g[m], o)
        same(g[0], g[2])
        same(g[1], g[2])


def wide_lines(S, V, seed=7, contig="chr7"):
    text, _ = synth.render_fixed_numpy(contig, synth.variant_table(seed, V, S), S, seed=seed)
    lines = bytes(text).split(b"\n")
    hdr = [x for x in lines if x.startswith(b"#")]
    return hdr, [x for x in lines if x and not x.startswith(b"#")]


@pytest.mark.parametrize("S", [800, 2504])
def test_fixed_width_shard_every_mode(ctx, S):
    """the shape the walk is for: every record costs one 1 KiB head"""
    hdr, rec = wide_lines(S, 900 if S == 800 else 400)
    t = b"\n".join(hdr + rec) + b"\n"
    g = encode_modes(ctx, t, S, "chr7")
    assert g[2]["n_kept"] == len(rec) and g[2]["stats"]["n_general_lines"] == 0
    assert_same_as_oracle(g[2], oracle.vcf_encode(t, S, region="chr7"))
    same(g[0], g[2])
    same(g[1], g[2])


@pytest.mark.parametrize("S", [800, 2100])
def test_heads_of_every_length(ctx, S):
    """INFO columns from 1 byte to several KiB: nine tabs inside the 1 KiB head, just inside, just outside, far outside"""
    hdr, rec = wide_lines(S, 64)
    out = []
    for k, ln in enumerate(rec):
        f = ln.split(b"\t")
        pad = [1, 200, 900, 960, 975, 985, 990, 1000, 1010, 1500, 5000, 70000][k % 12]
        f[7] = b"X=" + b"q" * pad
        if k % 7 == 3:
            f[8] = b"GT:DP"
            f[9:] = [x + b":%d" % (k % 50) for x in f[9:]]
        out.append(b"\t".join(f))
    t = b"\n".join(hdr + out) + b"\n"
    g = encode_modes(ctx, t, S, "chr7")
    assert_same_as_oracle(g[2], oracle.vcf_encode(t, S, region="chr7"))
    same(g[0], g[2])


@pytest.mark.parametrize("S", [800, 2100])
@pytest.mark.parametrize("tail", ["newline", "none", "crlf", "blank"])
def test_records_near_the_end_of_the_text(ctx, S, tail):
    """the last records of a text: their heads and candidates reach past the last KiB"""
    hdr, rec = wide_lines(S, 5)
    for n in (1, 2, 5):
        t = b"\n".join(hdr + rec[:n])
        t = {"newline": t + b"\n", "none": t, "crlf": t.replace(b"\n", b"\r\n") + b"\r\n", "blank": t + b"\n\n"}[tail]
        g = encode_modes(ctx, t, S, "chr7", modes=(0, 2))
        assert_same_as_oracle(g[2], oracle.vcf_encode(t, S, region="chr7"))
        same(g[0], g[2])


@pytest.mark.parametrize("S", [800, 2100])
def test_format_that_only_starts_with_gt(ctx, S):
    """FORMAT columns "GTX", "G", "GT:GT", "TG" and a FORMAT of "GT" whose calls are not three bytes wide"""
    hdr, rec = wide_lines(S, 12)
    out = []
    for k, ln in enumerate(rec):
        f = ln.split(b"\t")
        if k % 4 == 1:
            f[8] = b"GT:GQ"
            f[9:] = [x + b":9" for x in f[9:]]
        elif k % 4 == 2:
            f[9:] = [b"0" if (i + k) % 3 == 0 else (b"10|1" if (i + k) % 3 == 1 else x) for i, x in enumerate(f[9:])]
        elif k % 4 == 3:
            f[8] = b"DP:GT"
            f[9:] = [b"7:" + x for x in f[9:]]
        out.append(b"\t".join(f))
    t = b"\n".join(hdr + out) + b"\n"
    g = encode_modes(ctx, t, S, "chr7")
    assert_same_as_oracle(g[2], oracle.vcf_encode(t, S, region="chr7"))
    same(g[0], g[2])
    for bad in (b"GTX", b"TG", b"G"):          # no GT key: vcfpp.h:550-552 "genotypes not present"
        f = rec[3].split(b"\t")
        f[8] = bad
        t2 = b"\n".join(hdr + rec[:3] + [b"\t".join(f)] + rec[4:]) + b"\n"
        for m in (0, 2):
            ctx.set_index_mode(m)
            try:
                with pytest.raises(HhgtError, match="Error parsing VCF file"):
                    gpu_encode(ctx, t2, S, region="chr7")
            finally:
                ctx.set_index_mode(-1)


@pytest.mark.parametrize("S", [800, 2100])
def test_newline_of_another_line_where_the_head_points(ctx, S):
    """The one shape on which the walk takes a newline that is not the record's own: a FORMAT == "GT" record whose sample
    columns are ONE byte shorter than S diploid calls (one haploid call and one two-digit allele), followed by an empty
    line — the byte at soff + 4 S - 1 is then the empty line's newline.  The merged record is kept, its fields do not
    match "a|b\\t", the variable-width encoder finds the newline inside it: the pass FAILS with HHGT_ERR_MALFORMED (flagged,
    never a different matrix; the same rule as for records shorter than the hop's bound) — which is what the
    asynchronous form reports; the synchronous call then scans every byte (as hhgt_set_index_mode 0 does from the start) and
    decodes what the oracle decodes."""
    hdr, rec = wide_lines(S, 10)
    f = rec[4].split(b"\t")
    f[9 + 17] = b"1"
    f[9 + 40] = b"10|1"
    odd = b"\t".join(f)
    assert len(odd) == len(rec[4]) - 1
    t = b"\n".join(hdr + rec[:4] + [odd, b""] + rec[5:]) + b"\n"
    try:
        want = oracle.vcf_encode(t, S, region="chr7")
    except Exception:
        want = None
    import torch
    from haplohyped_varawareml_amd import device as dev
    from tests.gpu_util import to_dev
    try:
        ctx.set_index_mode(2)
        lay = dev.make_layout(S, 128, sc=64, vc=128)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
        res = dev.EncodeResult(z(dev.layout_bytes(lay), torch.uint8), lay, z(lay.v_capacity, torch.int32), z(lay.v_capacity, torch.int32),
                               z(lay.v_capacity, torch.uint8), z(lay.v_capacity, torch.uint8), 0, {})
        with pytest.raises(HhgtError, match="Error parsing VCF file"):       # one pass (the asynchronous form): flagged
            ctx.encode_text_async(to_dev(t), S, res, z(1, torch.int64), max_lines=200, region="chr7").wait()
        if want is not None:                                                 # the synchronous call scans every byte then
            assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr7"), want)
        ctx.set_index_mode(0)
        if want is not None:
            assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr7"), want)
    finally:
        ctx.set_index_mode(-1)
    # without the empty line nothing is special: the candidate is not a newline, the search finds the record's own
    t3 = b"\n".join(hdr + rec[:4] + [odd] + rec[5:]) + b"\n"
    g = encode_modes(ctx, t3, S, "chr7", modes=(0, 2))
    assert_same_as_oracle(g[2], oracle.vcf_encode(t3, S, region="chr7"))
    same(g[0], g[2])
