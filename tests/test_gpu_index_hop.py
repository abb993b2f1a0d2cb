"""-m gpu: the hopping newline index (csrc/index.hip, k_index_hop) — used when every record must be at least 1.5 KB long
(S >= 760 samples) — against the oracle, on the cases where skipping the front of every line could go wrong: header and
empty lines between records, a text without a final newline, variable-width lines, line ends that sit exactly on the
bound, lines across the 16 KiB / 128 KiB range borders, a file cut off in mid-line, a short line in mid-file."""
import numpy as np
import pytest

from oracle import oracle
from haplohyped_varawareml_amd import synth
from haplohyped_varawareml_amd._lib import HhgtError
from tests.gpu_util import assert_same_as_oracle, gpu_encode

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 2], ids=["hop", "walk"])
def index_mode(ctx, request):
    """every case under both long-line index kernels: the hop by the bound (round 2) and the walk (round 4, the default)"""
    ctx.set_index_mode(request.param)
    yield request.param
    ctx.set_index_mode(-1)


def shard(S, V, seed=5, contig="chr5"):
    text, _ = synth.render_fixed_numpy(contig, synth.variant_table(seed, V, S), S, seed=seed)
    return bytes(text)


@pytest.mark.parametrize("S,V", [(800, 700), (1100, 500), (2100, 300), (2504, 257)])   # k_index_hop<3> / <5>
def test_fixed_width_matches_oracle(ctx, S, V):
    text = shard(S, V)
    g = gpu_encode(ctx, text, S, region="chr5")
    assert g["n_kept"] == V
    assert_same_as_oracle(g, oracle.vcf_encode(text, S, region="chr5"))


def body_lines(text):
    lines = text.split(b"\n")
    hdr = [x for x in lines if x.startswith(b"#")]
    return hdr, [x for x in lines if x and not x.startswith(b"#")]


@pytest.mark.parametrize("S", [800, 2100])
def test_no_final_newline_and_crlf(ctx, S):
    text = shard(S, 40)
    for t in (text[:-1], text.replace(b"\n", b"\r\n")):
        assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr5"), oracle.vcf_encode(t, S, region="chr5"))


@pytest.mark.parametrize("S", [800, 2100])
def test_header_and_empty_lines_between_records(ctx, S):
    """'#' lines and empty lines are short: the index must not hop behind them, wherever they stand"""
    hdr, rec = body_lines(shard(S, 30))
    mixed = hdr + rec[:5] + [b"##comment in the middle"] + rec[5:9] + [b"#x", b"#y"] + rec[9:20] + hdr[-1:] + rec[20:]
    t = b"\n".join(mixed) + b"\n"
    assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr5"), oracle.vcf_encode(t, S, region="chr5"))
    # empty lines: whatever the encoder says about them, it says the same with and without the hop (oracle: ignored or
    # an error) — compare the outcome
    t2 = b"\n".join(hdr + rec[:3] + [b""] + rec[3:]) + b"\n\n"
    try:
        want = oracle.vcf_encode(t2, S, region="chr5")
    except Exception:
        want = None
    if want is None:
        with pytest.raises(HhgtError):
            gpu_encode(ctx, t2, S, region="chr5")
    else:
        assert_same_as_oracle(gpu_encode(ctx, t2, S, region="chr5"), want)


@pytest.mark.parametrize("S", [800, 2100])
def test_variable_width_lines(ctx, S):
    """general-path lines of every length from the shortest a record with S samples can have (haploid calls: the newline
    sits right behind the bound) to long multi-digit / annotated ones, across range borders"""
    rng = np.random.default_rng(S)
    hdr = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + \
        b"\t".join(f"s{i}".encode() for i in range(S)) + b"\n"
    lines = []
    for v in range(260):
        kind = v % 5
        if kind == 0:      # shortest possible record: 1-byte columns, haploid calls
            fields = [b"1" if rng.random() < 0.3 else b"0" for _ in range(S)]
            fixed = b"5\t%d\t.\tA\tC\t.\t.\t.\tGT" % (v + 1)
        elif kind == 1:    # diploid fixed width
            fields = [b"%d|%d" % (rng.integers(0, 2), rng.integers(0, 2)) for _ in range(S)]
            fixed = b"5\t%d\trs%d\tA\tC\t.\tPASS\t.\tGT" % (v + 1, v)
        elif kind == 2:    # annotated
            fields = [b"%d|%d:%d:%d" % (rng.integers(0, 2), rng.integers(0, 2), rng.integers(0, 99), rng.integers(0, 999)) for _ in range(S)]
            fixed = b"5\t%d\t.\tG\tT\t50\tPASS\tAC=%d;AN=%d\tGT:GQ:DP" % (v + 1, v, 2 * S)
        elif kind == 3:    # missing / unphased
            fields = [[b"./.", b"0/1", b".|1", b"1|1"][rng.integers(0, 4)] for _ in range(S)]
            fixed = b"5\t%d\t.\tC\tA\t.\tPASS\t.\tGT" % (v + 1)
        else:              # long INFO in front
            fields = [b"%d|%d" % (rng.integers(0, 2), rng.integers(0, 2)) for _ in range(S)]
            fixed = b"5\t%d\t.\tT\tC\t.\tPASS\t" % (v + 1) + b"X=" + b"a" * int(rng.integers(1, 9000)) + b"\tGT"
        lines.append(fixed + b"\t" + b"\t".join(fields))
    t = hdr + b"\n".join(lines) + b"\n"
    g = gpu_encode(ctx, t, S, region="5")
    assert g["n_kept"] == 260
    assert_same_as_oracle(g, oracle.vcf_encode(t, S, region="5"))


@pytest.mark.parametrize("S", [800, 2100])
def test_truncated_file_is_reported(ctx, S):
    """a file cut off in mid-line ends in a line shorter than the bound: it reaches the parser and is reported"""
    text = shard(S, 50)
    cut = text[:len(text) - (2 * S + 300)]           # the last line keeps less than half of its sample columns
    with pytest.raises(HhgtError, match="Error parsing VCF file"):
        gpu_encode(ctx, cut, S, region="chr5")
    with pytest.raises(Exception):
        oracle.vcf_encode(cut, S, region="chr5")


@pytest.mark.parametrize("S", [800, 2100])
def test_short_line_in_mid_file_is_reported(ctx, S):
    """a kept record with fewer sample columns than the header declares: its newline lies inside a hop, the line merges
    with the next one, and the general encoder reports the newline it finds inside the sample columns"""
    hdr, rec = body_lines(shard(S, 40))
    short = b"\t".join(rec[10].split(b"\t")[:9 + S // 3])
    t = b"\n".join(hdr + rec[:10] + [short] + rec[11:]) + b"\n"
    with pytest.raises(HhgtError, match="Error parsing VCF file"):
        gpu_encode(ctx, t, S, region="chr5")


@pytest.mark.parametrize("S", [800, 2100])
def test_line_with_empty_sample_columns(ctx, S, index_mode):
    """A kept record whose sample columns are mostly EMPTY (tab tab tab ...) is shorter than 2 S + 17 bytes: valid for the oracle
    (empty columns are missing calls, -9), and at cohort widths shorter than the line index assumes any record to be.  The hop
    merges such a line with its successor, and so does the walk unless the newline lies inside the 1 KiB head it reads; the
    encoders then find the newline inside the merged record and the pass fails with HHGT_ERR_MALFORMED — flagged, never a
    silently different matrix.  Round 4: the synchronous call (hhgt_encode_text, what parse_vcf uses) then runs once more with
    every byte scanned and returns what the oracle returns; the asynchronous form reports the first pass's error."""
    hdr, rec = body_lines(shard(S, 40))
    cols = rec[10].split(b"\t")
    empty = b"\t".join(cols[:9 + 5] + [b""] * (S - 5))       # 5 calls, then S - 5 empty columns: ~S + 80 bytes
    assert len(empty) < 2 * S + 17
    t = b"\n".join(hdr + rec[:10] + [empty] + rec[11:]) + b"\n"
    o = oracle.vcf_encode(t, S, region="chr5")               # the oracle takes it
    assert o["n_kept"] == 40 and (o["G"][5:, 10] == -9).all()
    assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr5"), o)
    # the asynchronous form: one pass, the index's outcome
    import torch
    from haplohyped_varawareml_amd import device as dev
    from tests.gpu_util import to_dev
    lay = dev.make_layout(S, 128, sc=64, vc=128)
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
    res = dev.EncodeResult(z(dev.layout_bytes(lay), torch.uint8), lay, z(lay.v_capacity, torch.int32), z(lay.v_capacity, torch.int32),
                           z(lay.v_capacity, torch.uint8), z(lay.v_capacity, torch.uint8), 0, {})
    pend = ctx.encode_text_async(to_dev(t), S, res, z(1, torch.int64), max_lines=200, region="chr5")
    if index_mode == 2 and len(empty) < 1000:
        assert pend.wait().stats.n_kept == 40              # the walk sees the newline inside the head it reads
    else:
        with pytest.raises(HhgtError, match="Error parsing VCF file"):
            pend.wait()


def _as_indel(line):
    f = line.split(b"\t")
    f[3], f[4] = b"AT", b"A"          # REF of two bases: dropped by isSNP (cpp/vcfpp.h:990-1000)
    return b"\t".join(f)


@pytest.mark.parametrize("S", [1000, 2100])
@pytest.mark.parametrize("how", ["indel", "other_contig"])
def test_short_line_that_the_filter_drops_is_reported(ctx, S, how):
    """A record with too few sample columns that the isSNP / region filter DROPS: nobody reads its sample columns, and at
    cohort widths its newline lies in the part the hopping index jumps over — the valid record behind it used to vanish
    with it, unreported (round 2's review).  k_parse_fixed now looks at the skipped bytes of every dropped record: the
    call fails with HHGT_ERR_MALFORMED, as it does for a kept short line (htslib: a parse error either way)."""
    hdr, rec = body_lines(shard(S, 40))
    victim = _as_indel(rec[10]) if how == "indel" else rec[10].replace(b"chr5\t", b"chr6\t", 1)
    short = b"\t".join(victim.split(b"\t")[:9 + S // 3])
    assert len(short) < 2 * S + 17
    t = b"\n".join(hdr + rec[:10] + [short] + rec[11:]) + b"\n"
    # the pass with the hop / walk fails (asynchronous form) ...
    import torch
    from haplohyped_varawareml_amd import device as dev
    from tests.gpu_util import to_dev
    lay = dev.make_layout(S, 128, sc=64, vc=128)
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=ctx.device)
    res = dev.EncodeResult(z(dev.layout_bytes(lay), torch.uint8), lay, z(lay.v_capacity, torch.int32), z(lay.v_capacity, torch.int32),
                           z(lay.v_capacity, torch.uint8), z(lay.v_capacity, torch.uint8), 0, {})
    with pytest.raises(HhgtError, match="Error parsing VCF file"):
        ctx.encode_text_async(to_dev(t), S, res, z(1, torch.int64), max_lines=200, region="chr5").wait()
    # ... and the synchronous call, which then scans every byte, says what the oracle says about a dropped short record
    try:
        want = oracle.vcf_encode(t, S, region="chr5")
    except Exception:
        want = None
    if want is None:
        with pytest.raises(HhgtError, match="Error parsing VCF file"):
            gpu_encode(ctx, t, S, region="chr5")
    else:
        assert_same_as_oracle(gpu_encode(ctx, t, S, region="chr5"), want)


@pytest.mark.parametrize("S", [1000, 2100])
def test_full_length_dropped_lines_are_not_reported(ctx, S):
    """... and dropped records of full length (indels, another contig) between kept ones change nothing"""
    hdr, rec = body_lines(shard(S, 40))
    mixed = list(rec)
    for k in (3, 4, 17, 39):
        mixed[k] = _as_indel(rec[k])
    mixed[20] = rec[20].replace(b"chr5\t", b"chr6\t", 1)
    t = b"\n".join(hdr + mixed) + b"\n"
    g = gpu_encode(ctx, t, S, region="chr5")
    assert g["n_kept"] == 35
    assert_same_as_oracle(g, oracle.vcf_encode(t, S, region="chr5"))
