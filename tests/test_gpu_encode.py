"""-m gpu: the HIP encode path (through the C ABI) against the CPU oracle and the golden vectors.
Bar: bit-exact (integer / byte work)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from haplohyped_varawareml_amd import synth
from haplohyped_varawareml_amd._lib import HhgtError
from tests.gpu_util import assert_same_as_oracle, gpu_encode, to_dev

pytestmark = pytest.mark.gpu

HDR = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n"


def test_fixture_golden(ctx, fixture_text, fixture_golden, golden_dir):
    g = gpu_encode(ctx, fixture_text, 3, region="chr22")
    assert g["n_kept"] == 1000
    assert hashlib.sha256(g["G"].tobytes()).hexdigest() == fixture_golden["G_sha256"]
    assert np.array_equal(g["G"], np.load(os.path.join(golden_dir, "fixture_G.npy")))
    assert g["start"][:3].tolist() == fixture_golden["start_first3"]
    assert hashlib.sha256(g["start"].tobytes()).hexdigest() == fixture_golden["start_sha256"]
    assert_same_as_oracle(g, oracle.vcf_encode(fixture_text, 3, region="chr22"))
    assert g["res"].chrom_runs == [(0, "chr22")]
    # the fixture's columns are "a|b:GQ:DP": every line takes the variable-width path
    assert g["stats"]["n_general_lines"] == 1000


@pytest.mark.parametrize("region", ["", "chr22", "chr21", "chr22:10026999-10045798", "chr22:19000000"])
def test_fixture_regions(ctx, fixture_text, region):
    assert_same_as_oracle(gpu_encode(ctx, fixture_text, 3, region=region),
                          oracle.vcf_encode(fixture_text, 3, region=region))


KNOWN = [
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t./.\t.|1\t0/1\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT:GQ:DP\t1|0:12:.\t0|0:5:7\t1|1:.:.\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tDP:GT\t7:1|0\t8:0|1\t9:.|.\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t10|2\t200|1\t0|0\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t1\t.\t0|1\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t1|\t|1\t0|1|1\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t1|0\t1|1",                 # fixed width, no trailing newline
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t1|0\t1|1\r\n",
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t1x0\t1|1\n",               # right length, wrong separator
    "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t\t1|1:5\n",                # empty column
    "chr1\t10\t.\tA\tC,G\t.\tPASS\t.\tGT\t1|2\t0|0\t2|1\nchr1\t20\t.\tAT\tA\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\n"
    "chr1\t30\t.\tA\tAT\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\nchr1\t40\t.\tA\t*\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\n"
    "chr1\t50\t.\tA\ta\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\nchr1\t60\t.\tA\t<DEL>\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\n"
    "chr1\t70\t.\tA\t.\t.\tPASS\t.\tGT\t0|0\t0|0\t0|0\nchr1\t80\t.\tN\tT\t.\tPASS\t.\tGT\t0|1\t1|0\t1|1\n"
    "chr2\t90\t.\tG\tA\t.\tPASS\t.\tGT\t1|1\t0|0\t0|1\n",
]


@pytest.mark.parametrize("body", KNOWN)
@pytest.mark.parametrize("region", ["", "chr1"])
def test_known_answer_lines(ctx, body, region):
    text = (HDR + body).encode()
    assert_same_as_oracle(gpu_encode(ctx, text, 3, region=region), oracle.vcf_encode(text, 3, region=region))


def test_framing_edge_cases(ctx):
    line = "chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t1|0\t1|1"
    for text in (HDR + line, HDR + line + "\n", HDR + line + "\r\n", HDR + "\n" + line + "\n\n",
                 (HDR + line + "\n").replace("\n", "\r\n"), HDR, "", "\n\n\n"):
        t = text.encode()
        assert_same_as_oracle(gpu_encode(ctx, t, 3), oracle.vcf_encode(t, 3))


def test_malformed_raises(ctx):
    # the reference raises RuntimeError("Error parsing VCF file: ...") (cpp/parse_vcf.cpp:63-66)
    for body in ("chr1\t100\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\n",          # too few sample columns
                 "chr1\t100\t.\tA\tC\t.\tPASS\t.\tDP\t1\t2\t3\n",      # FORMAT without GT
                 "chr1\tx\t.\tA\tC\t.\tPASS\t.\tGT\t0|1\t0|1\t0|1\n",  # POS not a number
                 "chr1\t100\t.\tA\n"):
        with pytest.raises(HhgtError, match="Error parsing VCF file"):
            gpu_encode(ctx, (HDR + body).encode(), 3)
        with pytest.raises(RuntimeError):
            oracle.vcf_encode((HDR + body).encode(), 3)


def test_sites_only(ctx):
    txt = ("##x\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\nchr1\t5\t.\tA\tG\t.\t.\t.\n"
           "chr1\t9\t.\tAC\tG\t.\t.\t.\nchr2\t11\t.\tT\tC\t.\t.\t.\n").encode()
    g = gpu_encode(ctx, txt, 0)
    o = oracle.vcf_encode(txt, 0)
    assert g["n_kept"] == o["n_kept"] == 2 and np.array_equal(g["start"], o["start"])
    assert g["res"].chrom_runs == [(0, "chr1"), (1, "chr2")]


@pytest.mark.parametrize("S,V", [(1, 300), (3, 257), (5, 1000), (255, 129), (256, 128), (257, 500), (1001, 700),
                                 (2504, 300)])
@pytest.mark.parametrize("dense", [False, True])
def test_fixed_width_vs_oracle(ctx, S, V, dense):
    tab = synth.variant_table(100 + S, V, S)
    text, _ = synth.render_fixed_numpy("chr9", tab, S, seed=100 + S)
    kw = dict(sc=0, vc=0) if dense else {}
    g = gpu_encode(ctx, text, S, region="chr9", **kw)
    assert g["stats"]["n_general_lines"] == 0          # all lines on the tile kernel
    assert_same_as_oracle(g, oracle.vcf_encode(text, S, region="chr9"))
    assert np.array_equal(g["start"] + 1, tab["pos"])


def test_synth_gpu_renderer_matches_numpy(ctx):
    for S, V in ((7, 50), (1000, 300), (2504, 64)):
        tab = synth.variant_table(22, V, S)
        ref_text, _ = synth.render_fixed_numpy("chr22", tab, S, seed=22)
        t, n = ctx.synth_fixed("chr22", tab, S, seed=22)
        assert n == len(ref_text)
        assert bytes(t.cpu().numpy()) == ref_text


@pytest.mark.parametrize("crlf,trail", [(False, True), (True, True), (False, False)])
def test_mixed_c4_style_vs_oracle(ctx, crlf, trail):
    S = 37
    text = synth.render_mixed("chr4", 700, S, seed=4, crlf=crlf, trailing_newline=trail)
    g = gpu_encode(ctx, text, S, region="chr4")
    o = oracle.vcf_encode(text, S, region="chr4")
    assert o["stats"]["n_drop_filter"] > 0 and g["stats"]["n_general_lines"] > 0
    assert_same_as_oracle(g, o)


def test_append_batches(ctx):
    """streaming: consecutive text blocks appended at v_base (not tile aligned) == one-shot encode"""
    from haplohyped_varawareml_amd import device as dev
    S, V = 300, 1000
    tab = synth.variant_table(5, V, S)
    text, off = synth.render_fixed_numpy("chr5", tab, S, seed=5)
    o = oracle.vcf_encode(text, S, region="chr5")
    cuts = [0, int(off[77]), int(off[77 + 333]), int(off[900]), len(text)]
    lay = dev.make_layout(S, V, sc=64, vc=256)
    cap = lay.v_capacity
    junk = lambda n, dt: torch.full((n,), 0x55, dtype=dt, device="cuda")   # padding must not rely on zero-init
    res = dev.EncodeResult(junk(dev.layout_bytes(lay), torch.uint8), lay, junk(cap, torch.int32),
                           junk(cap, torch.int32), junk(cap, torch.uint8), junk(cap, torch.uint8), 0, {})
    v_base = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        t = to_dev(text[a:b])
        res = ctx.encode_text(t, S, region="chr5", v_base=v_base, out=res)
        v_base = res.n_kept
    assert res.n_kept == V
    assert np.array_equal(res.dense().cpu().numpy(), o["G"])
    assert np.array_equal(res.start[:V].cpu().numpy().view(np.uint32), o["start"])
    # padding of the last chunk column / sample-chunk is zero after pad_tail
    ctx.pad_tail(res)
    torch.cuda.synchronize()
    full = res.G.view(torch.int8).view(-1, 5, 64, 256, 2).cpu().numpy()
    assert full.shape[0] == 4 and not full[3, :, :, V - 3 * 256:, :].any()
    assert not full[:, 4, 300 - 256:, :, :].any()


def test_capacity_error(ctx):
    from haplohyped_varawareml_amd import device as dev
    S, V = 8, 600
    tab = synth.variant_table(6, V, S)
    text, _ = synth.render_fixed_numpy("chr6", tab, S, seed=6)
    with pytest.raises(HhgtError, match="exceed v_capacity"):
        ctx.encode_text(to_dev(text), S, layout=dev.make_layout(S, 256, sc=8, vc=256))


def test_config2_chr22_50k_x_1000(ctx):
    """BASELINE.json configs[1]: synthetic chr22, 50 000 variants x 1000 samples, biallelic phased"""
    S, V = 1000, 50_000
    tab = synth.variant_table(22, V, S)
    t, n = ctx.synth_fixed("chr22", tab, S, seed=22)
    g = gpu_encode(ctx, t, S, region="chr22")
    o = oracle.vcf_encode(t.cpu().numpy(), S, region="chr22", cap=V)
    assert o["n_kept"] == V
    assert_same_as_oracle(g, o)
    # size-independent property: every '1' character of the GT columns is exactly one set allele
    ones_in_text = int((t == ord("1")).sum().item())
    G = g["res"].dense()
    fixed_ones = ones_in_text - int(G.to(torch.int64).sum().item())
    pos_ones = sum(str(int(p)).count("1") for p in tab["pos"]) + synth.header_text("chr22", synth.sample_names(S)).count(b"1")
    assert fixed_ones == pos_ones
