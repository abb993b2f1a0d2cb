"""The labelled NON-REFERENCE filter mode of SURVEY.md §8(d) C4 (hhgt_set_keep_multiallelic): multi-allelic SNP sites
pass the record filter and their genotypes carry the allele index as int8 (cpp/vcfpp.h:574), where the reference's
isSNP (cpp/vcfpp.h:990-1000) drops every record with more than two alleles.  CPU: the oracle's mode against answers
written out by hand.  GPU: the HIP path against the oracle in that mode, and the default mode unchanged."""
import numpy as np
import pytest

from oracle import oracle
from haplohyped_varawareml_amd import synth

HDR = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n"
BODY = ("chr1\t10\t.\tA\tC,G\t.\tPASS\t.\tGT\t1|2\t0|0\t2|1\n"          # kept in the mode: alleles 0..2
        "chr1\t20\t.\tA\tC,G,T\t.\tPASS\t.\tGT\t3|0\t./.\t2/3\n"       # three ALTs
        "chr1\t30\t.\tAT\tA,C\t.\tPASS\t.\tGT\t0|1\t0|0\t0|0\n"        # REF longer than one base: dropped in both
        "chr1\t40\t.\tA\tC,GT\t.\tPASS\t.\tGT\t0|1\t0|2\t0|0\n"        # an ALT that is not a single base: dropped
        "chr1\t50\t.\tA\tC,<DEL>\t.\tPASS\t.\tGT\t0|1\t0|2\t0|0\n"     # symbolic ALT: dropped
        "chr1\t60\t.\tA\tC,*\t.\tPASS\t.\tGT\t0|1\t0|2\t0|0\n"         # '*': dropped
        "chr1\t70\t.\tA\tc,G\t.\tPASS\t.\tGT\t0|1\t0|2\t0|0\n"         # lower case: dropped (the filter is exact)
        "chr1\t80\t.\tA\tC,\t.\tPASS\t.\tGT\t0|1\t0|2\t0|0\n"          # trailing comma: dropped
        "chr1\t90\t.\tG\tT\t.\tPASS\t.\tGT\t1|1\t0|0\t0|1\n"           # plain SNP: kept in both
        "chr1\t100\t.\tG\tT,A\t.\tPASS\t.\tGT:DP\t2|1:5\t0|0:7\t10|2:1\n")  # annotated, a two-digit index


def test_oracle_mode_known_answers():
    text = (HDR + BODY).encode()
    ref_mode = oracle.vcf_encode(text, 3)
    assert ref_mode["start"].tolist() == [89]                      # the reference keeps the one biallelic SNP
    o = oracle.vcf_encode(text, 3, keep_multiallelic=True)
    assert o["start"].tolist() == [9, 19, 89, 99]
    assert bytes(o["ref"]) == b"AAGG" and bytes(o["alt"]) == b"CCTT"    # alt[] = the first ALT base
    assert o["G"][:, 0].tolist() == [[1, 2], [0, 0], [2, 1]]
    assert o["G"][:, 1].tolist() == [[3, 0], [-9, -9], [2, 3]]
    assert o["G"][:, 3].tolist() == [[2, 1], [0, 0], [10, 2]]
    # the switch does not stick
    assert oracle.vcf_encode(text, 3)["n_kept"] == 1


@pytest.mark.gpu
def test_gpu_known_answers_and_default_unchanged(ctx):
    from tests.gpu_util import assert_same_as_oracle, gpu_encode
    text = (HDR + BODY).encode()
    assert_same_as_oracle(gpu_encode(ctx, text, 3), oracle.vcf_encode(text, 3))
    ctx.set_keep_multiallelic(True)
    try:
        g = gpu_encode(ctx, text, 3)
        assert g["n_kept"] == 4
        assert_same_as_oracle(g, oracle.vcf_encode(text, 3, keep_multiallelic=True))
    finally:
        ctx.set_keep_multiallelic(False)
    assert gpu_encode(ctx, text, 3)["n_kept"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("S,V", [(40, 3000), (900, 400)])
def test_gpu_c4_mixture_in_keep_mode(ctx, S, V):
    """C4's mixture (10 % multi-allelic, missing and half-missing calls, '/' separators, GT:DP columns, indels): in the
    mode the multi-allelic SNP sites are kept with their allele indices; the byte planes with indices > 1 go through
    the byte-wise LZ4 coder and decode back"""
    from tests.gpu_util import assert_same_as_oracle, gpu_encode
    text = synth.render_mixed("chr4", V, S, seed=4)
    ref_mode = oracle.vcf_encode(text, S, region="chr4")
    ctx.set_keep_multiallelic(True)
    try:
        g = gpu_encode(ctx, text, S, region="chr4")
        o = oracle.vcf_encode(text, S, region="chr4", keep_multiallelic=True)
        assert_same_as_oracle(g, o)
        assert o["n_kept"] > ref_mode["n_kept"] and int(o["G"].max()) >= 2
        res = g["res"]
        ctx.pad_tail(res)
        lay = res.layout
        chunk = lay.sc * lay.vc * 2
        dst, off, total = ctx.compress(res.G, chunk)
        back, bad = ctx.decompress(dst, off, res.G.numel() // chunk, chunk)
        assert bad == 0 and bool((back == res.G).all())
    finally:
        ctx.set_keep_multiallelic(False)
