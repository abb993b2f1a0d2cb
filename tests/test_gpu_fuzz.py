"""-m gpu: randomised VCF text (deterministic seeds) through the HIP path vs the oracle: odd GT strings,
multi-digit alleles, missing sub-fields, FORMAT with GT in any position, long INFO, CRLF, blank lines,
header lines in the middle, contig changes, sample counts around the tile edges."""
import numpy as np
import pytest

from oracle import oracle
from tests.gpu_util import assert_same_as_oracle, gpu_encode

pytestmark = pytest.mark.gpu

GT_POOL = ["0|0", "0|1", "1|0", "1|1", "./.", ".|.", ".|1", "0|.", "0/1", "1/1", "2|1", "10|3", "0", "1", ".",
           "0|1|1", "1|", "|1", "", "0|0:3", "1|1:.", "7/12"]
FMT_POOL = ["GT", "GT:DP", "GT:GQ:DP", "DP:GT", "AD:DP:GT"]
ALT_POOL = ["A", "C", "G", "T", "A", "C", "G", "T", "AT", "A,C", "*", "<DEL>", "a", ".", "N"]
REF_POOL = ["A", "C", "G", "T", "N", "AC", "a"]


def make_text(rng, S, n_lines, fixed_share):
    names = [f"s{i}" for i in range(S)]
    out = ["##fileformat=VCFv4.2", "#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + names)]
    pos = 100
    contig = "chrA"
    for i in range(n_lines):
        r = rng.random()
        if r < 0.02:
            out.append("")                       # blank line
            continue
        if r < 0.03:
            out.append("##late header line")
            continue
        if r < 0.06:
            contig = rng.choice(["chrA", "chrB", "chrLongName_1"])
        pos += int(rng.integers(1, 5000))
        info = "." if rng.random() < 0.7 else "AC=%d;AF=0.5;XS=%s" % (rng.integers(0, 99), "z" * int(rng.integers(0, 300)))
        if rng.random() < fixed_share:
            fmt = "GT"
            cols = [rng.choice(["0|0", "0|1", "1|0", "1|1", "./.", ".|1", "0/1"]) for _ in range(S)]
            ref, alt = rng.choice(["A", "C", "G", "T"]), rng.choice(["A", "C", "G", "T"])
        else:
            fmt = rng.choice(FMT_POOL)
            gi = fmt.split(":").index("GT")
            nsub = len(fmt.split(":"))
            cols = []
            for _ in range(S):
                g = rng.choice(GT_POOL).split(":")[0]
                sub = [str(int(rng.integers(0, 100))) for _ in range(nsub)]
                sub[gi] = g
                if rng.random() < 0.1:
                    sub = sub[: gi + 1]          # trailing sub-fields dropped
                cols.append(":".join(sub))
            ref, alt = rng.choice(REF_POOL), rng.choice(ALT_POOL)
        out.append("\t".join([contig, str(pos), rng.choice([".", "rs%d" % i]), ref, alt, ".", "PASS", info, fmt] + cols))
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    text = eol.join(out)
    if rng.random() < 0.7:
        text += eol
    return text.encode()


@pytest.mark.parametrize("seed", range(12))
def test_random_text_matches_oracle(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 300]))
    text = make_text(rng, S, int(rng.integers(50, 600)), fixed_share=float(rng.choice([0.0, 0.5, 0.9])))
    for region in ("", "chrA", "chrB:1000-900000"):
        o = oracle.vcf_encode(text, S, region=region, want_chrom=True)
        g = gpu_encode(ctx, text, S, region=region, sc=int(rng.choice([0, 8, 64])), vc=int(rng.choice([0, 128, 256])) or 0)
        assert_same_as_oracle(g, o)
        # CHROM runs reproduce the per-record CHROM column
        runs = g["res"].chrom_runs
        got = [None] * g["n_kept"]
        bounds = [r[0] for r in runs] + [g["n_kept"]]
        for (a, name), b in zip(runs, bounds[1:]):
            got[a:b] = [name] * (b - a)
        assert got == o["chrom"]
