#!/usr/bin/env python3
"""Derives the golden genotype vectors from the reference's OWN fixture data
(tests/golden/chr22.filtered.vcf.gz, copied byte-for-byte from
/root/reference/tests/data/chr22.filtered.vcf.gz) with an INDEPENDENT pure-Python
tab/colon/pipe splitter -- no oracle code, no reference code is run.

Semantics applied (the ones the oracle must reproduce):
  keep record iff len(REF)==1 and ALT in {A,C,G,T}          cpp/vcfpp.h:990-1000
  GT = first ':' sub-field; alleles split on '|' or '/';
  '.' -> -9 else int(allele); narrowed to int8               cpp/vcfpp.h:546-588, cpp/parse_vcf.cpp:51-52
  start = POS-1, stop = start+len(REF)                       cpp/vcfpp.h:1118-1127
Writes tests/golden/fixture_golden.json and tests/golden/fixture_G.npy.
"""
import gzip, hashlib, json, os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

def main():
    raw = gzip.open(os.path.join(HERE, "chr22.filtered.vcf.gz"), "rb").read()
    samples, rows = None, []
    for line in raw.decode().split("\n"):
        if not line:
            continue
        if line.startswith("#CHROM"):
            samples = line.split("\t")[9:]
            continue
        if line.startswith("#"):
            continue
        f = line.split("\t")
        chrom, pos, _id, ref, alt = f[:5]
        if len(ref) != 1 or alt not in ("A", "C", "G", "T"):
            continue
        fmt = f[8].split(":")
        gi = fmt.index("GT")
        calls = []
        for col in f[9:]:
            gt = col.split(":")[gi]
            al = gt.replace("/", "|").split("|")
            al = [(-9 if a == "." else int(a)) for a in al]
            assert len(al) == 2
            calls.append(al)
        rows.append((chrom, int(pos) - 1, int(pos) - 1 + len(ref), ref, alt, calls))
    S, V = len(samples), len(rows)
    G = np.zeros((S, V, 2), np.int8)
    for v, r in enumerate(rows):
        for s in range(S):
            G[s, v, 0] = np.int8(r[5][s][0]); G[s, v, 1] = np.int8(r[5][s][1])
    np.save(os.path.join(HERE, "fixture_G.npy"), G)
    gold = {
        "source_sha256": hashlib.sha256(open(os.path.join(HERE, "chr22.filtered.vcf.gz"), "rb").read()).hexdigest(),
        "text_bytes": len(raw),
        "samples": samples,
        "n_records": V,
        "G_sha256": hashlib.sha256(G.tobytes()).hexdigest(),
        "phase_sums": [[int(G[s, :, 0].sum()), int(G[s, :, 1].sum())] for s in range(S)],
        "start_first3": [r[1] for r in rows[:3]],
        "start_last": rows[-1][1],
        "start_sha256": hashlib.sha256(np.array([r[1] for r in rows], np.uint32).tobytes()).hexdigest(),
        "stop_minus_start": sorted(set(r[2] - r[1] for r in rows)),
        "ref_sha256": hashlib.sha256("".join(r[3] for r in rows).encode()).hexdigest(),
        "alt_sha256": hashlib.sha256("".join(r[4] for r in rows).encode()).hexdigest(),
        "chrom_set": sorted(set(r[0] for r in rows)),
        "first_tuple_sample0": [rows[0][0], rows[0][1], rows[0][2], rows[0][3], rows[0][4], int(G[0,0,0]), int(G[0,0,1])],
    }
    json.dump(gold, open(os.path.join(HERE, "fixture_golden.json"), "w"), indent=1)
    print(json.dumps(gold, indent=1))

if __name__ == "__main__":
    main()
