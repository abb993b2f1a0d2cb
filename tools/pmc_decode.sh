#!/bin/bash
# SQ instruction-mix counters for the decode kernel on tools/decode_bench.py (one pass per counter set).
# usage (on the GPU box): bash tools/pmc_decode.sh <tag>
set -e
tag=${1:-x}
out=$PWD/gpurun_out/pmcdec_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES \
  --kernel-include-regex k_decode --kernel-trace -d $out/a -o a --output-format csv -- python3 $R/tools/decode_bench.py 60000 > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-include-regex k_decode --kernel-trace -d $out/b -o b --output-format csv -- python3 $R/tools/decode_bench.py 60000 > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for s in "ab":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % s, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter(); grid = 0
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; grid += int(r["Grid_Size"])
        k = max(n.values()); streams = grid / len(acc) / 64
        print(s, "launches", k, "streams(waves)", streams)
        for c, v in sorted(acc.items()): print("  %-24s %14.0f  per stream %10.1f" % (c, v, v / streams))
PY
