#!/bin/bash
# SQ instruction-mix counters for any kernel under any of the tools/ scripts (one pass per counter set).
# usage (on the GPU box): bash tools/pmc_kernel.sh <tag> <kernel regex> <script.py> [args...]
set -e
tag=$1; rx=$2; shift 2
out=$PWD/gpurun_out/pmck_$tag
mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES \
  --kernel-include-regex "$rx" --kernel-trace -d $out/a -o a --output-format csv -- python3 $R/tools/"$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-include-regex "$rx" --kernel-trace -d $out/b -o b --output-format csv -- python3 $R/tools/"$@" > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for s in "ab":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % s, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter(); grid = 0
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1; grid += int(r["Grid_Size"])
        k = max(n.values()); waves = grid / len(acc) / 64
        print(s, "launches", k, "waves", waves)
        for c, v in sorted(acc.items()): print("  %-24s %14.0f  per wave %10.1f" % (c, v, v / waves))
PY
