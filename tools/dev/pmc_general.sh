#!/bin/bash
# development: SQ counters of k_encode_general on the config-4 leg
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmc_general
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --kernel-include-regex k_encode_general --kernel-trace -d $out/a -o a --output-format csv -- python3 $R/bench.py --only-config C4:100000 > $out/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
  --kernel-include-regex k_encode_general --kernel-trace -d $out/b -o b --output-format csv -- python3 $R/bench.py --only-config C4:100000 > $out/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for s in "ab":
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % s, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        print(s, "launches", max(n.values()))
        for c, v in sorted(acc.items()): print("  %-24s %16.0f  per launch %14.1f" % (c, v, v / max(n.values())))
PY
