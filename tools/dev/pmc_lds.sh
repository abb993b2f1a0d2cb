#!/bin/bash
# development: LDS-side SQ counters of k_lz4_bitplanes (is the LDS array / its bank conflicts what the waves wait for?)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_lds
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --variants 300000 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-legs --no-check --no-other-configs"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES \
  --kernel-include-regex k_lz4_bitplanes -d $O -o a --output-format csv -- $B > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES \
  --kernel-include-regex k_lz4_bitplanes -d $O -o b --output-format csv -- $B > $O/b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --kernel-include-regex k_lz4_bitplanes -d $O -o c --output-format csv -- $B > $O/c.log 2>&1
python3 - <<'PY'
import csv, collections, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_lds"
for tag in "abc":
    for f in glob.glob(f"{O}/{tag}_counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            w = d.get("SQ_WAVES") or 1.0
            print(tag, k, {c: round(v, 1) for c, v in sorted(d.items())})
PY
