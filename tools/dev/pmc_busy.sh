#!/bin/bash
# development: derived busy metrics of the LZ4 kernel (which issue port is it on?)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_busy
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py --variants 300000 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-legs --no-check --no-other-configs"
for set in "VALUBusy SALUBusy" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" "LDSBankConflict MemUnitStalled" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_FLAT SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-include-regex "k_lz4_bitplanes" -d $O -o $tag --output-format csv -- $B > $O/$tag.log 2>&1 || echo "set failed: $set"
done
python3 - <<'PY'
import csv, collections, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_busy"
for f in sorted(glob.glob(f"{O}/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "true, false" not in k: continue
        print(os.path.basename(f)[:20], {c: (round(sum(v)/len(v), 2), len(v)) for c, v in sorted(d.items())})
PY
