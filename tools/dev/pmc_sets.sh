#!/bin/bash
# development: a few --pmc passes over tools/dev/enc_time.py for one kernel regex; prints per-kernel sums
# usage (GPU box): bash tools/dev/pmc_sets.sh <tag> <kernel regex> [enc_time.py args]
tag=$1; rx=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmcs_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "MemUnitBusy MemUnitStalled" "VALUBusy SALUBusy"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-include-regex "$rx" --kernel-trace -d $out/s$i -o s --output-format csv -- python3 $R/tools/dev/enc_time.py "$@" > $out/s$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/s*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:40], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"]); n[key] += 1
    for (k, c), v in sorted(acc.items()):
        print("%-42s %-30s sum %16.0f  launches %3d  per launch %14.1f" % (k, c, v, n[(k, c)], v / n[(k, c)]))
PY
