#!/bin/bash
# Collects everything profiles/ holds for one build, on the GPU box:
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
#   1. the default bench line                                  (bench_line.json)
#   2. rocprofv3 --kernel-trace --stats on the same command    (<tag>_kernel_stats.csv, bench_line_under_rocprof.json)
#   3. FETCH_SIZE and WRITE_SIZE, separate --pmc passes, small workload (f_/w_counter_collection.csv)
#   4. SQ instruction-mix counters of the LZ4 kernels          (sq_counter_collection.csv)
#   5. the end-to-end lines of tools/e2e_bench.py               (e2e_*.json)
# tools/summarize_profiles.py turns 3 into the per-variant byte counts DESIGN.md quotes.
set -e
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py > $O/bench.log 2>&1
grep "^{" $O/bench.log > $O/bench_line.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o $tag --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-legs --no-check --no-other-configs > $O/rocprof_stdout.log 2>&1
grep "^{" $O/rocprof_stdout.log > $O/bench_line_under_rocprof.json
B="python3 $R/bench.py --variants 300000 --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-legs --no-check --no-other-configs"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O -o f --output-format csv -- $B > $O/out_f.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O -o w --output-format csv -- $B > $O/out_w.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU \
  --kernel-include-regex k_lz4 -d $O -o sq --output-format csv -- $B > $O/out_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU \
  --kernel-include-regex k_encode_planes -d $O -o sqe --output-format csv -- $B > $O/out_sqe.log 2>&1
# the config-4 leg (exception-aware bit-plane coder, variable-width path) under the kernel trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c4 --output-format csv -- python3 $R/bench.py --only-config C4 > $O/c4_line.json 2> $O/c4.log
cd $R
# (the end-to-end legs are part of the default bench line since round 3: every shard, first / steady pass)
ls $O
