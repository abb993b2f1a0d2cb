#!/bin/bash
# development: kernel trace of a device-inflate end-to-end pass (where is the GPU idle?)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/e2e_trace
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O -o e2e --output-format csv -- python3 $R/tools/e2e_bench.py --variants 3000000 --kind bgzf --device-inflate --repeat 3 > $O/line.json 2> $O/err.log
tail -1 $O/line.json
python3 $R/tools/trace_gaps.py $(find $O -name "e2e_kernel_trace.csv" | head -1) 2>&1 | head -60
