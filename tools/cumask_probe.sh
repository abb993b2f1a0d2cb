#!/bin/bash
# development: does splitting the CUs between the HBM-bound encode stream and the issue-bound LZ4 stream pay?  And:
# keeping the LZ4 stream off a few CUs, so that the latency-bound small kernels of the encode stream (k_parse_fixed:
# 146 us per launch beside LZ4, 40 us alone) always find free slots?
run() { timeout -k 5 120 python bench.py "$@" --no-legs --no-cpu-baseline --no-check --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,1), 'M/s', {k: round(v['ms']/3,1) if isinstance(v,dict) else round(v,1) for k,v in d['stages_ms_per_step_timed_region'].items()})" | tee -a gpurun_out/cumask.log; }
run
run --cu-split 256,256
run --cu-split 256,248
run --cu-split 256,240
run --cu-split 256,224
run --cu-split 256,208
run --cu-split 256,192
run --cu-split 128,128
run --cu-split 96,160
