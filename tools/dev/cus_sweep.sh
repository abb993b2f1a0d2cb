#!/bin/bash
# development: the two-stream step for different sizes of the compress stream's CU mask (default: 192 of 256)
cd ${GRAFT_REPO_ROOT:-$PWD}
B="python3 bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-other-configs --no-check"
for cus in 192 224 160 128 0; do echo "== HHGT_COMPRESS_CUS=$cus"; HHGT_COMPRESS_CUS=$cus timeout -k 10 200 $B 2>/dev/null | python3 tools/bench_line.py x /dev/stdin; done
for sp in "64,192" "96,192" "128,192" "64,224" "96,224"; do echo "== --cu-split $sp"; timeout -k 10 200 $B --cu-split $sp 2>/dev/null | python3 tools/bench_line.py x /dev/stdin; done
