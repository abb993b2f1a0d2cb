#!/bin/bash
# development: single-stream stage times for the in-tree library and every build/variants/*.so
cd ${GRAFT_REPO_ROOT:-$PWD}
B="python3 bench.py --steps 6 --warmup 2 --no-legs --no-cpu-baseline --no-other-configs --no-check"
echo "== in-tree"; timeout -k 10 200 $B 2>/dev/null | python3 tools/bench_line.py x /dev/stdin
for so in build/variants/*.so; do echo "== $so"; HHGT_LIB=$PWD/$so timeout -k 10 200 $B 2>/dev/null | python3 tools/bench_line.py x /dev/stdin; done
