#!/bin/bash
# development: workgroup id -> (variant tile, sample band) mappings of k_encode_planes
cd ${GRAFT_REPO_ROOT:-$PWD}
B="python3 bench.py --steps 6 --warmup 2 --no-legs --no-cpu-baseline --no-other-configs"
for m in 0 1 2 0 1 2; do echo "== HHGT_ENC_MAP=$m"; HHGT_ENC_MAP=$m timeout -k 10 200 $B 2>/dev/null | python3 tools/bench_line.py x /dev/stdin; done
