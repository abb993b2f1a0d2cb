#!/bin/bash
# development: the two-stream step under wave-priority / LDS-room variants
cd ${GRAFT_REPO_ROOT:-$PWD}
B="python3 bench.py --steps 8 --warmup 2 --no-legs --no-cpu-baseline --no-other-configs --no-check"
run() { echo "== $1"; shift; env "$@" timeout -k 10 200 $B 2>/dev/null | python3 tools/bench_line.py x /dev/stdin; }
run "default"                          A=1
run "prio3"                            HHGT_LIB=$PWD/build/variants/libhhgt_prio3.so
run "prio3, no CU mask"                HHGT_LIB=$PWD/build/variants/libhhgt_prio3.so HHGT_COMPRESS_CUS=0
run "prio3, no mask, LZ4 13 WG/CU"     HHGT_LIB=$PWD/build/variants/libhhgt_prio3.so HHGT_COMPRESS_CUS=0 HHGT_LZ4_LDS_PAD=1100
run "prio3, no mask, LZ4 11 WG/CU"     HHGT_LIB=$PWD/build/variants/libhhgt_prio3.so HHGT_COMPRESS_CUS=0 HHGT_LZ4_LDS_PAD=3600
run "prio3, no mask, LZ4 9 WG/CU"      HHGT_LIB=$PWD/build/variants/libhhgt_prio3.so HHGT_COMPRESS_CUS=0 HHGT_LZ4_LDS_PAD=6900
run "default lib, no mask, 11 WG/CU"   HHGT_COMPRESS_CUS=0 HHGT_LZ4_LDS_PAD=3600
