#!/bin/bash
# standalone sensitivity
for pad in 0 2048 4096 6144; do
  HHGT_LZ4_LDS_PAD=$pad python tools/lz4_bench.py --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lz4 pad $pad', round(d['stages_ms']['lz4'],3))"
done
for pad in 0 16384 40000 100000; do
  HHGT_ENC_LDS_PAD=$pad python bench.py --no-overlap --no-legs --no-cpu-baseline --no-check --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enc pad $pad', {k: round(v,1) for k,v in d['stages_ms_per_step'].items()}, round(d['ms_per_step'],1))"
done
# combined
for cfg in "0 0 no" "0 0 yes" "4096 0 yes" "6144 0 yes" "4096 16384 yes" "6144 16384 yes" "6144 40000 yes" "4096 0 no" "6144 0 no"; do
  set -- $cfg
  F=""; if [ $3 = yes ]; then F="--lz4-priority"; fi
  HHGT_LZ4_LDS_PAD=$1 HHGT_ENC_LDS_PAD=$2 python bench.py $F --no-legs --no-cpu-baseline --no-check --steps 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lz4pad $1 encpad $2 lz4prio $3:', round(d['ms_per_step'],2), 'ms', round(d['value']/1e6,1), 'M/s')"
done
