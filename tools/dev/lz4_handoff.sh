#!/bin/bash
# development (round 4): what an in-launch hand-off would cost k_lz4_bitplanes — every wave drains its stores and draws one
# returning atomic on a word shared by its chunk before it ends (-DBP_PROBE_HANDOFF=1), against the default build.
# build first:  python -c "from haplohyped_varawareml_amd import build as b; b.build(lib='build/variants/libhhgt_handoff.so', extra_flags=['-DBP_PROBE_HANDOFF=1'])"
cd ${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2; do
  echo "== default"; python3 tools/lz4_bench.py --reps 20 2>/dev/null | tail -1
  echo "== hand-off probe"; HHGT_LIB=$PWD/build/variants/libhhgt_handoff.so python3 tools/lz4_bench.py --reps 20 2>/dev/null | tail -1
done
