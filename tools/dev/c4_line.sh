#!/bin/bash
# development: the config-4 leg alone, essentials of its line
cd ${GRAFT_REPO_ROOT:-$PWD}
python3 bench.py --only-config C4 $C4_ARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.load(sys.stdin); print(round(d['value']/1e6,2),'Mvar/s', round(d['ms_per_pass'],2),'ms', {k:round(v,2) for k,v in d['stages_ms'].items()}, 'ratio', round(d['compression_ratio'],3))"
