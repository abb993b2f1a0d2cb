#!/bin/bash
# development: tools/dev/enc_time.py for the in-tree library and every build/variants/*.so
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 120 python3 tools/dev/enc_time.py planes int8
for so in build/variants/*.so; do
  HHGT_LIB=$PWD/$so timeout -k 10 120 python3 tools/dev/enc_time.py planes
done
