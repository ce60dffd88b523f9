#!/bin/bash
cd /root/repo
for cfg in "8 4" "16 8" "32 8" "32 16" "64 4" "64 16"; do
  for dual in 1 0; do
    DSW_HIP_LIB=/root/repo/_ab_libs/diag.so DSW_BWD_DUAL=$dual python tools/bench_ns_dual.py $cfg --time-only 2>/dev/null | tail -1 | sed "s/^/nside,B=$cfg dual=$dual /"
  done
done
