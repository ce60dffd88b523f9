#!/bin/bash
cd /root/repo
for r in 1 2; do
for n in product old; do
  if [ $n = product ]; then python tools/bench_ns_dual.py 64 16 2>/dev/null | tail -3 | tr '\n' ' ' | sed "s/^/$n: /"; echo
  else DSW_HIP_LIB=/root/repo/_ab_libs/$n.so python tools/bench_ns_dual.py 64 16 2>/dev/null | tail -3 | tr '\n' ' ' | sed "s/^/$n: /"; echo; fi
done; done
