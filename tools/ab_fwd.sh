#!/bin/bash
# tools/ab_fwd.sh name1 name2 ...: time of the forward without basis stores at the north-star shape with each _ab_libs/<name>.so ("product" = the product library)
root="$(cd "$(dirname "$0")/.." && pwd)"
for n in "$@"; do
  if [ "$n" = product ]; then (cd "$root" && python tools/bench_ns_dual.py 64 16 --fwd-only 2>/dev/null | tail -1)
  else (cd "$root" && DSW_HIP_LIB="$root/_ab_libs/$n.so" python tools/bench_ns_dual.py 64 16 --fwd-only 2>/dev/null | tail -1); fi
done
