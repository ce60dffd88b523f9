#!/bin/bash
# tools/ab_libs.sh "<bench args>" ROUNDS name1 name2 ...: the same bench command with each _ab_libs/<name>.so in turn
args="$1"; rounds="$2"; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
for r in $(seq 1 "$rounds"); do
  for n in "$@"; do
    out=$(cd "$root" && DSW_HIP_LIB="$root/_ab_libs/$n.so" python bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "$n $args: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4))')"
  done
done
