#!/bin/bash
# Same-box A/B of the working tree against a copy of an older tree (built under _ab_old/, not tracked):
#   tools/ab_run.sh "<workload args>" [rounds]
# prints ms_per_step of old / new alternately.
args="$1"; rounds="${2:-2}"
root="$(cd "$(dirname "$0")/.." && pwd)"
for r in $(seq 1 "$rounds"); do
  for side in _ab_old .; do
    out=$(cd "$root/$side" && python bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "$side $args: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
