#!/bin/bash
# usage: tools/prof_stats.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ (rocprofv3 kernel stats csv)
# rocprofv3 has been seen to hang at process teardown on the GPU boxes: always under a hard timeout.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python $root/bench.py "$@" > $out/run.log 2>&1
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "== $tag: $f"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-70s calls=%-5s avg_us=%9.2f min_us=%9.2f max_us=%9.2f pct=%s" % (
        r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
grep metric $out/run.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_step', round(d['ms_per_step'],4), d.get('roofline'))"
