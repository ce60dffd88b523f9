#!/bin/bash
# usage: tools/pmc_pair_small.sh <tag> <bench_pair_one args...> : time + LDS counters of one recurrence configuration
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
python $root/tools/bench_pair_one.py "$@" 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmcs_${tag}
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out -o pmc -- python $root/tools/bench_pair_one.py "$@" > $out/run.log 2>&1
cd $root
python - "$out" <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(spmm2_\w+_kernel<[^>]*>)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print("   ", k, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
