#!/bin/bash
# usage: tools/pmc_pair.sh <tag> <bench_pair_one args...> : PMC passes for one recurrence configuration
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out -o pmc -- python $root/tools/bench_pair_one.py "$@" > $out/run.log 2>&1
  echo "pass $i ($pass): rc=$?"
done
cd $root
python - "$tag" <<'PY'
import csv, glob, sys, collections, os, re
tag = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(spmm2_\w+_kernel<[^>]*>)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
