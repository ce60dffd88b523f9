#!/bin/bash
# usage: tools/prof_pmc_cmd.sh <tag> <kernel regex> -- <command...>
# SQ-level PMC passes for an arbitrary command (each pass its own rocprofv3 run, --kernel-trace only).
tag=$1; kre=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "A FETCH_SIZE" "B WRITE_SIZE" "D TCC_HIT_sum TCC_MISS_sum" "C SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "E SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_LDS" "F SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM"; do
  set -- $pass; p=$1; shift
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $p "; then continue; fi
  out=$root/gpurun_out/pmc_${tag}_$p
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $out -o pmc -- ${CMD} > $out/run.log 2>&1
  echo "pass $p ($@): rc=$?"
done
cd $root
python - "$tag" "$kre" <<'PY'
import csv, glob, sys, collections, os, re
tag, kre = sys.argv[1], sys.argv[2]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(kre, r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%3d avg=%.4g min=%.4g max=%.4g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
PY
