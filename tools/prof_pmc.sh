#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <bench args...>
# PMC passes (each in its own rocprofv3 run, --kernel-trace only, as the MI355X guide prescribes):
#   pass A: FETCH_SIZE (3 TCC slots)   pass B: WRITE_SIZE   pass C: MFMA busy / active cycles
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "A FETCH_SIZE" "B WRITE_SIZE" "C SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "D TCC_HIT_sum TCC_MISS_sum" "E SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_LDS"; do
  set -- $pass; p=$1; shift
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $p "; then continue; fi
  out=$root/gpurun_out/pmc_${tag}_$p
  mkdir -p $out
  timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $out -o pmc -- python $root/bench.py ${BENCH_ARGS:---steps 10 --warmup 3 --no-cpu-baseline} > $out/run.log 2>&1
  echo "pass $p ($@): rc=$?"
done
cd $root
python - "$tag" <<'PY'
import csv, glob, sys, collections, os
tag = sys.argv[1]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(ts_gemm_x3s?_kernel<[^>]*>|ts_gemm_kernel<[^>]*>|cheb_wgrad\w*_kernel<[^>]*>|spmm_csr_rowsplit<[^>]*>|remap_\w+_kernel<[^>]*>|cheb3_bwd_fused_kernel|cheb3_bwd_dual_kernel<[^>]*>|cheb3_hop2mix_kernel<[^>]*>|spmm2_fused_kernel<[^>]*>|spmm1_dma_kernel<[^>]*>|spmm1_staged_kernel<[^>]*>|cheb3_fwd_fused_kernel<[^>]*>|cheb_wgrad_reduce)", r["Kernel_Name"])
        if not m:
            continue
        name = m.group(1)
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%3d avg=%.4g min=%.4g max=%.4g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
PY
