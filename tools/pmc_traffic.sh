#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <tag> [workloads...]   -> profiles/spmm_traffic.json
# HBM bytes of exactly the launches bench.py's roofline entry times: for every workload and leg (forward recurrence,
# adjoint recurrence, pooling products) `bench.py --pmc-leg <leg>` runs ONLY that leg (20 calls), once per counter pass
# (rocprofv3 --kernel-trace --pmc, FETCH_SIZE and WRITE_SIZE in separate runs as MI355X_MICROARCH.md prescribes).
tag=${1:-rXX}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
wls=${@:-ns ns_k20 c3 unet c5}
cd /tmp && export TMPDIR=/tmp
for wl in $wls; do
  case $wl in
    ns) wargs="";; ns_k20) wargs="--knn 20";; c3) wargs="--workload c3";; unet) wargs="--workload unet";; c5) wargs="--workload c5";;
  esac
  legs="fwd adj"; if [ $wl = unet ] || [ $wl = c5 ]; then legs="fwd adj pool"; fi
  if [ $wl = ns ]; then legs="fwd adj bwdd fwd1l"; fi     # (the step's backward / forward are ONE launch each there)
  for leg in $legs; do
    for pass in "A FETCH_SIZE" "B WRITE_SIZE"; do
      set -- $pass; p=$1; shift
      out=$root/gpurun_out/pmct_${tag}_${wl}_${leg}_$p
      mkdir -p $out
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d $out -o pmc -- python $root/bench.py $wargs --pmc-leg $leg > $out/run.log 2>&1
      echo "$wl $leg pass $p: rc=$?"
    done
  done
done
cd $root
python tools/make_traffic_json.py $tag $wls
find $root/gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
find $root/gpurun_out -path "*pmct_*" -name "*counter_collection.csv" -delete 2>/dev/null
