#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round tag, e.g. r02>
# Everything the DESIGN.md tables quote, into gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/):
#   bench lines of every workload, rocprofv3 kernel stats of the default bench command / unet / c3 / k=20,
#   PMC passes (HBM traffic, MFMA busy, LDS) of the ns / c3 / k=20 workloads.
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
: > $out/bench.err
cd $root
bash tools/prof_stats.sh ${tag}_default > $out/stats_default.txt 2>&1
bash tools/prof_stats.sh ${tag}_unet --workload unet --no-cpu-baseline --no-roofline --steps 10 > $out/stats_unet.txt 2>&1
bash tools/prof_stats.sh ${tag}_c3 --workload c3 --no-cpu-baseline --steps 20 > $out/stats_c3.txt 2>&1
bash tools/prof_stats.sh ${tag}_k20 --knn 20 --no-cpu-baseline --steps 20 > $out/stats_k20.txt 2>&1
for t in default unet c3 k20; do cp $root/gpurun_out/prof_${tag}_$t/${tag}_${t}_kernel_stats.csv $out/${tag}_${t}_kernel_stats.csv 2>/dev/null; done
bash tools/prof_stats.sh ${tag}_c5 --workload c5 --no-cpu-baseline --steps 20 > $out/stats_c5.txt 2>&1
cp $root/gpurun_out/prof_${tag}_c5/${tag}_c5_kernel_stats.csv $out/${tag}_c5_kernel_stats.csv 2>/dev/null
# SQ / TCC counters of the whole step (what each kernel is bound by), per workload
BENCH_ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline" bash tools/prof_pmc.sh ${tag}_ns > $out/${tag}_ns_pmc_summary.txt 2>&1
BENCH_ARGS="--knn 20 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" bash tools/prof_pmc.sh ${tag}_k20 > $out/${tag}_k20_pmc_summary.txt 2>&1
BENCH_ARGS="--workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" bash tools/prof_pmc.sh ${tag}_c3 > $out/${tag}_c3_pmc_summary.txt 2>&1
BENCH_ARGS="--workload unet --steps 5 --warmup 2 --no-cpu-baseline --no-roofline" PMC_PASSES="A B C" bash tools/prof_pmc.sh ${tag}_unet > $out/${tag}_unet_pmc_summary.txt 2>&1
BENCH_ARGS="--workload c5 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline" PMC_PASSES="A B" bash tools/prof_pmc.sh ${tag}_c5 > $out/${tag}_c5_pmc_summary.txt 2>&1
# HBM bytes of exactly the launches the roofline entries time (bench.py --pmc-leg) -> profiles/spmm_traffic.json
bash tools/pmc_traffic.sh ${tag} > $out/traffic.log 2>&1
cp profiles/spmm_traffic.json $out/spmm_traffic.json
# bench lines again, now WITH the counter bytes of this collection in them
for w in ns_default:"" ns_k20:"--knn 20" c3:"--workload c3" unet:"--workload unet" c5:"--workload c5"; do
  python bench.py ${w#*:} > $out/${tag}_bench_${w%%:*}.json 2>> $out/bench.err
done
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_ns_driver_style.json 2>> $out/bench.err
ls -la $out
# what travels back is limited (64 MiB): the raw traces have been summarised above
find $root/gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
find $root/gpurun_out -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $root/gpurun_out
