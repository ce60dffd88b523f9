#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh into profiles/spmm_traffic.json.

usage: tools/make_traffic_json.py <workload key> <tag> [<workload key> <tag> ...]
Reads gpurun_out/pmc_<tag>_{A,B}/**/*counter_collection.csv.  HBM bytes per dispatch of the SpMM recurrence
kernels = (2 * FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE reports half of wide (16 B/lane) coalesced reads on gfx950
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.  The per-launch figure bench.py reports next to the
algorithmic bytes is the median dispatch per kernel, mean over the SpMM launches of one step (forward recurrence + adjoint recurrence)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "spmm_traffic.json")
result = json.load(open(out_path)) if os.path.exists(out_path) else {}
if "hbm_bytes_per_launch" in result:      # old single-workload layout
    result = {}
args = sys.argv[1:]
for key, tag in zip(args[0::2], args[1::2]):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{ROOT}/gpurun_out/pmc_{tag}_[AB]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(spmm2_fused_kernel<[^>]*>|spmm_csr_rowsplit<[^>]*>|cheb3_fwd_fused_kernel<[^>]*>|cheb_wgrad_x3_kernel<[^>]*>|ts_gemm_x3_kernel<[^>]*>)", r["Kernel_Name"])
            if m and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels, per_kernel = {}, {}
    for name, d in sorted(vals.items()):
        if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
            continue
        # MEDIAN dispatch: in whole-model runs one kernel variant serves layers of different sizes; the profiled command is
        # arranged so that the launches of interest (the roofline leg's layer) are the majority of its dispatches
        med = lambda v: sorted(v)[len(v) // 2]
        rd = 2.0 * 1024 * med(d["FETCH_SIZE"])
        wr = 1024.0 * med(d["WRITE_SIZE"])
        kernels[name] = {"read": round(rd), "write": round(wr), "dispatches_sampled": len(d["WRITE_SIZE"])}
        if name.startswith("spmm"):          # other kernels of the step are listed, not part of the SpMM mean
            per_kernel[name] = rd + wr
    # launches of ONE step's recurrences (what bench.py's roofline leg times): every forward variant once (first pair
    # without epilogue operands, later pairs with Z1), the adjoint variant (Z1 and Z2) as often as there are forward
    # launches - independent of how many dispatches of each the profiled command happened to contain
    is_adj = lambda n: bool(re.search(r"<(true|false), \d+, true, true", n))
    fwd = [n for n in per_kernel if not is_adj(n)]
    adj = [n for n in per_kernel if is_adj(n)]
    total, launches = 0.0, 0
    for n in fwd:
        total += per_kernel[n]; launches += 1
    for n in adj:
        w = max(1, len(fwd)) / max(1, len(adj))
        total += per_kernel[n] * w; launches += w
    if launches:
        result[key] = {"hbm_bytes_per_launch": round(total / launches), "kernels": kernels,
                       "source": f"tools/prof_pmc.sh {tag} (passes A: FETCH_SIZE x2, B: WRITE_SIZE), median dispatch per kernel, mean over the SpMM launches of one step"}
json.dump(result, open(out_path, "w"), indent=1)
print(json.dumps(result, indent=1))
