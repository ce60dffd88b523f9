#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/spmm_traffic.json.

usage: tools/make_traffic_json.py <tag> <workload key> [<workload key> ...]
Reads gpurun_out/pmct_<tag>_<workload>_<leg>_{A,B}/**/*counter_collection.csv.  Each of those runs executed ONE leg of
bench.py's roofline measurement (`--pmc-leg fwd | adj | pool`: 20 calls of the forward recurrence / the adjoint
recurrence / the pooling products), so every dispatch of a hand-written sparse kernel in it belongs to the leg:
    HBM bytes per call = sum over those dispatches of (2 * FETCH_SIZE + WRITE_SIZE) KiB / calls
(FETCH_SIZE reports half of wide coalesced reads on gfx950 - MI355X_MICROARCH.md, HBM section; WRITE_SIZE is exact)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "spmm_traffic.json")
CALLS = 20
LEGS = ("fwd", "adj", "pool", "bwdd", "fwd1l")
KERNEL = re.compile(r"(spmm1_dma_kernel<[^>]*>|spmm1_staged_kernel<[^>]*>|spmm2_fused_kernel<[^>]*>|spmm_csr_rowsplit<[^>]*>|remap_\w+_kernel<[^>]*>|spmm_long_rows\w*<[^>]*>|cheb3_bwd_dual_kernel<[^>]*>|cheb3_fwd_fused_kernel<[^>]*>|cheb_wgrad_reduce_kernel<[^>]*>)")
tag, keys = sys.argv[1], sys.argv[2:]
result = {}
if os.path.exists(out_path):
    old = json.load(open(out_path))
    result = {k: v for k, v in old.items() if isinstance(v, dict) and any(leg in v for leg in LEGS)}
for key in keys:
    entry = {}
    for leg in LEGS:
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{ROOT}/gpurun_out/pmct_{tag}_{key}_{leg}_[AB]/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = KERNEL.search(r["Kernel_Name"])
                if m and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        kernels, rd_tot, wr_tot, n_disp = {}, 0.0, 0.0, 0
        for name, d in sorted(vals.items()):
            if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
                continue
            rd, wr = 2.0 * 1024 * sum(d["FETCH_SIZE"]), 1024.0 * sum(d["WRITE_SIZE"])
            kernels[name] = {"read_per_call": round(rd / CALLS), "write_per_call": round(wr / CALLS),
                             "dispatches_per_call": round(len(d["WRITE_SIZE"]) / CALLS, 2)}
            rd_tot += rd; wr_tot += wr; n_disp += len(d["WRITE_SIZE"])
        if kernels:
            entry[leg] = {"hbm_bytes_per_call": round((rd_tot + wr_tot) / CALLS), "read_per_call": round(rd_tot / CALLS),
                          "write_per_call": round(wr_tot / CALLS), "launches_per_call": round(n_disp / CALLS, 2), "kernels": kernels}
    if entry:
        entry["source"] = f"tools/pmc_traffic.sh {tag}: rocprofv3 --kernel-trace --pmc, FETCH_SIZE (x2) and WRITE_SIZE in separate passes over `bench.py --pmc-leg`, {CALLS} calls per leg"
        result[key] = entry
json.dump(result, open(out_path, "w"), indent=1)
print(json.dumps({k: {leg: v[leg]["hbm_bytes_per_call"] for leg in v if leg != "source"} for k, v in result.items()}, indent=1))
