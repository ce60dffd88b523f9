#!/usr/bin/env python3
"""Regenerate the measurement / parity tables of DESIGN.md from the committed profiles (prose and data must
not drift).  usage: tools/make_tables.py <round tag, e.g. r03> [--write]

Reads profiles/<tag>_bench_*.json (bench.py lines), profiles/<tag>_*_kernel_stats.csv (rocprofv3 --kernel-trace --stats of
the same commands), profiles/spmm_traffic.json (PMC passes: 2 x FETCH_SIZE + WRITE_SIZE) and
profiles/<tag>_parity_fullsize.json (tests/test_hip_fullsize.py); prints markdown, and with --write replaces the text
between the GENERATED markers of DESIGN.md."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
HBM = 8000.0


def load(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else None


def stats(name):
    path = os.path.join(P, name)
    rows = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = re.sub(r"\(.*", "", n)
            rows[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    return rows


out = []
w = out.append
w("| workload (`profiles/%s_bench_*.json`) | ms / step | nodes·channels/s | SpMM recurrences / 8 TB/s | forward recurrence alone (gate 0.60) | per-step HIP events (median, p10-p90) | CPU baseline (oracle port, same box) |" % tag)
w("|---|---|---|---|---|---|---|")
names = {"ns_default": "**NS** (default command): nside 64, K 3, 32->64, B 16, fp32, k = 8", "ns_k20": "NS shape, k = 20 stencil (the reference's default graph)",
         "c3": "C3: K 5, 64->128, bf16 (configs[2])", "unet": "C2: UNetSpherical nside 32, B 8, k = 20 (configs[1])",
         "c5": "C5: equiangular 200 x 400 + cross-sampling pooling (configs[4])"}
lines = {}
for key, label in names.items():
    d = load("%s_bench_%s.json" % (tag, key))
    if d is None:
        continue
    lines[key] = d
    r = d.get("roofline") or {}
    ps = d.get("per_step_us") or {}
    cb = d.get("cpu_baseline") or {}
    w("| %s | **%.4f** | %.3g | %s | %s | %s | %s |" % (
        label, d["ms_per_step"], d["value"], r.get("frac", "-"), r.get("fwd_recurrence_frac", "-"),
        "%s us (%s-%s)" % (ps.get("median"), ps.get("p10"), ps.get("p90")) if ps else "-",
        "%.2g /s on %s threads of %s" % (cb["value"], cb.get("cores"), cb.get("host_cpus")) if cb else "-"))
w("")
w("In-step kernels of the default command (`roofline.in_step` of the bench line: HIP events on the launch stream; "
  "algorithmic bytes = SURVEY 8d pass counts) next to rocprofv3 of the same command (`profiles/%s_default_kernel_stats.csv`) "
  "and the PMC traffic (`profiles/spmm_traffic.json`, `%s_ns_pmc_summary.txt`):" % (tag, tag))
w("")
w("| role | bench leg us | algorithmic MB -> fraction of 8 TB/s | rocprofv3 kernel (avg us, % of GPU time) | measured HBM MB (read + write) -> TB/s |")
w("|---|---|---|---|---|")
ns = lines.get("ns_default")
st = stats("%s_default_kernel_stats.csv" % tag)
tr = (load("spmm_traffic.json") or {}).get("ns", {}).get("kernels", {})
match = {"forward": "cheb3_fwd_fused_kernel", "backward GEMM": "cheb_wgrad_x3_kernel", "adjoint": "spmm2_fused_kernel<false, 3, true, true"}
if ns and ns.get("roofline", {}).get("in_step"):
    for e in ns["roofline"]["in_step"]:
        pat = next((v for k, v in match.items() if e["role"].startswith(k)), None)
        kn = next((n for n in st if pat and n.startswith(pat)), None)
        tk = next((n for n in tr if pat and n.startswith(pat)), None)
        prof = "`%s` %.1f us, %.1f %%" % (kn[:48], st[kn][1], st[kn][2]) if kn else "-"
        hbm = "-"
        if tk and kn:
            b = tr[tk]["read"] + tr[tk]["write"]
            hbm = "%.0f + %.0f = %.0f -> %.2f" % (tr[tk]["read"] / 1e6, tr[tk]["write"] / 1e6, b / 1e6, b / st[kn][1] / 1e6)
        w("| %s | %.1f | %.0f -> %.2f | %s | %s |" % (e["role"], e["avg_us"], e["algorithmic_bytes"] / 1e6, e["frac"], prof, hbm))
w("")
for key, fn in (("ns_k20", "k20"), ("c3", "c3"), ("unet", "unet"), ("c5", "c5")):
    st = stats("%s_%s_kernel_stats.csv" % (tag, fn))
    if not st:
        continue
    top = sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]
    w("%s (`profiles/%s_%s_kernel_stats.csv`), top kernels by total time: " % (key, tag, fn) + "; ".join(
        "`%s` %d x %.1f us (%.1f %%)" % (n[:44], c, a, p) for n, (c, a, p) in top) + ".")
    w("")
par = load("%s_parity_fullsize.json" % tag)
if par:
    w("Full-size parity (`tests/test_hip_fullsize.py` -> `profiles/%s_parity_fullsize.json`; max-rel errors normalised by max|ref|):" % tag)
    w("")
    w("| case | measured |")
    w("|---|---|")
    for k, v in par.items():
        w("| %s | %s |" % (k, ", ".join("%s %.2g" % (a, b) for a, b in v.items() if isinstance(b, (int, float)))))
text = "\n".join(out)
print(text)
if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, b = "<!-- BEGIN GENERATED %s -->" % tag, "<!-- END GENERATED %s -->" % tag
    if a in s and b in s:
        s = s[:s.index(a) + len(a)] + "\n" + text + "\n" + s[s.index(b):]
        open(path, "w").write(s)
        print("\n[make_tables] DESIGN.md updated", file=sys.stderr)
    else:
        print("\n[make_tables] markers not found in DESIGN.md", file=sys.stderr)
