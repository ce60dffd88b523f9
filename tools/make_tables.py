#!/usr/bin/env python3
"""Regenerate the measurement / parity tables of DESIGN.md from the committed profiles (prose and data must not drift).
usage: tools/make_tables.py <round tag, e.g. r04> [--write]

Reads profiles/<tag>_bench_*.json (bench.py lines, each carrying the counter bytes of profiles/spmm_traffic.json),
profiles/<tag>_*_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same commands) and
profiles/<tag>_parity_fullsize.json (tests/test_hip_fullsize.py); prints markdown, and with --write replaces the text
between the GENERATED markers of DESIGN.md."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r06"


def load(name):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except ValueError:
        return json.loads(txt.splitlines()[-1])


def stats(name):
    path = os.path.join(P, name)
    rows = []
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("dsw_gemm::", "")
            n = re.sub(r"\(.*", "", n)
            rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return rows


def f(v, fmt="%.3f"):
    return "-" if v is None else fmt % v


out = []
w = out.append
names = {"ns_default": "**NS** (default command): nside 64, K 3, 32->64, B 16, fp32, k = 8",
         "ns_k20": "NS shape, k = 20 stencil (the reference's default graph)",
         "c3": "C3: K 5, 64->128, bf16 (configs[2])", "unet": "C2: UNetSpherical nside 32, B 8, k = 20 (configs[1])",
         "c5": "C5: equiangular 200 x 400 + cross-sampling pooling (configs[4])"}
detail_key = {"ns_default": "ns", "ns_k20": "ns_k20", "c3": "c3", "unet": "unet", "c5": "c5"}
w("| workload (`profiles/%s_bench_*.json`, detail in `%s_bench_detail_*.json`) | ms / step | nodes·channels/s | headline `roofline`: dominant HBM-side launch, 8(d) recurrence bytes / in-graph time / 8 TB/s (counter bytes) | its GEMM flops / 157.3 TF | whole step: 8(d) bytes / 8 TB/s; flops / 157.3 TF | forward recurrence (gate 0.60; counter) | adjoint recurrence (counter) | per-step HIP events (median, p10-p90) | CPU baseline (oracle port, same box) |" % (tag, tag))
w("|---|---|---|---|---|---|---|---|---|---|")
lines, details = {}, {}
for key, label in names.items():
    d = load("%s_bench_%s.json" % (tag, key))
    if d is None:
        continue
    lines[key] = d
    det = load("%s_bench_detail_%s.json" % (tag, detail_key[key])) or {}
    details[key] = det
    r = d.get("roofline") or {}
    fr, ar = det.get("forward_recurrence") or {}, det.get("adjoint_recurrence") or {}
    ps = d.get("per_step_us") or {}
    cb = d.get("cpu_baseline") or {}
    w("| %s | **%.4f** | %.3g | `%s` %.1f us: %s (%s) | %s | %s; %s | %s (%s)%s | %s (%s)%s | %s | %s |" % (
        label, d["ms_per_step"], d["value"], (r.get("kernel") or "-").split(" [")[0], r.get("avg_launch_us") or float("nan"),
        f(r.get("frac")), f(r.get("frac_counter")), f(r.get("mfma_frac")), f(r.get("step_frac")), f(r.get("step_mfma_frac")),
        f(fr.get("frac")), f(fr.get("frac_counter")), "" if fr.get("in_step") else ", leg only",
        f(ar.get("frac")), f(ar.get("frac_counter")), "" if (not ar or ar.get("in_step")) else ", leg only",
        "%s us (%s-%s)" % (ps.get("median"), ps.get("p10"), ps.get("p90")) if ps else "-",
        "%.2g /s on %s threads of %s" % (cb["value"], cb.get("cores"), cb.get("host_cpus")) if cb else "-"))
w("")
w("`frac` = SURVEY 8(d) RECURRENCE bytes of ONE launch / its IN-GRAPH duration / 8 TB/s (durations: rocprofv3 kernel trace of the replayed "
  "step graph, folded per role by `bench.py`); GEMM work is priced in flops against the fp32 matrix peak (157.3 TFLOP/s), never as bytes of "
  "launches an unfused design would have made; in brackets the same time against the HBM bytes the counters saw for exactly these launches "
  "(`profiles/spmm_traffic.json`, `tools/pmc_traffic.sh`: 2 x FETCH_SIZE + WRITE_SIZE).  `leg only`: the step does not launch that recurrence "
  "on its own (it is inside a fused launch); the leg is timed isolated.")
w("")
for key in names:
    d = lines.get(key)
    ins = ((d or {}).get("roofline") or {}).get("in_step")
    if not ins:
        continue
    r = d["roofline"]
    w("In-step launches of %s (`roofline.in_step`; their sum is %.3f of `ms_per_step`):" % (key, r.get("in_step_sum_vs_ms_per_step") or float("nan")))
    w("")
    w("| role | kernel | in-graph us | share | 8(d) recurrence MB -> of 8 TB/s | GEMM GFLOP -> of the matrix peak (157.3 TF fp32; 2.5 PF for bf16 operands) | compulsory MB -> of 8 TB/s | counter MB -> of 8 TB/s |")
    w("|---|---|---|---|---|---|---|---|")
    for e in ins:
        sec = e["us"] * 1e-6
        b8, fl, cb_, mv = e.get("bytes_8d"), e.get("flops"), e.get("compulsory_bytes"), e.get("bytes_moved")
        w("| %s | `%s` | %.1f | %.2f | %s | %s | %s | %s |" % (
            e["role"], e["kernel"], e["us"], e.get("share") or float("nan"),
            "%.0f -> %.2f" % (b8 / 1e6, b8 / sec / 8e12) if b8 else "-",
            "%.2f -> %s" % (fl / 1e9, f(e.get("mfma_frac"), "%.2f")) if fl else "-",
            "%.0f -> %.2f" % (cb_ / 1e6, cb_ / sec / 8e12) if cb_ else "-",
            "%.0f -> %.2f" % (mv / 1e6, mv / sec / 8e12) if mv else "-"))
    w("")
for key, fn in (("ns_default", "default"), ("ns_k20", "k20"), ("c3", "c3"), ("unet", "unet"), ("c5", "c5")):
    rows = stats("%s_%s_kernel_stats.csv" % (tag, fn))
    if not rows:
        continue
    top = sorted(rows, key=lambda r_: -r_[1] * r_[2])[:7]
    w("%s (`profiles/%s_%s_kernel_stats.csv`, rocprofv3 `--kernel-trace --stats` of the same command), top kernels by total time: %s." % (
        key, tag, fn, "; ".join("`%s` %d x %.1f us (%.1f %%)" % (n[:46], c, us, pct) for n, c, us, pct in top)))
    w("")
for key in ("unet",):
    ts = (details.get(key) or {}).get("traced_step")
    if not ts:
        continue
    g = ts.get("graph") or {}
    w("Roles of one %s step (`traced_step` of the detail file: order and roles from the library's launch trace, durations in-graph; library "
      "kernels %.0f us + other kernels = %.0f us per step for %.0f us measured):" % (
          key, ts["sum_us"], g.get("all_kernels_us_per_step", float("nan")), lines[key]["ms_per_step"] * 1e3))
    w("")
    w("| role | shape (V / rows, Fin / C, Fout / K) | calls / step | avg us | us / step |")
    w("|---|---|---|---|---|")
    for e in ts["roles"][:12]:
        w("| %s | %s | %.0f | %.1f | %.1f |" % (e["role"], " x ".join(str(a) for a in e["aux"]), e["calls_per_step"], e["avg_us"], e["us_per_step"]))
    oth = g.get("other_kernels") or []
    if oth:
        w("")
        w("Kernels of that step that are not the library's (torch glue): " + "; ".join(
            "`%s` %.1f x %.1f us" % (o["kernel"], o["calls_per_step"], o["us_per_step"] / max(o["calls_per_step"], 1e-9)) for o in oth[:6]) + ".")
    w("")
for key in ("unet", "c5"):
    po = (details.get(key) or {}).get("pooling")
    if not po:
        continue
    w("Pooling products of %s (`pooling` of the detail file; bytes = input rows + output rows + operator):" % key)
    w("")
    w("| layer | product | rows out x in | channels | entries / row | us | fraction of 8 TB/s |")
    w("|---|---|---|---|---|---|---|")
    for e in po["products"]:
        w("| %s | %s | %d x %d | %d | %s | %.1f | %.2f |" % (e["layer"], e["product"], e["rows_out"], e["rows_in"], e["channels"],
                                                            e["nnz_per_row"], e["us"], e["frac"]))
    w("")
par = load("%s_parity_fullsize.json" % tag)
if par:
    w("Full-size parity (`tests/test_hip_fullsize.py` -> `profiles/%s_parity_fullsize.json`; max-rel errors normalised by max|ref|):" % tag)
    w("")
    w("| case | measured |")
    w("|---|---|")
    for k in sorted(par):
        v = par[k]
        w("| %s | %s |" % (k, ", ".join("%s %.2g" % (kk, vv) for kk, vv in sorted(v.items()) if isinstance(vv, (int, float)))))
text = "\n".join(out)
print(text)
if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    a, b = "<!-- BEGIN GENERATED %s -->" % tag, "<!-- END GENERATED %s -->" % tag
    if a in s and b in s:
        s = s[:s.index(a) + len(a)] + "\n" + text + "\n" + s[s.index(b):]
        open(path, "w").write(s)
        print("\n[DESIGN.md updated]", file=sys.stderr)
    else:
        print("\n[markers %s / %s not found in DESIGN.md]" % (a, b), file=sys.stderr)
