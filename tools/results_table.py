"""Markdown rows of DESIGN.md's results tables from the bench lines and rocprof summaries under profiles/.
usage: python tools/results_table.py r03"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
order = ["c3", "c3_200", "hr", "xq", "c2", "quads", "c5", "so", "so_quads", "houston", "houston_natural", "houston_hr", "houston_so", "houston_l7", "houston_l7_hr", "houston_l7_so", "delaunay",
         "self_exchange", "self_exchange_so"]
print("| variant | kernel | ms / step (unprofiled) | M cell-updates/s | frac of 8 TB/s (model B/cell) | PMC traffic vs algorithmic | rocprofv3 kernel avg (launches) |")
print("|---|---|---|---|---|---|---|")
for tag in order:
    p = os.path.join(ROOT, "profiles", f"{rnd}_bench_{tag}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    r = d["roofline"]
    alg = r["algorithmic_bytes_per_launch"]
    tr = r.get("traffic")
    ks = os.path.join(ROOT, "profiles", f"{rnd}_{tag.replace('c3_200', 'c3')}_kernel_stats.csv")
    kavg = ""
    if os.path.exists(ks):
        rows = list(csv.DictReader(open(ks)))
        top = max((x for x in rows if "swe_rhs" in x["Name"]), key=lambda x: int(x["Calls"]), default=None)
        if top:
            kavg = f"{float(top['AverageNs']) / 1e3:.1f} µs ({top['Calls']})"
    frac = f"{r['frac']:.3f} ({int(r['algorithmic_bytes_per_cell'])})"
    if "second_order_model" in r:
        frac += f"; {r['second_order_model']['frac']:.3f} ({int(r['second_order_model']['bytes_per_cell_update'])})"
    if "frac_176B_model" in r:
        frac += f"; {r['frac_176B_model']:.3f} (176)"
    print(f"| {tag} | `{r['kernel']}` | {d['ms_per_step']:.4f} | {d['value']:.0f} | {frac} | "
          f"{(f'{tr / 1e9:.3f} GB = {tr / alg:.3f} ×') if tr else 'null: ' + str((r.get('traffic_source') or {}).get('status'))} | {kavg} |")

# the whole forward-Euler step (rdyhip_euler_step: the update fused into the kernel's stores, F not written), same byte model
rows = []
for tag in order:
    # (<round>_bench_<tag>_euler.json: lines taken after the Euler-step kernels had been profiled -- another call, another box)
    p = os.path.join(ROOT, "profiles", f"{rnd}_bench_{tag}_euler.json")
    if not os.path.exists(p):
        p = os.path.join(ROOT, "profiles", f"{rnd}_bench_{tag}.json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    e = d.get("euler_step") or {}
    if "frac_of_hbm_roofline" not in e:
        continue
    alg = d["roofline"]["algorithmic_bytes_per_launch"]
    tr = e.get("traffic")
    rows.append(f"| {tag} | {d['ms_per_step']:.4f} | {e['fused_ms_per_step']:.4f} | {e['rhs_plus_axpy_ms_per_step']:.4f} | {e['frac_of_hbm_roofline']:.3f} | "
                f"{(f'{tr / 1e9:.3f} GB = {tr / alg:.3f} ×') if tr else 'not profiled'} |")
if rows:
    print()
    print("| variant | RHS alone, ms (this box) | fused Euler step, ms | RHS + axpy pair, ms | step: frac of 8 TB/s (RHS byte model) | PMC traffic of the Euler-step kernel vs algorithmic |")
    print("|---|---|---|---|---|---|")
    print("\n".join(rows))
