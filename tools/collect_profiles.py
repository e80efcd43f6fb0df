"""Copies what tools/profile_all.sh left under gpurun_out/prof_<round>_<tag>/ into profiles/ (kernel stats, the parsed PMC
summary, the bench line printed under the tracer) and writes the variant's entry of profiles/traffic.json.
usage: python tools/collect_profiles.py <round> [variant ...]"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ARGS = {
    "c3": [], "hr": ["--hr"], "xq": ["--source", "implicit_xq2018"], "c2": ["--workload", "c2"], "quads": ["--workload", "dambreak_quads"],
    "c5": ["--workload", "c5", "--emulate-world", "8", "--emulate-rank", "3"], "so": ["--second-order"],
    "so_quads": ["--second-order", "--workload", "dambreak_quads"], "houston": ["--workload", "houston_refined"],
    "houston_hr": ["--workload", "houston_refined", "--hr"], "houston_so": ["--workload", "houston_refined", "--second-order"],
    "delaunay": ["--workload", "delaunay"], "self_exchange": ["--emulate-world", "3", "--emulate-rank", "1", "--self-exchange"],
    "self_exchange_so": ["--emulate-world", "3", "--emulate-rank", "1", "--self-exchange", "--second-order"],
    "houston_natural": ["--workload", "houston_refined", "--order", "natural"],
    "houston_l7": ["--workload", "houston_refined", "--levels", "7"],
    "houston_l7_so": ["--workload", "houston_refined", "--levels", "7", "--second-order"],
    "houston_l7_hr": ["--workload", "houston_refined", "--levels", "7", "--hr"],
}
rnd = sys.argv[1]
tags = sys.argv[2:] or list(ARGS)
for tag in tags:
    d = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{tag}")
    summ = os.path.join(d, "summary.json")
    if not os.path.exists(summ):
        print(tag, ": no summary.json (not profiled yet)")
        continue
    ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(ROOT, "profiles", f"{rnd}_{tag}_kernel_stats.csv"))
    shutil.copy(summ, os.path.join(ROOT, "profiles", f"{rnd}_{tag}_summary.json"))
    log = os.path.join(d, "bench_trace.log")
    line = None
    if os.path.exists(log):
        lines = [ln for ln in open(log) if ln.startswith("{")]
        if lines:
            open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_bench_under_rocprof.json"), "w").write(lines[-1])
            line = json.loads(lines[-1])
    key = bench.traffic_key(bench.parse(ARGS[tag]))
    # launches of the RHS kernel per STEP: two (interior + halo tiles) only where the traced run really used the overlapped form
    # of the multi-rank step (config.halo_overlapped); the in-order form of small parts is one launch.  Second order: the
    # ghost-adjacent cells' gradient launch belongs to the step as well.
    per_step, extra = "1", ""
    if tag.startswith("self_exchange"):
        if line is None:
            print(tag, ": no bench line under the tracer, cannot tell the form of the step -- skipped")
            continue
        # ... as the PMC passes ran it, not the traced run: with counters on, kernels are serialised, the two-stream form cannot
        # overlap anything and the halo's own trial picks the in-order form (one launch per step) whatever the traced run chose
        forms = set()
        for name in ("bench_pmc_fetch.log", "bench_pmc_write.log"):
            lp = os.path.join(d, name)
            for ln in (open(lp) if os.path.exists(lp) else []):
                if ln.startswith("{"):
                    forms.add(bool(json.loads(ln)["config"].get("halo_overlapped")))
        if len(forms) != 1:
            print(tag, ": the FETCH_SIZE and WRITE_SIZE passes did not run the same form of the step -- skipped")
            continue
        per_step = "2" if forms.pop() else "1"
        if "second" in line["config"].get("spatial_order", ""):
            extra = "muscl_gradient_kernel"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), summ, key, f"profiles/{rnd}_{tag}_summary.json", per_step, extra],
                       capture_output=True, text=True)
    print(tag, key, "ok" if r.returncode == 0 else ("FAILED: " + r.stderr[-300:]))
    if not tag.startswith("self_exchange") and tag != "c5":
        # the fused Euler step's kernel was launched by the same runs (bench.py's euler_step extra): its entry, same guard
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), summ, key + "_euler_step", f"profiles/{rnd}_{tag}_summary.json", "1", ""],
                           capture_output=True, text=True, env=dict(os.environ, MAKE_TRAFFIC_EULER="1"))
        print(tag, key + "_euler_step", "ok" if r.returncode == 0 else ("FAILED: " + r.stderr[-300:]))
