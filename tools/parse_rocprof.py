"""Summarises the rocprofv3 output of tools/profile_gpu.sh into gpurun_out/prof_<tag>/summary.json + .md"""
import csv, glob, json, os, sys, collections

out, tag = sys.argv[1], sys.argv[2]

def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None

summary = {"tag": tag}
# ---- kernel stats
ks = find("trace", "*kernel_stats.csv")
rows = []
if ks:
    with open(ks) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
    summary["kernel_stats"] = [{k: r[k] for k in r} for r in rows[:12]]

def pmc_per_kernel(sub, counter):
    p = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if not p:
        return acc
    with open(p) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

def mean(x):
    return sum(x) / len(x) if x else None

pm = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
    acc = pmc_per_kernel(sub, ctr)
    pm[sub] = {k: {"mean": mean(v), "n": len(v)} for k, v in acc.items() if ("swe_rhs" in k or "axpy_owned" in k or "muscl_gradient" in k)}
summary["pmc_raw_KB"] = pm
# the SQ counters of the timed RHS kernel and of its Euler-step instantiation (template argument EULER = true; launched about as
# often by the bench's euler_step / advance_pattern extras) are kept apart: "sq" is the RHS kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdycore_amd.codeobj import is_euler_step_kernel  # noqa: E402
sq, sq_euler = {}, {}
for sub, ctrs in (("pmc_sq", ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")),
                  ("pmc_sq2", ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"))):
    for ctr in ctrs:
        acc = pmc_per_kernel(sub, ctr)
        for k, v in acc.items():
            if "swe_rhs" in k:
                (sq_euler if is_euler_step_kernel(k) else sq)[ctr] = mean(v)
summary["sq_euler_step"] = sq_euler
summary["sq"] = sq
with open(os.path.join(out, "summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
summary.pop("kernel_stats", None) if len(sys.argv) > 3 else None
print(json.dumps({k: summary[k] for k in ("pmc_raw_KB", "sq")}, indent=1))
if ks:
    for r in rows[:4]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
