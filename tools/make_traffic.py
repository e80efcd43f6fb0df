"""profiles/traffic.json entry from the PMC passes of tools/profile_gpu.sh.

usage: python tools/make_traffic.py gpurun_out/prof_<tag>/summary.json <key> [source note] [launches per step] [extra kernel]
  (launches per step = 2 for the --self-exchange step, whose interior and halo launches are two launches of ONE kernel: the
   entry then holds the bytes of a whole step = 2 x the mean over all launches; extra kernel: a kernel launched once per step
   beside the RHS kernel whose bytes belong to the step -- the second-order step's muscl_gradient_kernel)
  (the bench line printed during the traced run, gpurun_out/prof_<tag>/bench_trace.log, supplies the layout's byte count)
  key = <workload>_<nx>x<ny>_<order>_<source>[_hr]   (what bench.py looks up)

FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section) and both counters are
checked against the calibration kernel of the same session (rdyhip::axpy_owned_kernel on 10 M cells: 480 MB read,
240 MB written).  The entry carries the hash of the measured kernel's MACHINE CODE (rdycore_amd/codeobj.py), taken on the GPU box
from the library the profiled run loaded (gpurun_out/prof_<tag>/kernel_code_hashes.json); bench.py reports the figure only while
the library it runs has the same bytes for that kernel (a stale entry is reported as stale, never silently)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    summ, key = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else os.path.relpath(summ, ROOT)
    per_step = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] else 1
    extra = sys.argv[5] if len(sys.argv) > 5 else ""
    s = json.load(open(summ))
    raw = s["pmc_raw_KB"]

    # the Euler-step instantiations (template argument EULER = true) are launched by the bench's euler_step and advance_pattern
    # extras, about as often as the timed RHS kernel: told apart by their template arguments, not by their launch counts
    from rdycore_amd.codeobj import is_euler_step_kernel
    want_euler = bool(os.environ.get("MAKE_TRAFFIC_EULER"))

    def pick(sub, needle):
        best = (None, None, -1)
        for k, v in raw.get(sub, {}).items():
            ok = (needle in k and is_euler_step_kernel(k) == want_euler) if needle == "swe_rhs" else needle in k
            if ok and v["mean"] is not None and v.get("n", 0) > best[2]:
                best = (k, v["mean"], v.get("n", 0))
        return best[0], best[1]

    kname, fetch = pick("pmc_fetch", "swe_rhs")
    _, write = pick("pmc_write", "swe_rhs")
    _, cal_f = pick("cal_fetch", "axpy_owned")
    _, cal_w = pick("cal_write", "axpy_owned")
    if fetch is None or write is None:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE for the RHS kernel in " + summ)
    rd, wr = fetch * 1024 * 2 * per_step, write * 1024 * per_step
    extra_ent = None
    if extra:
        ek, ef = pick("pmc_fetch", extra)
        _, ew = pick("pmc_write", extra)
        if ef is None or ew is None:
            raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE for {extra} in " + summ)
        rd, wr = rd + ef * 1024 * 2, wr + ew * 1024
        extra_ent = {"kernel": ek, "FETCH_SIZE_KB_raw": ef, "WRITE_SIZE_KB_raw": ew, "launches_per_step": 1}
    second = "second_order" in key
    layout = None
    log = os.path.join(os.path.dirname(summ), "bench_trace.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{"):
                layout = json.loads(ln)["roofline"]["layout_bytes_per_launch"]
    hpath = os.path.join(os.path.dirname(summ), "kernel_code_hashes.json")
    if not os.path.exists(hpath):
        raise SystemExit(f"{hpath} is missing: the profiled run did not record the code it measured (tools/profile_gpu.sh)")
    hashes = json.load(open(hpath))
    if kname not in hashes or (extra_ent and extra_ent["kernel"] not in hashes):
        raise SystemExit(f"the measured kernel {kname!r} is not in {hpath}")
    if extra_ent:
        extra_ent["code_sha"] = hashes[extra_ent["kernel"]]
    ent = {"kernel": kname, "code_sha": hashes[kname], "source_sha_at_collection": bench.kernel_sha(second), "layout_bytes_per_launch": layout, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
           "source": note}
    if per_step != 1:
        ent["launches_per_step"] = per_step
    if extra_ent:
        ent["also_in_the_step"] = extra_ent
    if cal_f is not None and cal_w is not None:
        ent["calibration"] = {"kernel": "rdyhip::axpy_owned_kernel, 10 M cells: 480.0 MB read / 240.0 MB written by construction",
                              "FETCH_SIZE_x2_MB": round(cal_f * 1024 * 2 / 1e6, 1), "WRITE_SIZE_MB": round(cal_w * 1024 / 1e6, 1)}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path)) if os.path.exists(path) else {}
    t[key] = ent
    json.dump(t, open(path, "w"), indent=1)
    print(json.dumps({key: ent}, indent=1))


if __name__ == "__main__":
    main()
