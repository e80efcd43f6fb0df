"""GPU: bench.py prints ONE JSON line with the keys the driver's contract names (small mesh, a few steps)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--second-order"], ["--hr"], ["--workload", "dambreak_quads", "--nx", "160", "--ny", "80", "--cpu-sample", "80x40"],
                                   ["--workload", "c5", "--nx", "100", "--ny", "100", "--cpu-sample", "50x50", "--emulate-world", "4", "--emulate-rank", "1"],
                                   ["--emulate-world", "3", "--emulate-rank", "1", "--self-exchange"],
                                   ["--workload", "houston_refined", "--levels", "2", "--cpu-sample", "1x0"],
                                   ["--workload", "delaunay", "--nx", "90", "--cpu-sample", "40x40"]])
def test_bench_line_has_the_contract_keys(extra, rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("one kernel variant is enough")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--nx", "120", "--ny", "90",
           "--cpu-sample", "60x40", "--no-cpu-all-cores", "--condition-seconds", "0.1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("M cell-updates/s") and d["unit"] == "M cell-updates/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    strong = any(w in extra for w in ("dambreak_quads", "c5", "houston_refined", "delaunay"))
    assert d["higher_is_better"] is True and d["scaling"] == ("strong" if strong else "weak") and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "untimed RHS launches" in d["config"]["conditioning"] and d["config"]["world_size"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["finite"] is True
    if "houston_refined" in extra:      # the unstructured real-DEM workload: 2 746 x 4^2 triangles, wet / dry fronts, Hilbert order
        assert d["config"]["cells_per_gpu"] == 2746 * 16 and d["config"]["cell_order"] == "hilbert" and "Houston1km" in d["config"]["workload"]
        assert d["roofline"]["tile_edge_records_per_cell"] < 1.72 and d["roofline"]["halo_cells_per_tile"] < 70
    if "--self-exchange" in extra:      # the multi-rank step on one GPU: exchange looped back through a one-rank RCCL communicator
        assert "one-rank RCCL communicator" in d["config"]["partition"] and d["config"]["cells_per_gpu"] == 2 * 120 * 90
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and r["achieved"] > 0 and r["steady_state_period_median_ms"] > 0
    assert r["algorithmic_bytes_per_cell"] == (192.0 if "dambreak_quads" in extra else 176.0)
    if not extra:
        assert r["traffic_source"]["kernel_source_sha"] and set(d["cell_order_study"]) >= {"tiled", "rowmajor", "hilbert"}
    e = d["euler_step"]     # the product's step beside the metric: same byte model, PMC traffic under the same guard (null at these sizes)
    assert e["fused_ms_per_step"] > 0 and abs(e["frac_of_hbm_roofline"] - r["algorithmic_bytes_per_launch"] / (e["fused_ms_per_step"] * 1e-3) / 8e12) < 1e-3
    if "--self-exchange" not in extra:
        assert "traffic" in e and e["traffic_source"]["key"].endswith("_euler_step")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "M cell-updates/s" and "sample" in c
    # the CPU figure is taken on the benchmark mesh itself (the sample of earlier rounds is the second key); a rank's part of an
    # emulated partition keeps the sample only
    if "--emulate-world" in extra:
        assert "sample mesh" in c["sample"] and "cpu_baseline_sample" not in d
    else:
        assert "the benchmark mesh itself" in c["sample"] and str(d["config"]["cells_per_gpu"]) in c["sample"]
        assert d["cpu_baseline_sample"]["cores"] == 1 and "sample mesh" in d["cpu_baseline_sample"]["sample"]
    # the drop-in's real loop beside the back-to-back rate: 20 steps per RDyAdvance, rain refreshed from a host array, the Courant
    # struct read back (or not: fixed dt)
    if "--self-exchange" not in extra:
        a = d["advance_pattern"]
        assert a["steps_per_advance"] == 20 and a["ms_per_step_adaptive_dt"] > 0 and a["ms_per_step_fixed_dt"] > 0
        assert abs(a["vs_back_to_back_adaptive_dt"] - a["ms_per_step_adaptive_dt"] / a["back_to_back_ms_per_step"]) < 1e-2
    else:
        f1 = d["config"]["per_rank"][0]["rhs_step_form"]
        assert f1["form"] in ("in_order", "two_streams") and f1["source"] in ("measured", "trial_running", "forced")
    assert d["value"] > 0 and abs(d["value"] - d["config"]["cells_per_gpu"] / d["ms_per_step"] / 1e3) <= 1e-3 * d["value"] + 0.11
