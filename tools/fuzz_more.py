import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_fuzz as T
from rdycore_amd.operator import RDyFlowConfig
from test_gpu_parity import run_both, check_all
bad = 0; worst = 0.0
for seed in range(100, 180):
    for variant in ("first", "second_minmod", "second_vanleer", "second_none", "hr"):
        rng = np.random.default_rng(seed)
        cfg = RDyFlowConfig(tiny_h=float(rng.choice([1e-7, 1e-5])), h_anuga_regular=float(rng.choice([0.0, 0.0, 1e-3])), source_method=int(seed % 2))
        if variant.startswith("second"):
            cfg.second_order = True; cfg.limiter = {"second_minmod": 0, "second_none": 1, "second_vanleer": 2}[variant]
        if variant == "hr": cfg.well_balancing = 2
        mesh = T.random_tri_mesh(rng, int(rng.integers(5, 40)), int(rng.integers(5, 30)), project_2d=(variant == "hr"))
        case = T.random_case(rng, mesh, cfg)
        try:
            f, fr, op, orc = run_both(case)
            err = check_all(case, f, fr, op, orc); worst = max(worst, err)
        except AssertionError as e:
            bad += 1; print("FAIL", seed, variant, str(e)[:200], flush=True)
print("done: failures", bad, "worst rel L-inf", worst)
