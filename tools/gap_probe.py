"""Where do the ~20 us between the per-call kernel time and the per-step wall time go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
case = bench.build_case(2500, 2000, 0, 1, "tiled", "semi_implicit")
from rdycore_amd import cases as CS
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")
for _ in range(20): op.rhs_function(case.dt, u, f)
torch.cuda.synchronize()
def wall(n, with_events):
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if with_events:
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
        op.rhs_function(case.dt, u, f)
        if with_events:
            e.record(); evs.append((s, e))
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n * 1e3
    ev = np.mean([s.elapsed_time(e) for s, e in evs]) if evs else None
    return t, ev
for rep in range(2):
    print("no events  : wall/step %.4f ms" % wall(200, False)[0])
    w, ev = wall(200, True)
    print("with events: wall/step %.4f ms, mean event time %.4f ms" % (w, ev))
# host-side cost of one call
t0 = time.perf_counter()
for i in range(200): op.rhs_function(case.dt, u, f)
print("host enqueue per call %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6)); torch.cuda.synchronize()
# graph replay of 10 steps
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    op.rhs_function(case.dt, u, f)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10): op.rhs_function(case.dt, u, f)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print("graph replay: wall/step %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
