"""Does a HIP graph of one RHS evaluation (RHS kernel + Courant finalize) shorten the step? (exploratory)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rdycore_amd import cases as CS

case = bench.build_case(2500, 2000, 0, 1, "tiled", "semi_implicit", "c3")
op = CS.create_operator(case)
u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
f = torch.empty((case.mesh.num_owned_cells, 3), dtype=torch.float64, device="cuda")

def loop(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

plain = lambda: op.rhs_function(case.dt, u, f)
print("plain   ms/step", loop(plain), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    plain(); plain()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plain()
print("graph   ms/step", loop(g.replay), flush=True)
print("plain   ms/step", loop(plain), flush=True)
g10 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g10):
    for _ in range(10): plain()
print("graph10 ms/step", loop(g10.replay, 20) / 10, flush=True)
