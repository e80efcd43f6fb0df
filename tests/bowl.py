"""Thacker's planar free surface oscillating in a parabolic bowl (frictionless): an analytic solution of the shallow-
water equations with a moving shoreline.  Bed z = h0 r^2/a^2 - h0 about the bowl's centre, surface
eta = (s h0 / a^2)(2 X cos wt + 2 Y sin wt - s), velocity (-s w sin wt, s w cos wt), w = sqrt(2 g h0)/a, period 2 pi / w.
Run with hydrostatic reconstruction, as the reference runs its own parabolic bowl (driver/tests/swe_roe/parabolic_bowl.yaml:
well_balancing: hydrostatic_reconstruction); without it the first-order scheme does not survive the steep dry bed."""
import numpy as np

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import RDyFlowConfig

G = 9.806
H0, A, S, C = 10.0, 3000.0, 750.0, 4000.0
OMEGA = np.sqrt(2 * G * H0) / A
PERIOD = 2 * np.pi / OMEGA


def bed(x, y):
    return H0 * ((x - C) ** 2 + (y - C) ** 2) / A ** 2 - H0


def surface(x, y, t):
    return (S * H0 / A ** 2) * (2 * (x - C) * np.cos(OMEGA * t) + 2 * (y - C) * np.sin(OMEGA * t) - S)


def case_and_steps(n):
    d = 2 * C / n
    mesh = M.structured_tri_mesh(n, n, d, zfunc=bed, project_2d=True, order="tiled", tile=8)
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    h = np.maximum(0.0, surface(xc, yc, 0.0) - mesh.cell_zc)
    u = np.stack([h, h * 0.0, h * S * OMEGA], axis=1)          # velocity (0, s w) at t = 0
    nsteps = int(np.ceil(PERIOD / (0.25 * d / (np.sqrt(G * H0) + S * OMEGA))))
    case = CS.Case("bowl", mesh, RDyFlowConfig(well_balancing=2), [M.CONDITION_REFLECTING] * len(mesh.boundaries), u,
                   np.zeros(mesh.num_cells), np.zeros((mesh.num_cells, 3)), {}, PERIOD / nsteps)
    return case, nsteps


def error_after_one_period(case, u):
    mesh = case.mesh
    xc, yc = mesh.cell_centroids[:, 0], mesh.cell_centroids[:, 1]
    he = np.maximum(0.0, surface(xc, yc, PERIOD) - mesh.cell_zc)
    a = mesh.cell_areas
    return float((np.abs(u[:, 0] - he) * a).sum() / (he * a).sum()), float((u[:, 0] * a).sum() / (case.u_local[:, 0] * a).sum())
