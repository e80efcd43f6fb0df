"""Known answers for the branches the reference's own tests hold no values for, derived by hand from the scheme's
definition (not from the oracle's code): each entry builds a tiny Case and says what the right-hand side or a boundary
edge's flux must be, in closed form.  tests/test_known_answers_cpu.py asserts them against the oracle,
tests/test_gpu_known_answers.py against the HIP operator -- both against the same numbers.

Notation: g = 9.806 (src/swe/swe_types_petsc.h:7), n = (cn, sn) the outward unit normal of a boundary edge.

Roe flux facts used (src/swe/swe_roe_flux_petsc.h:15-132):
  * two identical states: the dissipation term vanishes, the flux is the physical flux
    F(U).n = (h un, h u un + g h^2/2 cn, h v un + g h^2/2 sn);
  * a state and its mirror image across the edge (the reflecting BC, src/swe/swe_petsc.c:434-461): uhat.n = 0, dh = 0, and
    the two acoustic waves carry equal and opposite mass, so F_h = 0; at rest the momentum flux is the hydrostatic
    pressure g h^2/2 n; with normal velocity un >= 0 it is (h un^2 + g h^2/2 + h un sqrt(g h)) n.
"""
import math

import numpy as np

from rdycore_amd import cases as CS
from rdycore_amd import mesh as M
from rdycore_amd.operator import (RDyFlowConfig, SOURCE_IMPLICIT_XQ2018, SOURCE_SEMI_IMPLICIT, WELL_BALANCING_HR)

G = 9.806


def one_quad(theta=0.0, z=(0.0, 0.0, 0.0, 0.0), size=1.0):
    """a single square cell of side `size`, rotated by theta about the origin; vertex elevations z"""
    c, s = math.cos(theta), math.sin(theta)
    base = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=float) * size
    xyz = np.zeros((4, 3))
    xyz[:, 0] = c * base[:, 0] - s * base[:, 1]
    xyz[:, 1] = s * base[:, 0] + c * base[:, 1]
    xyz[:, 2] = z
    return M.build_mesh(xyz, np.array([[0, 1, 2, 3]], dtype=np.int32), boundary_classifier=M.single_boundary())


def one_cell_case(mesh, state, ctype, bvalues=None, config=None, mannings=0.0, dt=0.1, src=(0.0, 0.0, 0.0)):
    u = np.tile(np.asarray(state, dtype=float), (mesh.num_cells, 1))
    no = mesh.num_owned_cells
    bv = {} if bvalues is None else {0: np.tile(np.asarray(bvalues, dtype=float), (mesh.boundaries[0].num_edges, 1))}
    return CS.Case("kat", mesh, config or RDyFlowConfig(), [ctype], u, np.full(no, float(mannings)), np.tile(np.asarray(src, dtype=float), (no, 1)), bv, dt)


def edge_normals(mesh):
    b = mesh.boundaries[0]
    return mesh.edge_cn[b.edge_ids], mesh.edge_sn[b.edge_ids], mesh.edge_lengths[b.edge_ids]


def physical_flux(h, u, v, cn, sn):
    un = u * cn + v * sn
    return np.stack([h * un, h * u * un + 0.5 * G * h * h * cn, h * v * un + 0.5 * G * h * h * sn], axis=-1)


# ---------------------------------------------------------------------------------------------------------------------
# each entry: name -> dict(case=Case, flux=expected boundary flux [edges,3] or None, rhs=expected F [cells,3] or None,
#                          exact=list of (kind, column) that must hold to the last bit, tol=relative tolerance otherwise)
# ---------------------------------------------------------------------------------------------------------------------
def entries():
    out = {}
    h = 2.5

    # 1. reflecting wall, water at rest, axis-aligned and oblique cells: flux = (0, g h^2/2 cn, g h^2/2 sn); the closed
    #    cell's RHS is the sum of those over its four sides = 0
    for name, theta in (("reflecting_rest_axis", 0.0), ("reflecting_rest_oblique", 0.37)):
        m = one_quad(theta)
        cn, sn, _ = edge_normals(m)
        fl = np.stack([np.zeros(4), 0.5 * G * h * h * cn, 0.5 * G * h * h * sn], axis=1)
        # Courant number (swe_petsc.c:589-596): amax len / area dt with amax = chat + |uhat.n| = sqrt(g h) against a mirror state
        out[name] = dict(case=one_cell_case(m, (h, 0.0, 0.0), M.CONDITION_REFLECTING), flux=fl, rhs=np.zeros((1, 3)),
                         exact=[("flux", 0)] if theta == 0.0 else [], tol=1e-14, courant=math.sqrt(G * h) * 0.1)

    # 2. reflecting wall, axis-aligned, flow along +x with speed u0: the walls with n = +-y see tangential flow only (pure
    #    pressure flux), the wall n = +x is hit head on (un = u0 >= 0: h u0^2 + g h^2/2 + h u0 sqrt(g h)), the wall n = -x
    #    is left behind (un = -u0 < 0; the entropy fix of swe_roe_flux_petsc.h:56-67 then replaces |lambda| = c by
    #    (c^2/d + d)/2 with d = 4 u0 if c < d).  Mass flux exactly zero on all four.
    u0 = 0.8
    m = one_quad(0.0)
    cn, sn, _ = edge_normals(m)
    c = math.sqrt(G * h)
    fl = np.zeros((4, 3))
    for e in range(4):
        un = u0 * cn[e]
        a = c
        if un < 0.0:
            d = -4.0 * un
            if c < d:
                a = 0.5 * (c * c / d + d)
        p = 0.5 * G * h * h
        fl[e] = [0.0, (h * un * un + p) * cn[e] + a * h * un * cn[e], p * sn[e]]
    #    Courant number: the mirror state has uhat.n = 0 on every wall, so amax = sqrt(g h) on all four (the entropy fix
    #    changes the dissipation, not amax)
    out["reflecting_normal_flow"] = dict(case=one_cell_case(m, (h, h * u0, 0.0), M.CONDITION_REFLECTING), flux=fl, rhs=None,
                                         exact=[("flux", 0)], tol=1e-14, courant=math.sqrt(G * h) * 0.1)

    # 3. Dirichlet boundary whose value is the cell's own state: identical states -> the physical flux; the closed sum over
    #    the cell vanishes, so F = 0
    st = (1.7, 1.7 * 0.4, 1.7 * -0.3)
    m = one_quad(0.21)
    cn, sn, _ = edge_normals(m)
    #    Courant number: identical states -> uhat = u, chat = sqrt(g h): the largest |u.n| + c over the four edges
    out["dirichlet_same_state"] = dict(case=one_cell_case(m, st, M.CONDITION_DIRICHLET, bvalues=st),
                                       flux=physical_flux(st[0], 0.4, -0.3, cn, sn), rhs=np.zeros((1, 3)), exact=[], tol=1e-14,
                                       courant=float(np.max(np.abs(0.4 * cn - 0.3 * sn)) + math.sqrt(G * st[0])) * 0.1)

    # 4. critical outflow (src/swe/swe_petsc.c:465-503): the outside state has the discharge of the inside one at Froude
    #    number 1: h_r = (q^2/g)^(1/3), velocity sqrt(g h_r) n.  If the inside flow is itself critical towards the edge
    #    (un = sqrt(g h)), the outside state equals it and the flux is the physical flux; where the flow points inward
    #    (un < 0) both states are declared dry and the edge contributes nothing.
    hc = 0.9
    cc = math.sqrt(G * hc)
    m = one_quad(0.0)
    cn, sn, ln = edge_normals(m)
    fl = np.zeros((4, 3))
    rhs = np.zeros((1, 3))
    for e in range(4):
        un = cc * cn[e]                       # state (hc, cc, 0): critical along +x
        if abs(cn[e] - 1.0) < 1e-12:          # the +x edge: critical outflow = physical flux
            fl[e] = physical_flux(hc, cc, 0.0, cn[e], sn[e])
            rhs[0] -= fl[e] * ln[e] / 1.0
        elif un < 0.0:                        # the -x edge: inflow -> dry / dry, skipped
            fl[e] = np.nan                    # the reference leaves the NaN of its 0/0 Roe average in boundary_fluxes (quirk 3)
        else:                                 # n = +-y: un = 0 -> q = 0 -> outside depth 0: a dam break into a dry bed
            fl[e] = None
    out["critical_outflow"] = dict(case=one_cell_case(m, (hc, hc * cc, 0.0), M.CONDITION_CRITICAL_OUTFLOW), flux=fl, rhs=None,
                                   rhs_partial=None, exact=[], tol=1e-13, flux_rows=[e for e in range(4) if abs(cn[e] - 1.0) < 1e-12],
                                   nan_rows=[e for e in range(4) if cc * cn[e] < 0.0])

    # 4b. Roe's property U on an isolated shock: two states joined by ONE shock (Rankine-Hugoniot with speed s:
    #     h_l (u_l - s) = h_r (u_r - s) = m, m^2 = g h_l h_r (h_l + h_r) / 2) are resolved exactly by the Roe linearisation;
    #     with s > 0 the interface flux is the physical flux of the LEFT state.  Cell = pre-shock state (shallow), Dirichlet
    #     value = post-shock state (deep) on the +x edge; only that edge's row is checked.
    hl_, hr_ = 1.0, 2.0
    mflux = math.sqrt(0.5 * G * hl_ * hr_ * (hl_ + hr_))
    s_shock = 0.5
    ul_ = s_shock + mflux / hl_
    ur_ = s_shock + mflux / hr_
    m = one_quad(0.0)
    cn, sn, _ = edge_normals(m)
    fl = np.zeros((4, 3))
    rows = [e for e in range(4) if abs(cn[e] - 1.0) < 1e-12]
    for e in rows:
        fl[e] = physical_flux(hl_, ul_, 0.0, 1.0, 0.0)
    out["roe_isolated_shock"] = dict(case=one_cell_case(m, (hl_, hl_ * ul_, 0.0), M.CONDITION_DIRICHLET, bvalues=(hr_, hr_ * ur_, 0.0)),
                                     flux=fl, rhs=None, exact=[], tol=1e-13, flux_rows=rows)

    # 5. friction, closed cell at rest-pressure balance with a uniform velocity is impossible in one cell (walls reflect), so
    #    friction is isolated with a Dirichlet boundary that repeats the cell's state: the flux sum is zero (entry 3) and
    #    F = -tb.  Semi-implicit (src/swe/swe_petsc.c:764-780): tb = (hu) k/(1 + dt k), k = g n^2 h^(-1/3) |u| / h.
    hf, uf, vf, nm, dt = 1.3, 0.6, -0.25, 0.03, 0.2
    m = one_quad(0.0)
    k = G * nm * nm * hf ** (-1.0 / 3.0) * math.hypot(uf, vf) / hf
    fac = k / (1.0 + dt * k)
    out["friction_semi_implicit"] = dict(
        case=one_cell_case(m, (hf, hf * uf, hf * vf), M.CONDITION_DIRICHLET, bvalues=(hf, hf * uf, hf * vf), mannings=nm, dt=dt),
        flux=None, rhs=np.array([[0.0, -hf * uf * fac, -hf * vf * fac]]), exact=[], tol=1e-13)

    # 6. XQ2018 (src/swe/swe_petsc.c:876-907; Xia & Liang 2018): m = hu (flux sum and bed slope zero),
    #    lambda = g n^2 h^(-4/3) |m/h|; dt lambda < threshold: q = m; else q = (m - m sqrt(1 + 4 dt lambda)) / (-2 dt lambda);
    #    F = -g n^2 h^(-7/3) q |q|.   (a) n = 0: no friction at all; (b) the implicit branch; (c) the threshold branch
    def xq(hh, uu, vv, nn, dtt, thr):
        mx, my = hh * uu, hh * vv
        lam = G * nn * nn * hh ** (-4.0 / 3.0) * math.hypot(mx / hh, my / hh)
        if dtt * lam < thr:
            qx, qy = mx, my
        else:
            r = math.sqrt(1.0 + 4.0 * dtt * lam)
            qx, qy = (mx - mx * r) / (-2.0 * dtt * lam), (my - my * r) / (-2.0 * dtt * lam)
        t = G * nn * nn * hh ** (-7.0 / 3.0) * math.hypot(qx, qy)
        return np.array([[0.0, -t * qx, -t * qy]])
    for name, nn, thr in (("xq2018_no_manning", 0.0, 1e-10), ("xq2018_implicit_branch", 0.03, 1e-10), ("xq2018_threshold_branch", 0.03, 1e3)):
        cfg = RDyFlowConfig(source_method=SOURCE_IMPLICIT_XQ2018, xq2018_threshold=thr)
        out[name] = dict(case=one_cell_case(one_quad(0.0), (hf, hf * uf, hf * vf), M.CONDITION_DIRICHLET, bvalues=(hf, hf * uf, hf * vf),
                                            config=cfg, mannings=nn, dt=dt),
                         flux=None, rhs=xq(hf, uf, vf, nn, dt, thr), exact=[], tol=1e-13)

    # 7. bed slope: a tilted closed cell at rest with a HORIZONTAL water surface cannot be represented by one constant depth,
    #    but the bed-slope term itself can: F_hu = -g h dz/dx + (pressure sum = 0) for water at rest in a closed cell
    #    (src/swe/swe_petsc.c:756-757, 783-785)
    sx, sy = 0.05, -0.02
    m = one_quad(0.0, z=(0.0, sx, sx + sy, sy))
    out["bed_slope"] = dict(case=one_cell_case(m, (h, 0.0, 0.0), M.CONDITION_REFLECTING), flux=None,
                            rhs=np.array([[0.0, -G * h * sx, -G * h * sy]]), exact=[("rhs", 0)], tol=1e-13)

    # 8. external source: added as is (783-785)
    out["external_source"] = dict(case=one_cell_case(one_quad(0.0), (h, 0.0, 0.0), M.CONDITION_REFLECTING, src=(1e-5, 2e-3, -3e-3)), flux=None,
                                  rhs=np.array([[1e-5, 2e-3, -3e-3]]), exact=[], tol=1e-13)
    return out


def hr_two_cell_step():
    """Hydrostatic reconstruction (src/swe/swe_petsc.c:1000-1161): two unit squares side by side, the right one 0.4 m
    higher, water at rest with a level surface eta = 1 (depths 1.0 and 0.6), closed walls.  Well balanced means F = 0:
    at the interior edge both depths are reconstructed against z_max = 0.4 (h* = 0.6 on both sides), the Roe flux of the
    two equal states is the pressure g h*^2/2, and the correction g (h^2 - h*^2)/2 restores each cell's own wall pressure."""
    xyz = np.array([[0, 0, 0.0], [1, 0, 0.0], [2, 0, 0.4], [0, 1, 0.0], [1, 1, 0.0], [2, 1, 0.4]], dtype=float)
    xyz[1, 2] = xyz[4, 2] = 0.0
    conn = np.array([[0, 1, 4, 3], [1, 2, 5, 4]], dtype=np.int32)
    m = M.build_mesh(xyz, conn, boundary_classifier=M.single_boundary(), project_2d=True)
    m.cell_zc = np.array([0.0, 0.4])          # per-cell bed elevation (grid.cell_elevation overrides the vertex mean)
    m.cell_dz_dx[:] = 0.0
    m.cell_dz_dy[:] = 0.0
    u = np.array([[1.0, 0.0, 0.0], [0.6, 0.0, 0.0]])
    case = CS.Case("hr_step", m, RDyFlowConfig(well_balancing=WELL_BALANCING_HR), [M.CONDITION_REFLECTING], u, np.zeros(2), np.zeros((2, 3)), {}, 0.1)
    return case, np.zeros((2, 3))


def second_order_linear_field(nx=9, ny=7, jitter=0.2, seed=5):
    """Second order without a limiter on a state that is LINEAR in the conserved variables, flat frictionless bed, no
    sources: the least-squares gradient of a cell with two non-collinear neighbours is exact, so both reconstructions at
    an edge midpoint give the field's value there, the Roe flux of two equal states is the physical flux, and a cell
    none of whose edges lies on the boundary (and none of whose neighbours has a degenerate stencil) must receive
        F = -(1/A) sum_edges  F_phys(q(x_mid)) . n_out  len
    (ApplyInteriorFlux2R, src/swe/swe_petsc.c:98-213; ReconstructFaceValues, src/operator_fluxes_ceed.c:1155-1206).
    Returns (case, expected rows, mask of the cells the statement holds for)."""
    from rdycore_amd.operator import LIMITER_NONE
    rng = np.random.default_rng(seed)
    m0 = M.structured_tri_mesh(nx, ny)
    xyz = m0.xyz.copy()
    lo, hi = xyz[:, :2].min(0), xyz[:, :2].max(0)
    inner = np.all((xyz[:, :2] > lo + 1e-9) & (xyz[:, :2] < hi - 1e-9), axis=1)
    dx = (hi - lo) / np.array([nx, ny])
    xyz[inner, :2] += jitter * dx * (rng.random((int(inner.sum()), 2)) - 0.5)
    xyz[:, 2] = 0.0
    m = M.build_mesh(xyz, m0.cell_conn[:, :3].astype(np.int32), boundary_classifier=M.single_boundary())

    def q(x, y):
        return np.stack([2.0 + 0.10 * x + 0.05 * y, 0.3 + 0.02 * x - 0.01 * y, -0.2 + 0.01 * x + 0.03 * y], axis=-1)

    u = q(m.cell_centroids[:, 0], m.cell_centroids[:, 1])
    no = m.num_owned_cells
    cfg = RDyFlowConfig(second_order=True, limiter=LIMITER_NONE)
    case = CS.Case("so_linear", m, cfg, [M.CONDITION_REFLECTING], u, np.zeros(no), np.zeros((no, 3)), {}, 0.1)
    # midpoint-rule flux integral, edge by edge, from the mesh's own edge geometry (normal = left -> right)
    rhs = np.zeros((m.num_cells, 3))
    interior = np.ones(m.num_cells, bool)
    v = m.edge_vertex_ids.reshape(-1, 2)
    mid = 0.5 * (m.xyz[v[:, 0], :2] + m.xyz[v[:, 1], :2])
    qm = q(mid[:, 0], mid[:, 1])
    fl = physical_flux(qm[:, 0], qm[:, 1] / qm[:, 0], qm[:, 2] / qm[:, 0], m.edge_cn, m.edge_sn) * m.edge_lengths[:, None]
    for e in range(m.num_edges):
        l, r = m.edge_cell_ids[2 * e], m.edge_cell_ids[2 * e + 1]
        if r < 0:
            interior[l] = False
            continue
        rhs[l] -= fl[e] / m.cell_areas[l]
        rhs[r] += fl[e] / m.cell_areas[r]
    # a neighbour with fewer than two internal edges (a corner triangle) has a degenerate stencil -> zero gradient
    # (operator_fluxes_ceed.c:947-955): its reconstruction is not exact, so its neighbours are left out as well
    nn = np.zeros(m.num_cells, int)
    for e in m.edge_internal_ids:
        nn[m.edge_cell_ids[2 * e]] += 1
        nn[m.edge_cell_ids[2 * e + 1]] += 1
    for e in m.edge_internal_ids:
        l, r = m.edge_cell_ids[2 * e], m.edge_cell_ids[2 * e + 1]
        if nn[l] < 2:
            interior[r] = False
        if nn[r] < 2:
            interior[l] = False
    return case, rhs[:no], interior[:no]
