"""Shared test helpers: drive the CPU oracle and the HIP operator from one Case."""
import numpy as np

from oracle import oracle as O


def oracle_from_case(case):
    cfg = case.config
    orc = O.OracleOperator(case.mesh, case.condition_types, cfg.tiny_h, cfg.h_anuga_regular, cfg.xq2018_threshold, cfg.source_method, cfg.well_balancing,
                           second_order=cfg.second_order, limiter=cfg.limiter)
    orc.mannings[:] = case.mannings
    orc.external_sources[:] = case.ext_src
    for b, vals in case.boundary_values.items():
        orc.boundary_values[b][:] = vals
    return orc


def rel_linf(a, b):
    """max |a-b| relative to max(1, max|b|): the RHS L-inf measure of BASELINE.json."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def second_order_oracle_ranks(cases):
    """One second-order RHS on a partitioned mesh the way the reference runs it
    (ApplyInteriorFlux2R, src/swe/swe_petsc.c:98-213), with the two exchanges done by
    hand: CommunicateCellGradients (ghost gradients from their owners) and
    DMLocalToGlobal(ADD_VALUES) (the ghost rows of each rank's local flux vector added
    onto the owners) between the interior-flux and the boundary/source sub-operators.
    `cases`: one Case per rank, local meshes carrying global cell ids.
    Returns ([F_rank], [oracle_rank])."""
    orcs = [oracle_from_case(c) for c in cases]
    owner = {}
    for r, c in enumerate(cases):
        m = c.mesh
        for lc in m.cell_owned_to_local:
            owner[int(m.cell_global_ids[lc])] = (r, int(lc))
    for o, c in zip(orcs, cases):
        o.compute_gradients(c.u_local)
    own_grads = [[g.copy() for g in o.gradients] for o in orcs]
    for r, (o, c) in enumerate(zip(orcs, cases)):
        m = c.mesh
        for lc in np.nonzero(m.cell_is_owned == 0)[0]:
            pr, plc = owner[int(m.cell_global_ids[lc])]
            for k in range(3):
                o.gradients[k][lc] = own_grads[pr][k][plc]
        o.set_gradients_ready(True)
    fs = [o.apply_interior(c.dt, c.u_local) for o, c in zip(orcs, cases)]
    for r, (o, c) in enumerate(zip(orcs, cases)):
        m = c.mesh
        for lc in np.nonzero(m.cell_is_owned == 0)[0]:
            pr, plc = owner[int(m.cell_global_ids[lc])]
            fs[pr][cases[pr].mesh.cell_local_to_owned[plc]] += o.rhs_local[lc]
    for o, c, f in zip(orcs, cases, fs):
        o.apply_rest(c.dt, c.u_local, f)
    return fs, orcs


def oracle_rk4(orc, u, dt, nsteps):
    """classical Runge-Kutta (TSRK4, src/rdysetup.c:1187-1189) driven by the oracle's RHS on one rank: every stage
    is OperatorRHSFunction with the full step's dt (src/rdysetup.c:1129)"""
    u = u.copy()
    for _ in range(nsteps):
        k1 = orc.apply(dt, u)
        k2 = orc.apply(dt, u + 0.5 * dt * k1)
        k3 = orc.apply(dt, u + 0.5 * dt * k2)
        k4 = orc.apply(dt, u + dt * k3)
        u = u + dt * (k1 / 6.0 + k2 / 3.0 + k3 / 3.0 + k4 / 6.0)
    return u


def interior_courant_numbers(case):
    """Per internal edge (position in edges.internal_edge_ids): the Courant number the reference's interior-flux loop forms
    (src/swe/swe_petsc.c:289; with hydrostatic reconstruction 1046-1071, 1117), restated in numpy for first order and HR --
    only the largest wave speed, chat + |uperp|, of the Roe solver is needed.  NaN where the edge is skipped (both sides dry).
    Used to tell a genuine disagreement about the edge of the maximal Courant number from two edges whose numbers agree to
    rounding (symmetric states: which of them a run reports then hangs on its last bits)."""
    m, cfg = case.mesh, case.config
    g = 9.806
    e = m.edge_internal_ids
    l, r = m.edge_cell_ids[2 * e], m.edge_cell_ids[2 * e + 1]
    u = case.u_local
    hl, hr = u[l, 0].copy(), u[r, 0].copy()

    def vel(h, hu, hv, strict):
        wet = (h > cfg.tiny_h) if strict else ~(h < cfg.tiny_h)
        den = h * h + cfg.h_anuga_regular ** 2
        with np.errstate(all="ignore"):
            return np.where(wet, hu * h / den, 0.0), np.where(wet, hv * h / den, 0.0)
    hr_on = cfg.well_balancing != 0
    ul, vl = vel(u[l, 0], u[l, 1], u[l, 2], hr_on)
    ur, vr = vel(u[r, 0], u[r, 1], u[r, 2], hr_on)
    skip = (u[l, 0] < cfg.tiny_h) & (u[r, 0] < cfg.tiny_h)
    if hr_on:
        zl, zr = m.cell_zc[l], m.cell_zc[r]
        zm = np.maximum(zl, zr)
        hl = np.maximum(0.0, (u[l, 0] + zl) - zm)
        hr = np.maximum(0.0, (u[r, 0] + zr) - zm)
        skip |= ~((hl > cfg.tiny_h) | (hr > cfg.tiny_h))
    with np.errstate(all="ignore"):
        dl, dr = np.sqrt(hl), np.sqrt(hr)
        uhat = (dl * ul + dr * ur) / (dl + dr)
        vhat = (dl * vl + dr * vr) / (dl + dr)
        amax = np.sqrt(0.5 * g * (hl + hr)) + np.abs(uhat * m.edge_cn[e] + vhat * m.edge_sn[e])
        c = amax * m.edge_lengths[e] / np.minimum(m.cell_areas[l], m.cell_areas[r]) * case.dt
    return np.where(skip, np.nan, c)
