"""GPU: device-side forcing ingestion (rdyhip_forcing_*, rdycore_amd/forcing.py)
against the oracle's restatement of RDyApplyForcing's loops.  Copies and one
multiply: the bar is bit-exact for the filled arrays and the maps, RHS L-inf
<= 1e-10 for the RHS evaluated with them."""
import numpy as np
import pytest

from oracle import oracle as O
from rdycore_amd import cases as CS
from rdycore_amd import forcing as F
from rdycore_amd import mesh as M
from rdycore_amd.mesh import CONDITION_DIRICHLET

from helpers import oracle_from_case, rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def raster_vec(ncols, nrows, xlc, ylc, cs, rng):
    return np.concatenate([[ncols, nrows, xlc, ylc, cs], rng.uniform(0.0, 50.0, ncols * nrows)])


def raster_centroids(ncols, nrows, xlc, ylc, cs):
    xs = xlc + np.arange(ncols) * cs + cs / 2.0
    ys = ylc + (nrows - 1 - np.arange(nrows)) * cs + cs / 2.0
    return np.tile(xs, nrows), np.repeat(ys, ncols)


def test_nearest_maps_bitwise(rdyhip_kernel):
    if rdyhip_kernel == "cell":
        pytest.skip("independent of the RHS kernel variant")
    torch = _torch()
    rng = np.random.default_rng(3)
    n = 5000
    mx, my = rng.uniform(-20, 120, n), rng.uniform(-20, 80, n)
    # raster with ties (mesh points exactly between pixels) and points beyond the search radius
    ncols, nrows, cs = 37, 23, 2.5
    mx[:50] = 10.0 + 2.5 * np.arange(50)
    my[:50] = 7.5
    mx[50:60] = 1e4
    px, py = raster_centroids(ncols, nrows, 0.0, 0.0, cs)
    ref = O.forcing_raster_map(mx, my, ncols, nrows, cs, px, py)
    got = F.nearest_map(mx, my, px, py, (max(ncols, nrows) + 1) * cs, "cuda").cpu().numpy()
    assert np.array_equal(got, ref)
    # unstructured: scattered points, sizes that are not a multiple of the LDS chunk
    for nd in (1, 255, 256, 257, 1031):
        qx, qy = rng.uniform(0, 100, nd), rng.uniform(0, 60, nd)
        ref = O.forcing_unstructured_map(mx, my, qx, qy)
        got = F.nearest_map(mx, my, qx, qy, -1.0, "cuda").cpu().numpy()
        assert np.array_equal(got, ref), nd


def test_apply_forcing_fills_match_reference_loops():
    torch = _torch()
    rng = np.random.default_rng(11)
    nx, ny = 24, 16
    mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=2 * np.pi / 20))
    case = CS.friction_slope_case(mesh, nx, ny, dt=1e-2, K=2 * np.pi / 20)
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    no = mesh.num_owned_cells
    oc = mesh.owned_centroids()

    # three disjoint regions: raster rain, a homogeneous series, a constant; the rest keeps the case's source
    perm = rng.permutation(no).astype(np.int32)
    r_raster, r_homog, r_const = perm[:300], perm[300:500], perm[500:520]
    vec = raster_vec(9, 7, -1.0, -2.0, 3.0, rng)
    ras = F.RasterDataset(vec, oc[r_raster, 0], oc[r_raster, 1], "cuda")
    px, py = raster_centroids(9, 7, -1.0, -2.0, 3.0)
    assert np.array_equal(ras.data_xc, px) and np.array_equal(ras.data_yc, py)
    ref_map = O.forcing_raster_map(oc[r_raster, 0], oc[r_raster, 1], 9, 7, 3.0, px, py)
    assert np.array_equal(ras.d_map.cpu().numpy(), ref_map)
    table = np.array([[0.0, 2e-5], [100.0, 6e-5], [200.0, 1e-5]])
    hom = F.HomogeneousDataset(table, temporally_interpolate=True)
    dirichlet = [b for b, t in enumerate(case.condition_types) if t == CONDITION_DIRICHLET][0]
    btable = np.array([[0.0, 0.8], [150.0, 1.4]])
    bhom = F.HomogeneousDataset(btable, temporally_interpolate=False)

    frc = F.Forcing(op)
    frc.add_raster_source(r_raster, ras)
    frc.add_homogeneous_source(r_homog, hom)
    frc.add_constant_source(r_const, 3.25e-6)
    frc.add_homogeneous_boundary(dirichlet, bhom)

    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.zeros((no, 3), dtype=torch.float64, device="cuda")
    for time in (0.0, 50.0, 160.0, 1e4):
        frc.apply(time)
        # the reference's loops, through the oracle
        orc.external_sources[r_raster, 0] = O.forcing_set_raster(vec, 5, ref_map)
        orc.external_sources[r_homog, 0] = O.forcing_current_data(table, time, True)[1]
        orc.external_sources[r_const, 0] = 3.25e-6
        orc.boundary_values[dirichlet][:] = [O.forcing_current_data(btable, time, False)[1], 0.0, 0.0]
        assert np.array_equal(op.external_sources.cpu().numpy(), orc.external_sources), time
        op.rhs_function(case.dt, u, f)
        fr = orc.apply(case.dt, case.u_local)
        torch.cuda.synchronize()
        assert rel_linf(f.cpu().numpy(), fr) <= TOL
        bf, rf = op.boundary_fluxes(dirichlet), orc.boundary_fluxes[dirichlet]
        assert rel_linf(np.nan_to_num(bf), np.nan_to_num(rf)) <= TOL

    # the next hourly raster file keeps the map, replaces the values (rdyforcing_dataset.c:166-196)
    assert not ras.needs_next_file(3599.0) and ras.needs_next_file(3600.0)
    vec2 = raster_vec(9, 7, -1.0, -2.0, 3.0, rng)
    ras.load_next(vec2)
    frc.apply(3600.0)
    orc.external_sources[r_raster, 0] = O.forcing_set_raster(vec2, 5, ref_map)
    orc.external_sources[r_homog, 0] = O.forcing_current_data(table, 3600.0, True)[1]
    assert np.array_equal(op.external_sources.cpu().numpy(), orc.external_sources)
    with pytest.raises(F._lib.RDyHipError):
        ras.load_next(raster_vec(8, 7, -1.0, -2.0, 3.0, rng))


def test_unstructured_source_and_boundary_datasets():
    torch = _torch()
    rng = np.random.default_rng(5)
    nx, ny = 16, 12
    mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=2 * np.pi / 20))
    case = CS.friction_slope_case(mesh, nx, ny, dt=1e-2, K=2 * np.pi / 20)
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    no = mesh.num_owned_cells
    oc = mesh.owned_centroids()
    nd = 41
    qx, qy = rng.uniform(0, nx, nd), rng.uniform(0, ny, nd)
    svec = np.concatenate([[nd, 1], rng.uniform(0, 1e-4, nd)])
    sds = F.UnstructuredDataset(svec, 1, qx, qy, oc[:, 0], oc[:, 1], "cuda")
    dirichlet = [b for b, t in enumerate(case.condition_types) if t == CONDITION_DIRICHLET][0]
    ec = mesh.edge_centroids[mesh.boundaries[dirichlet].edge_ids]
    nb = 9
    bx, by = np.zeros(nb), np.linspace(0, ny, nb)
    bvec = np.concatenate([[nb, 3], np.stack([rng.uniform(0.5, 1.5, nb), rng.normal(size=nb) * 0.1, rng.normal(size=nb) * 0.1], 1).ravel()])
    bds = F.UnstructuredDataset(bvec, 3, bx, by, ec[:, 0], ec[:, 1], "cuda")
    frc = F.Forcing(op)
    frc.add_unstructured_source(None, sds)
    frc.add_unstructured_boundary(dirichlet, bds)
    frc.apply(0.0)
    smap = O.forcing_unstructured_map(oc[:, 0], oc[:, 1], qx, qy)
    bmap = O.forcing_unstructured_map(ec[:, 0], ec[:, 1], bx, by)
    assert np.array_equal(sds.d_map.cpu().numpy(), smap) and np.array_equal(bds.d_map.cpu().numpy(), bmap)
    orc.external_sources[:, 0] = O.forcing_set_unstructured(svec, 1, smap)[:, 0]
    orc.boundary_values[dirichlet][:] = O.forcing_set_unstructured(bvec, 3, bmap)
    assert np.array_equal(op.external_sources.cpu().numpy(), orc.external_sources)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    f = torch.zeros((no, 3), dtype=torch.float64, device="cuda")
    op.rhs_function(case.dt, u, f)
    fr = orc.apply(case.dt, case.u_local)
    torch.cuda.synchronize()
    assert rel_linf(f.cpu().numpy(), fr) <= TOL
    assert rel_linf(np.nan_to_num(op.boundary_fluxes(dirichlet)), np.nan_to_num(orc.boundary_fluxes[dirichlet])) <= TOL


def test_forcing_argument_errors():
    _torch()
    mesh = M.structured_tri_mesh(6, 4)
    case = CS.dam_break_case(mesh, 6.0)
    op = CS.create_operator(case)
    L = F._lib.load()
    assert L.rdyhip_forcing_fill_source(op._h, 3, 1, None, 1.0, None) == 83
    assert L.rdyhip_forcing_fill_source(op._h, 0, mesh.num_owned_cells + 1, None, 1.0, None) == 60
    assert L.rdyhip_forcing_fill_boundary(op._h, 99, 1, 1.0, None) == 83
    assert L.rdyhip_forcing_fill_boundary(op._h, 0, mesh.boundaries[0].num_edges + 1, 1.0, None) == 83
    assert L.rdyhip_forcing_gather_source(op._h, 0, 4, None, None, None, 1, 0, 1.0, None) == 83


def test_time_stepping_with_forcing_matches_the_oracle_loop():
    """the driver's loop -- RDyApplyForcing, then RDyAdvance over one coupling interval -- on the device
    (EulerStepper with a Forcing, fused Euler steps) against the same loop on the oracle"""
    torch = _torch()
    from rdycore_amd.timestep import EulerStepper
    nx, ny = 16, 10
    mesh = M.structured_tri_mesh(nx, ny, 1.0, zfunc=CS.mms_bathymetry(K=2 * np.pi / 20))
    case = CS.friction_slope_case(mesh, nx, ny, dt=5e-3, K=2 * np.pi / 20, dry_disc=False)
    op = CS.create_operator(case)
    orc = oracle_from_case(case)
    rain = np.array([[0.0, 0.0], [0.05, 4e-3], [0.1, 1e-3], [0.2, 0.0]])
    stage = np.array([[0.0, 1.0], [0.1, 1.3]])
    dirichlet = [b for b, t in enumerate(case.condition_types) if t == CONDITION_DIRICHLET][0]
    frc = F.Forcing(op)
    frc.add_homogeneous_source(None, F.HomogeneousDataset(rain, temporally_interpolate=True))
    frc.add_homogeneous_boundary(dirichlet, F.HomogeneousDataset(stage, temporally_interpolate=False))
    st = EulerStepper(op, forcing=frc)
    u = torch.tensor(case.u_local, dtype=torch.float64, device="cuda")
    uc = case.u_local.copy()
    t, interval = 0.0, 0.025
    for _ in range(9):
        st.advance(u, case.dt, interval)
        orc.external_sources[:, 0] = O.forcing_current_data(rain, t, True)[1]
        orc.boundary_values[dirichlet][:] = [O.forcing_current_data(stage, t, False)[1], 0.0, 0.0]
        t_end = t + interval
        while t < t_end * (1.0 - 1e-14):
            h = min(case.dt, t_end - t)
            uc = uc + h * orc.apply(h, uc)
            t += h
    torch.cuda.synchronize()
    assert st.step == 45
    assert rel_linf(u.cpu().numpy(), uc) <= 1e-10
