"""CPU: the forcing oracle (oracle/forcing_oracle.c) against hand-computed known
answers, and the product's host-side time lookup (rdycore_amd/forcing.py)
against the oracle.  The device kernels are covered by tests/test_gpu_forcing.py."""
import numpy as np

from oracle import oracle as O
from rdycore_amd import forcing as F

TABLE = np.array([[0.0, 1.0], [3600.0, 3.0], [7200.0, 0.5], [10800.0, 2.0]])


def test_current_data_known_answers():
    # inside an interval: the lower value, or the linear interpolant
    assert O.forcing_current_data(TABLE, 1800.0, False) == (0, 1.0)
    assert O.forcing_current_data(TABLE, 1800.0, True) == (0, 2.0)
    # the interval is closed on the left, open on the right
    assert O.forcing_current_data(TABLE, 3600.0, False) == (1, 3.0)
    assert O.forcing_current_data(TABLE, 7199.999, False)[0] == 1
    # before the first time and after the last: the last value (rdyforcing_dataset.c:55-58)
    assert O.forcing_current_data(TABLE, -1.0, True) == (3, 2.0)
    assert O.forcing_current_data(TABLE, 10800.0, True) == (3, 2.0)
    assert O.forcing_current_data(TABLE, 1e9, False) == (3, 2.0)
    # a single-entry table
    assert O.forcing_current_data(TABLE[:1], 5.0, True) == (0, 1.0)


def test_host_time_lookup_matches_oracle_bitwise():
    rng = np.random.default_rng(7)
    for _ in range(20):
        n = int(rng.integers(1, 12))
        t = np.cumsum(rng.uniform(0.5, 100.0, n))
        tab = np.stack([t, rng.normal(size=n)], axis=1)
        for cur in np.concatenate([rng.uniform(t[0] - 10, t[-1] + 10, 30), t]):
            for interp in (False, True):
                assert F.current_data(tab, float(cur), interp) == O.forcing_current_data(tab, float(cur), interp)


def test_homogeneous_dataset_refill_rule():
    # rdyforcing_dataset.c:334: the array is rewritten when interpolating or when the interval index changes
    ds = F.HomogeneousDataset(TABLE, temporally_interpolate=False)
    assert ds.advance(10.0) == 1.0
    assert ds.advance(20.0) is None
    assert ds.advance(3600.0) == 3.0
    ds = F.HomogeneousDataset(TABLE, temporally_interpolate=True)
    assert ds.advance(900.0) == 1.5
    assert ds.advance(900.0) == 1.5


def test_raster_and_unstructured_loops_known_answers():
    # 3 columns x 2 rows raster, header of 5, values in mm/h
    vec = np.array([3, 2, 0.0, 0.0, 10.0, 36.0, 72.0, 360.0, 3.6, 7.2, 0.0])
    rain = O.forcing_set_raster(vec, 5, [0, 2, 5, 3])
    assert np.array_equal(rain, np.array([36.0, 360.0, 0.0, 3.6]) * (1.0 / (1000.0 * 3600.0)))
    uns = np.array([3, 3, 1, 2, 3, 4, 5, 6, 7, 8, 9], dtype=float)
    out = O.forcing_set_unstructured(uns, 3, [2, 0])
    assert np.array_equal(out, [[7, 8, 9], [1, 2, 3]])


def test_nearest_neighbour_maps_known_answers():
    # raster centroids as rdyforcing.c:254-260 lays them out: top row first
    ncols, nrows, cs = 3, 2, 10.0
    xs = 0.0 + np.arange(ncols) * cs + cs / 2
    ys = 0.0 + (nrows - 1 - np.arange(nrows)) * cs + cs / 2
    px, py = np.tile(xs, nrows), np.repeat(ys, ncols)
    mx = np.array([1.0, 29.0, 14.0, 10.0, 1000.0])
    my = np.array([19.0, 1.0, 10.0, 15.0, 1000.0])
    m = O.forcing_raster_map(mx, my, ncols, nrows, cs, px, py)
    # (14,10): equidistant from the two rows -> first index; (10,15): equidistant from columns 0 and 1 -> first;
    # the far point is beyond (max(ncols,nrows)+1)*cellsize from every pixel: the calloc'ed 0 stays
    assert m.tolist() == [0, 5, 1, 0, 0]
    m2 = O.forcing_unstructured_map(mx, my, px, py)
    assert m2.tolist() == [0, 5, 1, 0, 2]
