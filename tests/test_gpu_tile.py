"""GPU tests of the sixteen-points-per-wavefront matfun kernel (csrc/letkf_tile.hip) through the C ABI
(mia_letkf_analysis_matfun_f32 via LetkfEngine.analysis): parity with the oracle, with the per-point kernel it
replaces, invariance of a point's result under tile composition, the split path for tiles whose union does not fit,
non-finite records, overflow, declined points.  Every test runs on both sets of instantiations: products as split
half-precision MFMAs (option tile_split = 1, the default) and as f32 MFMAs (0)."""
import numpy as np
import pytest
import torch

from conftest import rel_fro, set_option
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu
TOL32 = 1e-5


@pytest.fixture(scope="module")
def eng():
    import torch_assimilate_amd as mia
    mia.build()
    return mia.LetkfEngine("cuda:0")


@pytest.fixture(params=[1, 0], ids=["split", "f32"])
def split(request):
    set_option("tile_split", request.param)
    return request.param


def close(a, b, split, tol=1e-6):
    """bit for bit on the f32 products; to rounding on the split-precision ones, whose operand scale is the tile's"""
    if split:
        assert rel_fro(a, b) < tol
    else:
        np.testing.assert_array_equal(a, b)


_ORACLE = {}


def oracle_analysis(key, case, c, inf):
    """float64 oracle of a synthetic case, computed once for both kernel variants (most of this file's run time)"""
    if key + (inf,) not in _ORACLE:
        _ORACLE[key + (inf,)] = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, inf)[0]
    return _ORACLE[key + (inf,)]


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def run(eng, case, nb, inf=1.1, **kw):
    xa, fl = eng.analysis(dev(case["state"]), dev(case["yb"]), dev(case["d"]), nb, inf, return_flags=True,
                          method="matfun", **kw)
    return xa.cpu().numpy(), fl.cpu().numpy()


@pytest.mark.parametrize("k,stride,c,m", [(40, 2, 10.0, 1), (40, 2, 10.0, 3), (10, 1, 1.6, 1), (24, 2, 6.5, 2), (64, 2, 12.0, 1),
                                          (40, 1, 10.0, 1), (64, 1, 13.0, 2), (20, 3, 12.0, 1), (33, 2, 3.0, 5),
                                          (80, 1, 16.5, 1), (96, 1, 20.0, 2), (72, 1, 12.0, 1)])
def test_tile_kernel_vs_oracle_and_per_point_kernel(eng, split, k, stride, c, m):
    """Union tiles of 1 .. 6 sixteen-row blocks (p_max 4 .. 80, incl. config 4's k = 80 with 64 local observations), 1 .. 6
    member blocks, ragged last tile (G = 203),
    several state rows.  Against the oracle (north-star tolerance, also on the increments) and against the per-point
    kernel on the same lists (option tile = 0): same mathematics, different summation order."""
    case = O.synthetic_case(203, k, stride, seed=k + m, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    assert nb.p_max <= min(k, 88)
    for inf in (1.0, 1.1):
        xa, fl = run(eng, case, nb, inf)
        assert int((fl & 0xff).max()) == 0 and int(((fl >> 8) & 0xff).min()) >= 3
        ref = oracle_analysis((k, stride, c, m), case, c, inf)
        assert rel_fro(xa, ref) < TOL32
        mean = case["state"].mean(axis=1, keepdims=True)
        assert rel_fro(xa - mean, ref - mean) < 5e-5
    set_option("tile", 0)
    xo, fo = run(eng, case, nb, 1.1)
    assert rel_fro(xa, xo) < 3e-6
    if split:       # (half-precision operands in the Gershgorin product: margin 1.002 instead of 1.0001 -- a table entry up at times)
        assert int(np.abs(((fl >> 8) & 0xff) - ((fo >> 8) & 0xff)).max()) <= 1
    else:
        np.testing.assert_array_equal((fl >> 8) & 0xff, (fo >> 8) & 0xff)     # same bound, same table entry, same degree


def test_result_of_a_point_does_not_depend_on_its_tile(eng, split):
    """Slots follow the rank of the observation index and the products enumerate them in that order, so a point's own
    observations are always summed in the same order: shards that cut the grid anywhere (other tile compositions, other
    unions) reproduce the full run BIT FOR BIT on the f32 products.  The split-precision products scale their operands by
    the tile's largest record value, so there a point's result depends on its tile at rounding level (checked per point)."""
    case = O.synthetic_case(400, 40, 2, seed=3)
    X, yb, d = dev(case["state"]), dev(case["yb"]), dev(case["d"])
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    full = eng.analysis(X, yb, d, nb, 1.1, method="matfun").cpu().numpy()
    for g0, g1 in ((0, 400), (5, 93), (93, 400), (16, 17), (199, 231), (1, 399)):
        part_nb = eng.localize(case["grid_x"], case["obs_x"], [10.0], g0=g0, g1=g1)
        part = eng.analysis(X, yb, d, part_nb, 1.1, method="matfun").cpu().numpy()
        if split:
            err = np.linalg.norm(part - full[:, :, g0:g1], axis=(0, 1)) / np.linalg.norm(full[:, :, g0:g1], axis=(0, 1))
            assert float(err.max()) < 2e-6
        else:
            np.testing.assert_array_equal(part, full[:, :, g0:g1])
    again = eng.analysis(X, yb, d, nb, 1.1, method="matfun").cpu().numpy()
    np.testing.assert_array_equal(again, full)


def test_tiles_whose_union_does_not_fit_are_split(eng, split):
    """A scattered 2-D network whose grid points are visited in random order: sixteen consecutive points share next to
    nothing, the union of their lists (~16 x 10) exceeds the 48 slots of the instantiation and every tile is analysed in
    halves, quarters, ... down to single points.  Same result as the oracle."""
    rs = np.random.RandomState(11)
    G, P, k = 150, 500, 32
    grid, obs = rs.uniform(0, 1, size=(G, 2)), rs.uniform(0, 1, size=(P, 2))
    state = rs.normal(size=(2, k, G))
    hx = rs.normal(size=(k, P)) * 0.7
    yb, d = hx - hx.mean(axis=0), rs.normal(size=P) * 0.7
    nb = eng.localize(grid, obs, [0.05])
    cnt = nb.cnt.cpu().numpy()
    assert 8 <= nb.p_max <= 32 and cnt.reshape(-1)[:144].reshape(9, 16).sum(axis=1).min() > 40
    xa, fl = eng.analysis(dev(state), dev(yb), dev(d), nb, 1.1, return_flags=True, method="matfun")
    assert int((fl.cpu().numpy() & 0xff).max()) == 0
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, 0.05, 1.1)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    # ... and a grid whose points are sorted along x: neighbouring points share most observations, few tiles split
    order = np.argsort(grid[:, 0] * 40 // 1 * 10 + grid[:, 1])
    nb2 = eng.localize(grid[order], obs, [0.05])
    xa2 = eng.analysis(dev(state[:, :, order]), dev(yb), dev(d), nb2, 1.1, method="matfun")
    assert rel_fro(xa2.cpu().numpy(), ref[:, :, order]) < TOL32
    close(xa2.cpu().numpy(), xa.cpu().numpy()[:, :, order], split)      # (composition-independent)


def test_non_finite_record_stays_with_the_points_that_use_it(eng, split):
    """A NaN observation must poison exactly the grid points whose lists contain it (as in the reference, where every
    point gathers its own block) -- not the other columns of the tile through the shared Gram matrix."""
    case = O.synthetic_case(160, 40, 2, seed=5)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    clean, _ = run(eng, case, nb)
    bad = dict(case)
    bad["yb"] = case["yb"].copy()
    j = 37                                            # observation at x = 74: seen by grid points 55 .. 93
    bad["yb"][3, j] = np.nan
    xa, fl = run(eng, bad, nb)
    idx, cnt = nb.idx.cpu().numpy(), nb.cnt.cpu().numpy()
    sees = np.array([j in idx[g, :cnt[g]] for g in range(160)])
    assert 30 < sees.sum() < 45
    assert np.isnan(xa[:, :, sees]).all(axis=(0, 1)).all() and ((fl[sees] & 4) != 0).all()
    close(xa[:, :, ~sees], clean[:, :, ~sees], split)       # (the affected tiles are analysed point by point)
    assert ((fl[~sees] & 0xff) == 0).all()
    # a non-finite STATE value stays in its own column anyway
    bad2 = dict(case)
    bad2["state"] = case["state"].copy()
    bad2["state"][0, 7, 100] = np.inf
    xa2, fl2 = run(eng, bad2, nb)
    ok = np.arange(160) != 100
    np.testing.assert_array_equal(xa2[:, :, ok], clean[:, :, ok])
    assert (fl2[100] & 4) != 0 and not np.isfinite(xa2[:, :, 100]).all()


def test_overflowing_list_is_flagged_not_truncated(eng, split):
    from torch_assimilate_amd.engine import NeighbourLists
    case = O.synthetic_case(64, 40, 2, seed=9)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    short = NeighbourLists(nb.cnt, nb.idx, nb.w, nb.p_cap, nb.p_max - 1, nb.g0, nb.g1)     # bound one too small
    xa, fl = run(eng, case, short)
    over = nb.cnt.cpu().numpy() > nb.p_max - 1
    assert over.any() and not over.all()
    assert ((fl[over] & 0xff) == 1).all() and np.isnan(xa[:, :, over]).all()
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)
    assert rel_fro(xa[:, :, ~over], ref[:, :, ~over]) < TOL32 and ((fl[~over] & 0xff) == 0).all()


def test_declined_points_are_redone_by_the_eigensolver(eng, split):
    """Observations accurate enough to push lambda_max / reg beyond the polynomial route for part of the grid: the tile
    kernel declines those points (MIA_FLAG_RETRY, counted), the engine redoes them with the eigensolver kernel."""
    case = O.synthetic_case(300, 40, 2, seed=13)
    scale = np.where(np.arange(150) % 50 < 20, 40.0, 1.0)              # strong observations in three stretches
    yb, d = case["yb"] * scale, case["d"] * scale
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    xa, fl, fin = eng.analysis(dev(case["state"]), dev(yb), dev(d), nb, 1.1, return_flags=True, method="matfun",
                               defer_retry=True)
    declined = (fl.cpu().numpy() & 8) != 0
    n = fin()
    assert n == int(declined.sum()) and 50 < n < 250
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, 10.0, 1.1)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    assert int((fl.cpu().numpy() & 0xff & ~8).max()) == 0


def test_empty_and_tiny_lists(eng, split):
    """Grid points without any local observation return the inflated prior (core/etkf.py:91-95); lists of one."""
    rs = np.random.RandomState(4)
    G, k = 70, 12
    grid = np.arange(G, dtype=np.float64)
    obs = np.array([3.0, 40.0, 41.5])
    state = rs.normal(size=(1, k, G))
    hx = rs.normal(size=(k, 3))
    yb, d = hx - hx.mean(axis=0), rs.normal(size=3)
    nb = eng.localize(grid, obs, [2.0])
    assert int(nb.cnt.min()) == 0 and nb.p_max <= 2
    xa, fl = eng.analysis(dev(state), dev(yb), dev(d), nb, 1.21, return_flags=True, method="matfun")
    assert int((fl.cpu().numpy() & 0xff).max()) == 0
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, 2.0, 1.21)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    far = nb.cnt.cpu().numpy() == 0
    mean = state.mean(axis=1, keepdims=True)
    np.testing.assert_allclose(xa.cpu().numpy()[:, :, far], (mean + 1.1 * (state - mean))[:, :, far], rtol=2e-6, atol=1e-6)


def test_split_products_follow_the_magnitude_of_their_operands(eng):
    """The half-precision operands are scaled by powers of two taken from the data (records per tile, x' and the recurrence
    vectors per column): a state 2^40 times larger or smaller gives the same analysis times 2^+-40 BIT FOR BIT (every scale
    moves with it), and observation-space inputs a million times weaker or sixty times stronger stay inside the north-star
    tolerance (weak: analysis = inflated prior to 1e-7; strong: the points the polynomial route declines go to the eigensolver)."""
    set_option("tile_split", 1)
    case = O.synthetic_case(203, 40, 2, seed=77)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    base, fl = run(eng, case, nb)
    assert int((fl & 0xff).max()) == 0
    for e in (40, -40):
        scaled = dict(case)
        scaled["state"] = case["state"] * 2.0 ** e
        xa, fl = run(eng, scaled, nb)
        assert int((fl & 0xff).max()) == 0
        np.testing.assert_array_equal(xa, base * np.float32(2.0 ** e))
    for s in (2.0 ** -20, 1e-3, 7.0, 60.0):
        scaled = dict(case)
        scaled["yb"], scaled["d"] = case["yb"] * s, case["d"] * s
        xa, fl, fin = eng.analysis(dev(scaled["state"]), dev(scaled["yb"]), dev(scaled["d"]), nb, 1.1, return_flags=True,
                                   method="matfun", defer_retry=True)
        fin()
        assert int((fl.cpu().numpy() & 0xff & ~8).max()) == 0
        ref, _ = O.letkf_analysis(scaled["state"], scaled["grid_x"], scaled["obs_x"], scaled["yb"], scaled["d"], 10.0, 1.1)
        assert rel_fro(xa.cpu().numpy(), ref) < TOL32, s


@pytest.mark.parametrize("k", [2, 3, 5])
def test_tiny_ensembles_with_large_innovations(eng, k):
    """Two to five members, innovations hundreds of times the state's spread (the analysis mean moves by hundreds of spreads): the
    engine's list route (round-2 tile kernel: split products from three members on, f32 products for two -- tools/small_k_sweep.py)
    and the tile route (letkf_tile2_kernel) against the oracle."""
    set_option("tile_split", 1)
    rs = np.random.RandomState(100 + k)
    G = 320
    grid = np.arange(G, dtype=np.float64)
    obs = np.arange(0, G, 3.0) + 0.1
    P = obs.shape[0]
    state = rs.normal(size=(1, k, G)) * 7e-4
    hx = rs.normal(size=(k, P)) * 350.0
    yb, d = hx - hx.mean(axis=0), rs.normal(size=P) * 350.0
    nb = eng.localize(grid, obs, [0.75])
    assert nb.p_max <= k
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, 0.75, 1.0)
    xa, fl, fin = eng.analysis(dev(state), dev(yb), dev(d), nb, 1.0, return_flags=True, method="matfun", defer_retry=True)
    fin()
    assert int((fl.cpu().numpy() & 0xff & ~8).max()) == 0
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    tiles = eng.localize_tiles(grid, obs, [0.75], nb.p_max)
    assert int(tiles.stats[1].item()) == 0
    xa2, fl2, retry = eng.analysis_tiles(dev(state), eng.pack_split(dev(yb), dev(d)), P, tiles, 1.0)
    if int(retry.item()):
        eng.retry_points(dev(state), dev(yb), dev(d), nb, 1.0, xa2, fl2)
    assert rel_fro(xa2.cpu().numpy(), ref) < TOL32
