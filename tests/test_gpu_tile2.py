"""GPU tests of the tile route (round 3) through the C ABI: split records (mia_letkf_pack_split_f32), tile lists
(mia_letkf_localize_tiles_f64) and the analysis from both (mia_letkf_analysis_tiles_f32, csrc/letkf_tile2.hip).

Parity: tile lists against the per-point lists of mia_letkf_localize_f64 (same masks, same float32 sqrt(weight)), the
analysis against the float64 oracle (north-star tolerance 1e-5 relative Frobenius, also on the increments) and against the
round-2 kernel on per-point lists; edge cases: ragged last tile, no observations, records of very different magnitudes
inside one tile, non-finite records, unions that do not fit."""
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


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def slot_of_rank(rk):
    return 16 * (rk >> 4) + 4 * (rk & 3) + ((rk >> 2) & 3)


def lists_from_tiles(tiles):
    """per-point {obs index: sqrt(weight)} from the tile format + checks of the format itself"""
    hdr, uidx, D = tiles.unpack()
    n = tiles.g1 - tiles.g0
    out = []
    for t in range(hdr.shape[0]):
        U, longest, npts = int(hdr[t, 0]), int(hdr[t, 1]), int(hdr[t, 2])
        assert npts == min(16, n - 16 * t)
        keys = uidx[t]
        used = np.flatnonzero(keys >= 0)
        assert len(used) == U
        # slots follow the rank of the observation index
        order = np.argsort(keys[used], kind="stable")
        for rk, s in enumerate(used[order]):
            assert s == slot_of_rank(rk)
        assert len(set(keys[used].tolist())) == U
        # D[t][tb][lane][q] = sqrt(rho) of (point lane & 15, slot 16 tb + 4 (lane >> 4) + q)
        Dm = np.zeros((16, uidx.shape[1]), dtype=np.float32)
        for tb in range(D.shape[1]):
            for lane in range(64):
                for q in range(4):
                    Dm[lane & 15, 16 * tb + 4 * (lane >> 4) + q] = D[t, tb, lane, q]
        assert not Dm[:, keys < 0].any() and not Dm[npts:].any()
        assert (Dm[:npts] != 0).sum(axis=1).max(initial=0) == longest
        if U:
            assert (Dm != 0).any(axis=0)[used].all()          # every member of the union is local to some point
        for p in range(npts):
            out.append({int(keys[s]): float(Dm[p, s]) for s in np.flatnonzero(Dm[p])})
    return out


def lists_from_points(nb):
    cnt, idx, w = nb.cnt.cpu().numpy(), nb.idx.cpu().numpy(), nb.w.cpu().numpy()
    return [{int(idx[g, j]): float(np.float32(w[g, j])) for j in range(cnt[g])} for g in range(len(cnt))]


GEOMS = {
    "c2": lambda rs: (np.arange(203.0), np.arange(0, 203, 2.0), [10.0], None, 0),
    "c4": lambda rs: (np.arange(150.0), np.arange(150.0), [16.5], None, 0),
    "stride3_inf": lambda rs: (np.arange(120.0) * 0.7, np.arange(0, 84, 3.0), [6.0], None, 1),
    "edge_outside": lambda rs: (np.arange(-30.0, 90.0), np.arange(0, 50, 2.0), [5.0], None, 0),
    "2d_rows": lambda rs: (np.stack(np.meshgrid(np.arange(4.0), np.arange(24.0), indexing="ij"), -1).reshape(-1, 2),
                           rs.uniform(-1, 24, size=(60, 2)) * [0.2, 1.0], [1.6], None, 0),
    "3d_two_radii": lambda rs: (np.stack(np.meshgrid(np.arange(2.0), np.arange(3.0), np.arange(16.0), indexing="ij"), -1).reshape(-1, 3),
                                rs.uniform(0, 1, size=(80, 3)) * [2, 3, 16], [1.5, 2.5], [0, 0, 1], 0),
}


@pytest.mark.parametrize("name", sorted(GEOMS))
def test_tile_lists_equal_the_per_point_lists(eng, name):
    """Same masks (float64 `w > eps` decision) as mia_letkf_localize_f64 and the same sqrt(weight) -- to 2e-6 for the
    Gaspari-Cohn taper, which the tile kernel evaluates in float32 (cancellation-free form), bit for bit for the other taper --
    for 1-D / 2-D / 3-D networks, two radii, grid points outside the observations' bounding box, ragged tiles."""
    grid, obs, radii, cg, taper = GEOMS[name](np.random.RandomState(5))
    nb = eng.localize(grid, obs, radii, cg, taper=taper)
    for g0, g1 in ((0, len(grid)), (7, min(len(grid), 59))):
        part = eng.localize(grid, obs, radii, cg, g0=g0, g1=g1, taper=taper)
        # (2-D / 3-D rows: sixteen consecutive points sweep far more observations than one list holds -- the caller's bound may be
        #  any value >= the longest list, here the largest the format offers)
        tiles = eng.localize_tiles(grid, obs, radii, nb.p_max if grid.ndim == 1 else 88, cg, g0=g0, g1=g1, taper=taper)
        longest, n_over = tiles.stats.tolist()
        if n_over:
            hdr = tiles.unpack()[0]
            assert (hdr[:, 0] < 0).sum() == n_over
            pytest.skip("union of %d tiles exceeds the slots of this bound (scattered points): list route" % n_over)
        assert longest == part.p_max
        got, ref = lists_from_tiles(tiles), lists_from_points(part)
        assert [sorted(a) for a in got] == [sorted(b) for b in ref]                     # the masks: exact
        if taper:
            assert got == ref
        else:
            for a, b in zip(got, ref):
                np.testing.assert_allclose([a[j] for j in sorted(a)], [b[j] for j in sorted(b)], rtol=2e-6 * len(radii), atol=0)


def test_decisions_at_the_edge_of_the_support_are_the_float64_ones(eng):
    """Pairs whose weight is within a few 1e-7 (relative) of eps on either side: float32 cannot tell them apart, the kernel takes
    those decisions again in float64, and the masks equal the per-point lists'."""
    from scipy.optimize import brentq
    eps, c = 1e-5, 10.0

    def gc(r):
        return r ** 5 / 12 - 0.5 * r ** 4 + 0.625 * r ** 3 + 5 / 3 * r ** 2 - 5 * r + 4 - 2 / (3 * r)
    r_eps = brentq(lambda r: gc(r) - eps, 1.5, 1.999)
    grid = np.arange(64.0)
    # observations at distance c r_eps (1 +- delta) from grid points 8, 24, 40, 56 on either side, delta from 1e-9 to 1e-6
    deltas = np.array([-1e-6, -1e-7, -1e-8, -1e-9, 1e-9, 1e-8, 1e-7, 1e-6])
    obs = np.concatenate([g + s * c * r_eps * (1 + deltas) for g in (8.0, 24.0, 40.0, 56.0) for s in (-1, 1)])
    obs = np.concatenate([obs, np.arange(0, 64, 2.0)])
    nb = eng.localize(grid, obs, [c])
    tiles = eng.localize_tiles(grid, obs, [c], nb.p_max, extra_blocks=2)
    assert tiles.stats.tolist()[1] == 0
    got, ref = lists_from_tiles(tiles), lists_from_points(nb)
    assert [sorted(a) for a in got] == [sorted(b) for b in ref]
    for gi, g in enumerate((8, 24, 40, 56)):                    # the probes fall on both sides of the decision
        for si in range(2):
            base = (2 * gi + si) * 8
            assert [j - base for j in sorted(got[g]) if base <= j < base + 8] == [0, 1, 2, 3]


def test_pack_split_records(eng):
    """hi + lo halves times the record's scale reproduce the members to 2^-21 of the record's largest magnitude, also for
    records 1e-6 and 1e5 times the others; tail = (innovation in the record's scale, scale); record P (the source of unused slots) is zero with scale 1; a
    non-finite member or innovation marks the record (scale = NaN)."""
    rs = np.random.RandomState(2)
    for k in (40, 20, 33, 80):
        P = 131
        yb = rs.normal(size=(k, P)).astype(np.float32)
        yb[:, 3] *= 1e-6
        yb[:, 4] *= 1e5
        yb[:, 7] = 0.0
        d = rs.normal(size=P).astype(np.float32)
        d[9] = 300.0
        yb[2, 11] = np.inf
        d[12] = np.nan
        rec = eng.pack_split(dev(yb), dev(d)).cpu().numpy()
        nc8 = (k + 7) // 8
        rb = 32 * nc8 + 16
        rec = rec.reshape(P + 1, rb)
        halves = rec[:, :32 * nc8].copy().view(np.float16).reshape(P + 1, nc8, 2, 8).astype(np.float64)
        tail = rec[:, 32 * nc8:].copy().view(np.float32)
        val = (halves[:, :, 0] + halves[:, :, 1]).reshape(P + 1, nc8 * 8)
        assert not rec[P, :32 * nc8].any() and tail[P].tolist() == [0.0, 1.0, 0.0, 0.0]      # the record of unused slots
        for j in range(P):
            w, E = tail[j, 0], tail[j, 1]
            if j in (11, 12):
                assert np.isnan(E)
                continue
            assert np.log2(E) == np.round(np.log2(E))
            mx = np.abs(yb[:, j]).max()
            if mx > 0:
                assert 2.0 ** 9 <= mx / E < 2.0 ** 10
            np.testing.assert_allclose(val[j, :k] * E, yb[:, j], rtol=0, atol=mx * 2.0 ** -21)
            assert not val[j, k:].any()
            assert w * E == d[j]


CASES = [(40, 2, 10.0, 1), (40, 2, 10.0, 3), (10, 1, 1.6, 1), (24, 2, 6.5, 2), (64, 2, 12.0, 1), (40, 1, 10.0, 1),
         (64, 1, 13.0, 2), (20, 3, 12.0, 1), (33, 2, 3.0, 5), (80, 1, 16.5, 1), (96, 1, 20.0, 2), (72, 1, 12.0, 1)]


def expected_degrees(case, c, inf):
    """Chebyshev degree per grid point as the kernels choose it: Gershgorin bound L of S = D G D, T = 1.002 L / reg rounded up to
    the table's geometric grid (32 per octave), degree = ceil(12 / log rho) + 2, rho = (sqrt(1 + T) + 1) / (sqrt(1 + T) - 1)"""
    yb = case["yb"].astype(np.float32).astype(np.float64)
    k = yb.shape[0]
    reg = (k - 1) / inf
    out = []
    for g in range(len(case["grid_x"])):
        w = O.gaspari_cohn(np.abs(case["obs_x"] - case["grid_x"][g]) / c)
        use = w > 1e-5
        D, Y = np.sqrt(w[use]), yb[:, use]
        L = np.max(D * (np.abs(Y.T @ Y) @ D), initial=1e-300) * 1.002
        Tg = 2.0 ** (np.ceil(32 * np.log2(L / reg)) / 32)
        sq = np.sqrt(1 + Tg)
        out.append(max(3, int(np.ceil(12 / np.log((sq + 1) / (sq - 1))) + 2)))
    return np.array(out)


@pytest.mark.parametrize("k,stride,c,m", CASES)
def test_tile_route_vs_oracle_and_round2_kernel(eng, k, stride, c, m):
    """Union tiles of 1 .. 6 sixteen-row blocks, 1 .. 6 member blocks (incl. config 4's k = 80 with 64 local observations and
    ensemble sizes that are not multiples of eight), ragged last tile (G = 203), one and several state rows.  Against the
    oracle (north-star tolerance, also on the increments) and against the round-2 kernel on per-point lists."""
    case = O.synthetic_case(203, k, stride, seed=k + m, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    if stride == 1 and nb.p_max + 15 > 16 * tiles.ut:
        # one observation per grid step: sixteen consecutive points see p_max + 15 of them -- sixteen more slots
        assert tiles.stats.tolist()[1] > 0
        tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
    assert tiles.stats.tolist() == [nb.p_max, 0]
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    P = case["yb"].shape[1]
    for inf in (1.0, 1.1):
        xa, fl, retry = eng.analysis_tiles(dev(case["state"]), rec, P, tiles, inf)
        xa, fl = xa.cpu().numpy(), fl.cpu().numpy()
        assert int(retry.item()) == 0 and int((fl & 0xff).max()) == 0 and int(((fl >> 8) & 0xff).min()) >= 3
        ref = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, inf)[0]
        assert rel_fro(xa, ref) < TOL32
        mean = case["state"].mean(axis=1, keepdims=True)
        assert rel_fro(xa - mean, ref - mean) < 5e-5
    xo, fo = eng.analysis(dev(case["state"]), dev(case["yb"]), dev(case["d"]), nb, 1.1, return_flags=True, method="matfun")
    assert rel_fro(xa, xo.cpu().numpy()) < 3e-6
    # the degree every point ran at = the table's degree for ITS Gershgorin bound (float64 restatement; half-precision operands
    # in the bound's product: margin 1.002, a table entry up or down at times)
    assert int(np.abs(((fl >> 8) & 0xff) - expected_degrees(case, c, 1.1)).max()) <= 1


def test_shards_and_determinism(eng):
    """Any sub-range [g0, g1) reproduces the full run to rounding per point (other tile compositions); the same call twice is
    bit-for-bit identical."""
    case = O.synthetic_case(400, 40, 2, seed=3)
    X, rec = dev(case["state"]), eng.pack_split(dev(case["yb"]), dev(case["d"]))
    P = case["yb"].shape[1]
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [10.0], 20)
    full = eng.analysis_tiles(X, rec, P, tiles, 1.1)[0].cpu().numpy()
    for g0, g1 in ((5, 93), (93, 400), (16, 17), (199, 231), (1, 399)):
        part_t = eng.localize_tiles(case["grid_x"], case["obs_x"], [10.0], 20, g0=g0, g1=g1)
        part = eng.analysis_tiles(X, rec, P, part_t, 1.1)[0].cpu().numpy()
        err = np.linalg.norm(part - full[:, :, g0:g1], axis=(0, 1)) / np.linalg.norm(full[:, :, g0:g1], axis=(0, 1))
        assert float(err.max()) < 2e-6
    again = eng.analysis_tiles(X, rec, P, tiles, 1.1)[0].cpu().numpy()
    np.testing.assert_array_equal(again, full)


def test_mixed_magnitudes_inside_one_tile(eng):
    """Two observation types whose R^-1/2-normalised perturbations differ by 1e5 (and innovations far larger than the
    perturbations) alternate along the grid, so every tile holds both.  Every record carries its own power of two, so the
    small ones keep their bits: error on the increments as on a uniform network."""
    case = O.synthetic_case(203, 40, 2, seed=17)
    yb, d = case["yb"].copy(), case["d"].copy()
    yb[:, 1::2] *= 1e-5
    d[1::2] *= 1e-5
    yb[:, 0::4] *= 3e-3            # innovation ~ 300 x the spread on every other accurate observation
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [10.0], nb.p_max)
    rec = eng.pack_split(dev(yb), dev(d))
    mean = case["state"].mean(axis=1, keepdims=True)
    for inf in (1.0, 1.1):
        xa, fl, retry = eng.analysis_tiles(dev(case["state"]), rec, yb.shape[1], tiles, inf)
        assert int(retry.item()) == 0
        ref = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, 10.0, inf)[0]
        assert rel_fro(xa.cpu().numpy(), ref) < TOL32
        assert rel_fro(xa.cpu().numpy() - mean, ref - mean) < 5e-5
    # the small type alone decides the analysis where only it is observed: scale everything, results must follow
    for s in (1e-4, 2.0):
        rec_s = eng.pack_split(dev(yb * s), dev(d * s))
        xs, _, retry = eng.analysis_tiles(dev(case["state"]), rec_s, yb.shape[1], tiles, 1.1)
        assert int(retry.item()) == 0
        xs = xs.cpu().numpy()
        ref = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb * s, d * s, 10.0, 1.1)[0]
        assert rel_fro(xs - mean, ref - mean) < 5e-5


def test_no_observations_and_empty_tiles(eng):
    """P = 0 and grid points far from every observation: the inflated prior sqrt(inf) x' + mean (etkf.py:91-95)."""
    case = O.synthetic_case(100, 20, 2, seed=1)
    X = dev(case["state"])
    tiles = eng.localize_tiles(case["grid_x"], np.zeros((0,)), [5.0], 8)
    rec = eng.pack_split(torch.zeros((20, 0)), torch.zeros((0,)))
    xa, fl, retry = eng.analysis_tiles(X, rec, 0, tiles, 1.21)
    mean = case["state"].mean(axis=1, keepdims=True)
    np.testing.assert_allclose(xa.cpu().numpy(), mean + 1.1 * (case["state"] - mean), rtol=2e-6, atol=2e-6)
    far = case["obs_x"][:10]
    nb = eng.localize(case["grid_x"], far, [3.0])
    tiles = eng.localize_tiles(case["grid_x"], far, [3.0], nb.p_max)
    rec = eng.pack_split(dev(case["yb"][:, :10]), dev(case["d"][:10]))
    xa, fl, retry = eng.analysis_tiles(X, rec, 10, tiles, 1.1)
    ref = O.letkf_analysis(case["state"], case["grid_x"], far, case["yb"][:, :10], case["d"][:10], 3.0, 1.1)[0]
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32


def test_non_finite_record_hands_its_tiles_to_the_eigensolver(eng):
    """A NaN perturbation: every point of the tiles whose union holds that observation is flagged MIA_FLAG_RETRY (the shared
    Gram matrix would spread the NaN over all 16 columns) and left to mia_letkf_analysis_retry_f32; all other tiles are
    analysed as usual."""
    case = O.synthetic_case(203, 40, 2, seed=4)
    yb = case["yb"].copy()
    yb[5, 40] = np.nan                     # observation 40 sits at x = 80
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [10.0], nb.p_max)
    rec = eng.pack_split(dev(yb), dev(case["d"]))
    xa, fl, retry = eng.analysis_tiles(dev(case["state"]), rec, yb.shape[1], tiles, 1.1)
    fl = fl.cpu().numpy()
    _, uidx, _ = tiles.unpack()
    bad_tiles = np.flatnonzero((uidx == 40).any(axis=1))
    assert len(bad_tiles) >= 2
    expect = np.zeros(203, dtype=bool)
    for t in bad_tiles:
        expect[16 * t:16 * t + 16] = True
    np.testing.assert_array_equal((fl & 8) != 0, expect)
    assert int(retry.item()) == int(expect.sum())
    clean = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
    keep = ~expect
    # (points of clean tiles do not see observation 40 at all)
    assert rel_fro(xa.cpu().numpy()[:, :, keep], clean[:, :, keep]) < TOL32


def test_union_overflow_is_loud(eng):
    """A scattered 2-D network visited in random order: sixteen consecutive points share next to nothing and the union of
    their lists exceeds the tile's slots -- the tile is reported (stats[1], header -1) and NOT analysed (MIA_FLAG_OVERFLOW,
    NaN), never truncated."""
    rs = np.random.RandomState(11)
    G, P, k = 150, 500, 32
    grid, obs = rs.uniform(0, 1, size=(G, 2)), rs.uniform(0, 1, size=(P, 2))
    state = rs.normal(size=(1, k, G))
    yb, d = rs.normal(size=(k, P)), rs.normal(size=P)
    nb = eng.localize(grid, obs, [0.05])
    tiles = eng.localize_tiles(grid, obs, [0.05], nb.p_max)
    longest, n_over = tiles.stats.tolist()
    assert longest == nb.p_max and n_over > 0
    hdr = tiles.unpack()[0]
    assert (hdr[:, 0] < 0).sum() == n_over
    xa, fl, retry = eng.analysis_tiles(dev(state), eng.pack_split(dev(yb), dev(d)), P, tiles, 1.1)
    xa, fl = xa.cpu().numpy(), fl.cpu().numpy()
    for t in range(hdr.shape[0]):
        sl = slice(16 * t, min(16 * t + 16, G))
        if hdr[t, 0] < 0:
            assert (fl[sl] & 0xff == 1).all() and np.isnan(xa[:, :, sl]).all()
        else:
            assert (fl[sl] & 0xff == 0).all() and np.isfinite(xa[:, :, sl]).all()


def test_declined_points_are_flagged(eng):
    """Strong observations (lambda_max / reg beyond the degree cap): MIA_FLAG_RETRY per point, counted, untouched output."""
    case = O.synthetic_case(120, 40, 2, seed=8)
    set_option("cheb_dmax", 12)
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [10.0], 20)
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    out = torch.full((1, 40, 120), 7.0, dtype=torch.float32, device="cuda:0")
    xa, fl, retry = eng.analysis_tiles(dev(case["state"]), rec, case["yb"].shape[1], tiles, 1.1, out=out)
    fl = fl.cpu().numpy()
    decl = (fl & 8) != 0
    assert decl.sum() > 0 and int(retry.item()) == int(decl.sum())
    assert (xa.cpu().numpy()[:, :, decl] == 7.0).all()
    ref = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
    if (~decl).any():
        assert rel_fro(xa.cpu().numpy()[:, :, ~decl], ref[:, :, ~decl]) < TOL32


@pytest.mark.parametrize("k,c,sy,sx", [(27, 6.25, 2e2, 4e-3), (40, 5.0, 5e2, 1.0), (20, 4.0, 50.0, 1e3)])
def test_strong_observations_are_redone_in_float64(eng, k, c, sy, sx):
    """Observations 50 .. 500 times more accurate than the ensemble spread (lambda_max / reg ~ 1e4 .. 1e6): the matrix-function
    kernel declines every point and the eigensolver redoes them -- in float64 arithmetic on the float32 data, because a spectrum
    that wide is also what a float32 eigensolver resolves worst (the float32 redo left 1.4e-5 on the k = 27 case of
    tools/stress_tile.py).  Against the float64 analysis of the same inputs: north-star tolerance."""
    rs = np.random.RandomState(k)
    G = 203
    grid, obs = np.arange(G, dtype=np.float64), np.arange(0, G, 1.0) + 0.13
    state = rs.normal(size=(2, k, G)) * sx
    hx = rs.normal(size=(k, len(obs))) * 0.7 * sy
    yb, d = hx - hx.mean(axis=0), rs.normal(size=len(obs)) * 0.7 * sy
    X, Yb, D = dev(state), dev(yb), dev(d)
    nb = eng.localize(grid, obs, [c])
    assert nb.p_max <= k
    ref = eng.analysis(X.double(), Yb.double(), D.double(), nb, 1.1, method="eig").cpu().numpy()
    tiles = eng.localize_tiles(grid, obs, [c], nb.p_max, extra_blocks=1)
    assert tiles.stats.tolist()[1] == 0
    xa, fl, retry = eng.analysis_tiles(X, eng.pack_split(Yb, D), len(obs), tiles, 1.1)
    assert int(retry.item()) > G // 2                         # (declined: the spectrum needs a degree beyond the cap)
    eng.retry_points(X, Yb, D, nb, 1.1, xa, fl)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    # the same through the round-2 kernel on per-point lists (same redo)
    xo = eng.analysis(X, Yb, D, nb, 1.1, method="matfun")
    assert rel_fro(xo.cpu().numpy(), ref) < TOL32


WCASES = [(40, 2, 10.0, 1), (10, 1, 1.6, 1), (24, 2, 6.5, 2), (64, 2, 12.0, 1), (33, 2, 3.0, 1), (20, 4, 18.0, 1), (96, 3, 16.0, 1),
          (16, 2, 7.0, 1)]


@pytest.mark.parametrize("k,stride,c,m", WCASES)
def test_weights_on_the_tile_route_vs_oracle(eng, k, stride, c, m):
    """LETKF.estimate_weights semantics (letkf.py:127-146: W[g, i, j] = w_mean_i + W_pert_ij) from tile lists and split
    records (mia_letkf_weights_tiles_f32, csrc/letkf_tile2w.hip): unions of one and two sixteen-row blocks, one to six member
    blocks, ensemble sizes that are not multiples of 8 or 16, ragged last tile.  Weights and analysis against the oracle; the
    analysis equals the plain tile launch bit for bit; applying the weights reproduces the analysis."""
    case = O.synthetic_case(203, k, stride, seed=3 * k + m, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    if tiles.stats.tolist()[1]:
        tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
    assert tiles.stats.tolist() == [nb.p_max, 0] and tiles.ut <= 2
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    P = case["yb"].shape[1]
    X = dev(case["state"])
    for inf in (1.0, 1.1):
        xa, W, fl, retry = eng.weights_tiles(X, rec, P, tiles, inf)
        assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0
        ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, inf)
        Wn = W.cpu().numpy()
        assert rel_fro(Wn, ref_w) < TOL32
        assert rel_fro(Wn - np.eye(k), ref_w - np.eye(k)) < 5e-5
        assert rel_fro(xa.cpu().numpy(), ref_xa) < TOL32
        xa2 = eng.analysis_tiles(X, rec, P, tiles, inf)[0]
        assert torch.equal(xa, xa2)
        mean = X.mean(dim=1, keepdim=True)
        applied = mean + torch.einsum("mig,gij->mjg", X - mean, W)
        assert rel_fro(applied.cpu().numpy(), ref_xa) < TOL32


def test_weights_on_tiles_declined_points_and_mixed_magnitudes(eng):
    """Declined points keep MIA_FLAG_RETRY and are redone with weights by the eigensolver kernel (mia_letkf_weights_retry_f32);
    observation types of very different magnitudes inside one tile keep the weights within the tolerance."""
    case = O.synthetic_case(203, 40, 2, seed=77)
    c = 10.0
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    P = case["yb"].shape[1]
    X = dev(case["state"])
    # (a) strong observations on part of the domain: those points are declined
    sc = np.where(case["obs_x"] < 80, 14.0, 1.0)
    yb, d = case["yb"] * sc, case["d"] * sc
    rec = eng.pack_split(dev(yb), dev(d))
    xa, W, fl, retry = eng.weights_tiles(X, rec, P, tiles, 1.1)
    n_retry = int(retry.item())
    f = fl.cpu().numpy()
    assert 0 < n_retry < 203 and ((f & 8) != 0).sum() == n_retry
    eng.weights_retry(X, dev(yb), dev(d), nb, 1.1, xa, W, fl)
    ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, c, 1.1)
    assert rel_fro(W.cpu().numpy(), ref_w) < TOL32 and rel_fro(xa.cpu().numpy(), ref_xa) < TOL32
    # (b) two observation types, normalised magnitudes 1e-4 and 2
    s = np.where(np.arange(P) % 2 == 0, 1e-4, 2.0)
    yb, d = case["yb"] * s, case["d"] * s
    rec = eng.pack_split(dev(yb), dev(d))
    xa, W, fl, retry = eng.weights_tiles(X, rec, P, tiles, 1.1)
    assert int(retry.item()) == 0
    ref_xa, ref_w = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, c, 1.1)
    assert rel_fro(W.cpu().numpy(), ref_w) < TOL32 and rel_fro(W.cpu().numpy() - np.eye(40), ref_w - np.eye(40)) < 5e-5


@pytest.mark.parametrize("k,stride,c", [(80, 1, 16.5), (40, 1, 9.0), (64, 1, 14.0), (96, 2, 30.0), (33, 1, 7.5)])
def test_two_waves_per_tile_equal_one(eng, k, stride, c):
    """Unions of more than 32 slots, one state row: the two-waves-per-tile kernel (csrc/letkf_tile2p.hip, option tile_pair) against
    letkf_tile2_kernel on the same lists and records -- equal to rounding (x' w_mean is summed in two parts), same flags and
    degrees -- and against the oracle; ragged last tile, three to six row blocks."""
    case = O.synthetic_case(203, k, stride, seed=5 * k)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max)
    if tiles.stats.tolist()[1]:
        tiles = eng.localize_tiles(case["grid_x"], case["obs_x"], [c], nb.p_max, extra_blocks=1)
    assert tiles.stats.tolist()[1] == 0 and tiles.ut >= 3
    rec = eng.pack_split(dev(case["yb"]), dev(case["d"]))
    P = case["yb"].shape[1]
    X = dev(case["state"])
    set_option("tile_pair", 0)
    xa1, fl1, r1 = eng.analysis_tiles(X, rec, P, tiles, 1.1)
    set_option("tile_pair", 1)
    xa2, fl2, r2 = eng.analysis_tiles(X, rec, P, tiles, 1.1)
    assert torch.equal(fl1, fl2) and int(r1.item()) == int(r2.item())
    ok = ((fl1 & 0xff) == 0)
    assert float(torch.linalg.norm(xa1[:, :, ok] - xa2[:, :, ok]) / torch.linalg.norm(xa1[:, :, ok])) < 1e-6
    assert torch.equal(xa2, eng.analysis_tiles(X, rec, P, tiles, 1.1)[0])
    if int(r2.item()):
        eng.retry_points(X, dev(case["yb"]), dev(case["d"]), nb, 1.1, xa2, fl2)
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1)
    assert rel_fro(xa2.cpu().numpy(), ref) < TOL32
