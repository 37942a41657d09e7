"""GPU parity: the HIP path (through the C ABI) against the committed golden vectors that
were produced by the reference itself, and against the CPU oracle on seeded inputs.

Tolerances: float32 kernels <= 1e-5 relative Frobenius error on the analysis ensemble
(BASELINE.json north_star); float64 kernels <= 1e-10; localisation masks bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import rel_fro, set_option
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu

TOL32 = 1e-5
TOL64 = 1e-10


@pytest.fixture(scope="module")
def eng():
    import torch_assimilate_amd as mia
    mia.build()
    return mia.LetkfEngine("cuda:0")


def dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device="cuda:0", dtype=dtype)


def all_obs_lists(eng, n_pts, p):
    """every grid point sees every observation with weight 1 (distance 0)"""
    cap = max(p, 1)
    cand = np.tile(np.arange(cap, dtype=np.int32), (n_pts, 1))
    if p == 0:
        cand[:] = -1
    return eng.localize_from_dist(np.zeros((1, n_pts, cap)), cand, [1.0])


def test_gaspari_cohn_kernel(eng, golden):
    g = golden("g5_gaspari_cohn.npz")
    r = g["r"]
    w64 = eng.gaspari_cohn(dev(r, torch.float64)).cpu().numpy()
    np.testing.assert_allclose(w64, g["w_c1.0"], rtol=0, atol=4e-15)
    assert w64[r >= 2.0].max() == 0.0
    w32 = eng.gaspari_cohn(dev(r, torch.float32)).cpu().numpy()
    np.testing.assert_allclose(w32, g["w_c1.0"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("name,c", [("c2", 10.0), ("c4", 16.5)])
def test_localize_1d_matches_reference_mask(eng, golden, name, c):
    g = golden("g7_synthetic_configs.npz")
    gx, ox = g[f"{name}_grid_x"], g[f"{name}_obs_x"]
    nb = eng.localize(gx, ox, [c])
    cnt, idx, w = nb.cnt.cpu().numpy(), nb.idx.cpu().numpy(), nb.w.cpu().numpy()
    for gi in range(len(gx)):
        use, wt = O.localize_obs(O.abs_distance_1d(gx[gi], ox), c)
        ref = np.nonzero(use)[0]
        assert cnt[gi] == len(ref)
        got = idx[gi, :cnt[gi]]
        order = np.argsort(got)
        np.testing.assert_array_equal(got[order], ref)
        np.testing.assert_allclose(w[gi, :cnt[gi]][order] ** 2, wt[ref], rtol=1e-9, atol=1e-15)
        assert (idx[gi, cnt[gi]:] == -1).all()
    assert nb.p_max == cnt.max()


@pytest.mark.parametrize("nc,groups,radii", [(2, [0, 0], [0.08]), (3, [0, 0, 1], [0.15, 0.3]), (3, [0, 1, 2], [0.2, 0.1, 0.4])])
def test_localize_nd_matches_oracle(eng, nc, groups, radii):
    rs = np.random.RandomState(7)
    G, P = 300, 2000
    grid = rs.uniform(-0.1, 1.1, size=(G, nc))
    obs = rs.uniform(0, 1, size=(P, nc))
    nb = eng.localize(grid, obs, radii, coord_group=groups, p_cap=16)   # forces the retry path
    cnt, idx, w = nb.cnt.cpu().numpy(), nb.idx.cpu().numpy(), nb.w.cpu().numpy()
    for gi in range(G):
        dist = O.grouped_euclid_distance(grid[gi], obs, groups, len(radii))
        use, wt = O.localize_obs(dist, radii)
        ref = np.nonzero(use)[0]
        assert cnt[gi] == len(ref), gi
        got = idx[gi, :cnt[gi]]
        order = np.argsort(got)
        np.testing.assert_array_equal(got[order], ref)
        np.testing.assert_allclose(w[gi, :cnt[gi]][order] ** 2, wt[ref], rtol=1e-9, atol=1e-15)
    assert cnt.max() > 16


@pytest.mark.parametrize("dtype,tol", [(torch.float64, TOL64), (torch.float32, TOL32)])
def test_core_blocks_vs_reference(eng, golden, dtype, tol):
    """ETKFModule outputs of the reference for random (k, p) blocks, both eigen routes
    (p <= k dual, p > k primal), incl. the (40,20)/(40,19)/(80,64) shapes of the configs."""
    g = golden("g3_g4_core_blocks.npz")
    for ci, (k, p) in enumerate(g["cases"]):
        yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
        X = np.random.RandomState(ci).normal(size=(2, k, 1))
        nb = all_obs_lists(eng, 1, p)
        for inf in (1.0, 1.1):
            tag = f"{ci}_{str(inf).replace('.', 'p')}"
            xa, W, fl = eng.analysis(dev(X, dtype), dev(yb, dtype), dev(d, dtype), nb, inf,
                                     return_weights=True, return_flags=True)
            assert (int(fl.cpu()[0]) & 0xff) == 0
            ref_w = g[f"etkf_{tag}"]
            assert rel_fro(W.cpu().numpy()[0], ref_w) < tol, (k, p, inf)
            assert rel_fro(xa.cpu().numpy(), O.apply_weights(X, ref_w[None])) < tol, (k, p, inf)


def test_known_answer_and_prior(eng, golden):
    g = golden("g1_known_answer.npz")
    X = np.array([[[0.5], [-0.5]]])
    nb = all_obs_lists(eng, 1, 1)
    _, W = eng.analysis(dev(X, torch.float64), dev(g["yb"], torch.float64), dev(g["d"].ravel(), torch.float64),
                        nb, 1.0, return_weights=True)
    np.testing.assert_allclose(W.cpu().numpy()[0], g["weights"], atol=1e-12)
    # empty observation set -> sqrt(inf) * I  (tests/unit_tests/core/test_etkf.py:227-233)
    g2 = golden("g2_prior.npz")
    X = np.random.RandomState(0).normal(size=(1, 10, 5))
    nb = eng.localize(np.arange(5.0), np.zeros((0, 1)), [10.0])
    assert nb.p_max == 0
    for dtype in (torch.float32, torch.float64):
        xa, W = eng.analysis(dev(X, dtype), torch.zeros((10, 0)), torch.zeros(0), nb, 1.1, return_weights=True)
        np.testing.assert_allclose(W.cpu().numpy()[3], g2["weights"], atol=1e-6 if dtype == torch.float32 else 1e-14)
        assert rel_fro(xa.cpu().numpy(), O.apply_weights(X, g2["weights"])) < 1e-6
    with pytest.raises(ValueError):
        eng.analysis(dev(X, torch.float32), torch.ones((10, 4)), torch.ones(3), nb, 1.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, TOL64), (torch.float32, TOL32)])
def test_reference_fixture_letkf(eng, golden, dtype, tol):
    """End-to-end on the reference's own test data (tests/data/test_state.nc / test_single_obs.nc),
    the case of tests/unit_tests/interface/test_letkf.py:106-157."""
    g = golden("g6_reference_fixture_letkf.npz")
    ti = int(g["time_index"])
    X = g["state"][:, ti]                               # (2, 10, 40): m = n_var
    nb = eng.localize(g["grid"], g["obs_grid"], [10.0])
    for inf in (1.0, 1.1):
        tag = str(inf).replace(".", "p")
        xa, W = eng.analysis(dev(X, dtype), dev(g["yb"], dtype), dev(g["d"], dtype), nb, inf, return_weights=True)
        assert rel_fro(W.cpu().numpy(), g[f"weights_{tag}"]) < tol
        assert rel_fro(xa.cpu().numpy(), g[f"analysis_{tag}"][:, 0]) < tol
        Wg = eng.etkf_weights(dev(g["yb"], dtype), dev(g["d"], dtype), inf)
        assert rel_fro(Wg.cpu().numpy(), g[f"weights_global_{tag}"]) < tol
        xg = eng.apply_weights(dev(X, dtype), Wg)
        assert rel_fro(xg.cpu().numpy(), g[f"analysis_global_{tag}"][:, 0]) < tol


@pytest.mark.parametrize("name,c,gamma", [("c2", 10.0, None), ("c4", 16.5, None), ("c2m3", 10.0, None), ("c5", 10.0, 0.5)])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, TOL64), (torch.float32, TOL32)])
def test_scaled_configs_vs_reference(eng, golden, name, c, gamma, dtype, tol):
    g = golden("g7_synthetic_configs.npz")
    X = g[f"{name}_state"]
    nb = eng.localize(g[f"{name}_grid_x"], g[f"{name}_obs_x"], [c])
    for inf in (1.0, 1.1):
        tag = f"{name}_{str(inf).replace('.', 'p')}"
        xa, W, fl = eng.analysis(dev(X, dtype), dev(g[f"{name}_yb"], dtype), dev(g[f"{name}_d"], dtype), nb, inf,
                                 return_weights=True, rbf_gamma=gamma, return_flags=True)
        assert int((fl & 0xff).max().cpu()) == 0
        ref = g[f"{tag}_analysis"]
        err = rel_fro(xa.cpu().numpy(), ref)
        mean = ref.mean(axis=1, keepdims=True)
        err_inc = rel_fro(xa.cpu().numpy() - X.mean(axis=1, keepdims=True), ref - X.mean(axis=1, keepdims=True))
        werr = rel_fro(W.cpu().numpy()[g[f"{name}_widx"]], g[f"{tag}_weights"])
        print(f"{tag} {dtype}: analysis {err:.2e} increments {err_inc:.2e} weights {werr:.2e}")
        assert err < tol and werr < tol and err_inc < 10 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, TOL64), (torch.float32, TOL32)])
def test_c1_global_etkf(eng, golden, dtype, tol):
    g = golden("g7_synthetic_configs.npz")
    X = g["c1_state"]
    for inf in (1.0, 1.1):
        tag = f"c1_{str(inf).replace('.', 'p')}"
        W = eng.etkf_weights(dev(g["c1_yb"], dtype), dev(g["c1_d"], dtype), inf)
        assert rel_fro(W.cpu().numpy(), g[f"{tag}_weights"][0]) < tol
        xa = eng.apply_weights(dev(X, dtype), W)
        assert rel_fro(xa.cpu().numpy(), g[f"{tag}_analysis"]) < tol


def test_ketkf_blocks_vs_reference(eng, golden):
    g = golden("g3_g4_core_blocks.npz")
    for ci, (k, p) in enumerate(g["cases"]):
        yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
        X = np.random.RandomState(ci).normal(size=(1, k, 1))
        nb = all_obs_lists(eng, 1, p)
        for gname, gamma in (("rbf0p5", 0.5), ("rbf10", 10.0), ("gauss2", 0.125)):
            for dtype, tol in ((torch.float64, 1e-9), (torch.float32, TOL32)):
                _, W = eng.analysis(dev(X, dtype), dev(yb, dtype), dev(d, dtype), nb, 1.1,
                                    return_weights=True, rbf_gamma=gamma)
                assert rel_fro(W.cpu().numpy()[0], g[f"ketkf_{gname}_{ci}_1p1"]) < tol, (k, p, gname, dtype)


def test_overflow_is_flagged_not_truncated(eng, golden):
    g = golden("g7_synthetic_configs.npz")
    X = g["c2_state"]
    nb = eng.localize(g["c2_grid_x"], g["c2_obs_x"], [10.0])
    nb.p_max = 10                                   # lie about the bound
    xa, fl = eng.analysis(dev(X, torch.float32), dev(g["c2_yb"], torch.float32), dev(g["c2_d"], torch.float32),
                          nb, 1.0, return_flags=True)
    fl = fl.cpu().numpy()
    assert (fl[20:-20] & 1).all()
    assert np.isnan(xa.cpu().numpy()[0, :, 100]).all()


def test_large_grid_properties(eng):
    """BASELINE config-2 size: properties that need no oracle at scale -- analysis mean/perturbation
    structure (weights rows sum: W 1 = 1 when perturbations are centred), determinism, shard invariance."""
    case = O.synthetic_case(100000, 40, 2)
    X = dev(case["state"], torch.float32)
    yb, d = dev(case["yb"], torch.float32), dev(case["d"], torch.float32)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    assert nb.p_max == 20
    xa = eng.analysis(X, yb, d, nb, 1.1)
    xa2 = eng.analysis(X, yb, d, nb, 1.1)
    assert torch.equal(xa, xa2)
    xe = eng.analysis(X, yb, d, nb, 1.1, method="eig")           # the two routes agree at full size
    assert float(torch.linalg.norm(xa - xe) / torch.linalg.norm(xe)) < TOL32
    assert torch.isfinite(xa).all()
    # shard [30000, 30100) equals the same columns of the full run: to rounding per point on the split-precision products
    # (the last four points sit in a tile of their own, whose operand scale differs), bit for bit on the f32 products
    nb_s = eng.localize(case["grid_x"], case["obs_x"], [10.0], g0=30000, g1=30100)
    xs = eng.analysis(X, yb, d, nb_s, 1.1)
    part = xa[:, :, 30000:30100]
    assert float(((xs - part).norm(dim=(0, 1)) / part.norm(dim=(0, 1))).max()) < 2e-6
    set_option("tile_split", 0)
    xf = eng.analysis(X, yb, d, nb, 1.1)
    assert float(torch.linalg.norm(xf - xa) / torch.linalg.norm(xa)) < 1e-6
    assert torch.equal(eng.analysis(X, yb, d, nb_s, 1.1), xf[:, :, 30000:30100])
    set_option("tile_split", 1)
    # oracle on a random subset of points
    sel = np.random.RandomState(1).choice(100000, 64, replace=False)
    for gi in sel:
        dist = O.abs_distance_1d(case["grid_x"][gi], case["obs_x"])
        w = O.localized_weights(dist, case["yb"], case["d"], [10.0], 1.1)
        ref = O.apply_weights(case["state"][:, :, [gi]], w[None])
        assert rel_fro(xa[:, :, gi].cpu().numpy(), ref[:, :, 0]) < TOL32


# ---------------------------------------------------------------- eigensolver-free (matfun) route
@pytest.mark.parametrize("name,c,gamma", [("c2", 10.0, None), ("c4", 16.5, None), ("c2m3", 10.0, None), ("c5", 10.0, 0.5)])
def test_matfun_route_vs_reference(eng, golden, name, c, gamma):
    """The Chebyshev matrix-function route against the reference-generated analysis (same gate as the
    eigensolver route: 1e-5 relative Frobenius, increments within 1e-4) and against the eigensolver route."""
    g = golden("g7_synthetic_configs.npz")
    X = g[f"{name}_state"]
    nb = eng.localize(g[f"{name}_grid_x"], g[f"{name}_obs_x"], [c])
    xm = X.mean(axis=1, keepdims=True)
    for inf in (1.0, 1.1):
        tag = f"{name}_{str(inf).replace('.', 'p')}"
        args = (dev(X, torch.float32), dev(g[f"{name}_yb"], torch.float32), dev(g[f"{name}_d"], torch.float32), nb, inf)
        xa, fl = eng.analysis(*args, rbf_gamma=gamma, return_flags=True, method="matfun")
        xe = eng.analysis(*args, rbf_gamma=gamma, method="eig")
        f = fl.cpu().numpy()
        assert int((f & 0xff).max()) == 0
        ref = g[f"{tag}_analysis"]
        assert rel_fro(xa.cpu().numpy(), ref) < TOL32
        assert rel_fro(xa.cpu().numpy() - xm, ref - xm) < 10 * TOL32
        assert rel_fro(xa.cpu().numpy(), xe.cpu().numpy()) < TOL32


def test_matfun_blocks_and_edge_cases(eng, golden):
    """Random (k, p) blocks of both routes (dual p <= k, primal p > k), the empty observation set and the
    weights request (which must refuse the matfun route)."""
    g = golden("g3_g4_core_blocks.npz")
    for ci, (k, p) in enumerate(g["cases"]):
        if k > 128 or min(k, p) > 64:
            continue
        yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
        X = np.random.RandomState(ci).normal(size=(2, k, 1))
        nb = all_obs_lists(eng, 1, p)
        xa, fl = eng.analysis(dev(X, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1,
                              return_flags=True, method="matfun")
        assert (int(fl.cpu()[0]) & 0xff) == 0
        assert rel_fro(xa.cpu().numpy(), O.apply_weights(X, g[f"etkf_{ci}_1p1"][None])) < TOL32, (k, p)
    X = np.random.RandomState(0).normal(size=(1, 10, 5))
    nb = eng.localize(np.arange(5.0), np.zeros((0, 1)), [10.0])
    xa = eng.analysis(dev(X, torch.float32), torch.zeros((10, 0)), torch.zeros(0), nb, 1.1, method="matfun")
    assert rel_fro(xa.cpu().numpy(), O.apply_weights(X, np.sqrt(1.1) * np.eye(10))) < 1e-6
    # the weights come from the eigensolver-free route as well (mia_letkf_weights_matfun_f32); no observation: prior
    xa, W = eng.analysis(dev(X, torch.float32), torch.zeros((10, 0)), torch.zeros(0), nb, 1.1, method="matfun",
                         return_weights=True)
    np.testing.assert_allclose(W.cpu().numpy()[2], np.sqrt(1.1) * np.eye(10), atol=1e-6)
    assert rel_fro(xa.cpu().numpy(), O.apply_weights(X, np.sqrt(1.1) * np.eye(10))) < 1e-6
    with pytest.raises(ValueError):
        eng.analysis(dev(X, torch.float64), torch.zeros((10, 0)), torch.zeros(0), nb, 1.1, method="matfun")


def test_matfun_declines_wide_spectra_and_eigensolver_redoes_them(eng, golden, monkeypatch):
    """Strong observations (large lambda_max / reg) exceed the degree cap: those grid points are flagged
    MIA_FLAG_RETRY by the matfun kernel and redone by the eigensolver kernel; the result is the oracle's."""
    case = O.synthetic_case(300, 40, 2)
    yb, d = case["yb"] * 12.0, case["d"] * 12.0          # obs error variance / 144: kappa ~ 1e3
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    args = (dev(case["state"], torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1)
    xa, fl, finish = eng.analysis(*args, return_flags=True, method="matfun", defer_retry=True)
    f_before = fl.cpu().numpy().copy()
    n_retry = finish()
    assert n_retry == int(((f_before & 8) != 0).sum()) and n_retry > 100
    f_after = fl.cpu().numpy()
    assert int((f_after & 0xff).max()) == 0                # every declined point was redone
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, 10.0, 1.1)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    xm = case["state"].mean(axis=1, keepdims=True)
    assert rel_fro(xa.cpu().numpy() - xm, ref - xm) < 10 * TOL32
    # a moderate case mixes both kernels inside one shard
    set_option("cheb_dmax", 14)
    yb2, d2 = case["yb"], case["d"]
    xa2, fl2, fin2 = eng.analysis(dev(case["state"], torch.float32), dev(yb2, torch.float32), dev(d2, torch.float32),
                                  nb, 1.1, return_flags=True, method="matfun", defer_retry=True)
    n2 = fin2()
    assert 0 < n2 < 300
    ref2, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb2, d2, 10.0, 1.1)
    assert rel_fro(xa2.cpu().numpy(), ref2) < TOL32


def test_fused_localisation_equals_list_route_bitwise(eng, golden, monkeypatch):
    """The kernel that scans the observation index itself must reproduce the explicit-list route of the SAME per-point
    kernel bit for bit (same scan code, same order; option tile = 0 keeps the list route off the sixteen-point kernel, whose
    summation order differs), in 1-D and with two radii in 3-D; an under-estimated list bound is reported."""
    set_option("tile", 0)
    rs = np.random.RandomState(11)
    cases = [(np.arange(3000.0)[:, None], np.arange(0, 3000, 2.0)[:, None], [10.0], [0]),
             (rs.uniform(0, 1, size=(1500, 3)), rs.uniform(0, 1, size=(4000, 3)), [0.12, 0.25], [0, 0, 1])]
    for grid, obs, radii, groups in cases:
        k = 24
        P = obs.shape[0]
        X = dev(rs.normal(size=(2, k, grid.shape[0])), torch.float32)
        yb = rs.normal(size=(k, P)); yb -= yb.mean(axis=0)
        d = rs.normal(size=P)
        nb = eng.localize(grid, obs, radii, coord_group=groups)
        rec = eng.pack_obs(dev(yb, torch.float32), dev(d, torch.float32))
        ref = eng.analysis(X, None, None, nb, 1.1, rec=rec, method="matfun")
        index = eng.build_index(obs, radii, groups)
        xa, flags, finish = eng.analysis_fused(X, rec, grid, index, nb.p_max, 1.1)
        ok, p_max, n_retry = finish()
        assert ok and p_max == nb.p_max
        assert torch.equal(xa, ref)
        assert int((flags & 0xff).max().cpu()) == 0
        # shard of the same problem
        xs, _, fin = eng.analysis_fused(X, rec, grid, index, nb.p_max, 1.1, g0=100, g1=900)
        assert fin()[0] and torch.equal(xs, ref[:, :, 100:900])
        # assumed bound too small: reported, never silently truncated
        xb, fb, finb = eng.analysis_fused(X, rec, grid, index, max(nb.p_max - 3, 1), 1.1)
        okb, pb, _ = finb()
        assert not okb and pb == nb.p_max
        assert int((fb & 1).max().cpu()) == 1


@pytest.mark.parametrize("scale", [0.01, 1.0, 3.0, 6.0])
@pytest.mark.parametrize("inf", [0.8, 1.0, 1.5])
def test_both_routes_across_observation_strength_and_inflation(eng, scale, inf):
    """Spectra from nearly zero (weak observations: analysis ~ inflated prior) to lambda_max/reg ~ 100 (strong
    observations: matfun declines, eigensolver takes over), deflation (inf < 1) and inflation, both routes
    against the oracle on every grid point incl. the ragged domain edges."""
    case = O.synthetic_case(160, 24, 2, seed=5)
    yb, d = case["yb"] * scale, case["d"] * scale
    nb = eng.localize(case["grid_x"], case["obs_x"], [6.0])
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, 6.0, inf)
    xm = case["state"].mean(axis=1, keepdims=True)
    for method in ("matfun", "eig"):
        xa, fl = eng.analysis(dev(case["state"], torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, inf,
                              return_flags=True, method=method)
        assert int((fl & 0xff).max().cpu()) == 0
        assert rel_fro(xa.cpu().numpy(), ref) < TOL32, (method, scale, inf)
        assert rel_fro(xa.cpu().numpy() - xm, ref - xm) < 20 * TOL32, (method, scale, inf)


def test_degenerate_ensembles(eng):
    """All members equal in observation space (S = 0) and a single member perturbed: no division by zero, the
    result is the oracle's (prior perturbations inflated by sqrt(inf), mean shifted by nothing / by the gain)."""
    rs = np.random.RandomState(2)
    G, k = 64, 16
    state = rs.normal(size=(1, k, G))
    grid_x = np.arange(G, dtype=np.float64); obs_x = np.arange(0, G, 2, dtype=np.float64)
    yb0 = np.zeros((k, obs_x.size)); d0 = rs.normal(size=obs_x.size)
    yb1 = yb0.copy(); yb1[3] = 1.0; yb1 -= yb1.mean(axis=0)
    nb = eng.localize(grid_x, obs_x, [5.0])
    for yb in (yb0, yb1):
        ref, _ = O.letkf_analysis(state, grid_x, obs_x, yb, d0, 5.0, 1.2)
        for method in ("matfun", "eig"):
            xa = eng.analysis(dev(state, torch.float32), dev(yb, torch.float32), dev(d0, torch.float32), nb, 1.2, method=method)
            assert torch.isfinite(xa).all()
            assert rel_fro(xa.cpu().numpy(), ref) < TOL32


@pytest.mark.parametrize("m", [8, 19, 40])
@pytest.mark.parametrize("k,stride,c", [(40, 2, 10.0), (24, 1, 6.0), (10, 3, 7.0), (40, 1, 30.0), (12, 1, 9.0)])
def test_matfun_many_state_rows_on_the_matrix_cores(eng, monkeypatch, m, k, stride, c):
    """m >= 8 state rows per grid point: the matfun kernel transforms them 16 at a time as one matrix recurrence on
    the MFMA units (letkf_cheb_rows_kernel).  Against the oracle (<= 1e-5, north star) and against the row-by-row
    path of the same kernel (option cheb_rowbatch = 0), incl. a ragged last batch, edge points with short lists, an
    ensemble size that is not a multiple of 4, and two primal geometries (more local observations than members)."""
    G = 300
    case = O.synthetic_case(G, k, stride, seed=17, m=m)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    X, yb, d = dev(case["state"], torch.float32), dev(case["yb"], torch.float32), dev(case["d"], torch.float32)
    xa, flags = eng.analysis(X, yb, d, nb, 1.1, method="matfun", return_flags=True)
    assert int((flags & 0xff).max().item()) == 0
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.1)
    got = xa.cpu().numpy()
    assert rel_fro(got, ref) < TOL32
    xm = case["state"].mean(axis=1, keepdims=True)
    assert rel_fro(got - xm, ref - xm) < 2e-5                      # increments: the stricter measure
    for mi in (0, m // 2, m - 1):                                  # every row on its own, not only on average
        assert rel_fro(got[mi], ref[mi]) < TOL32
    set_option("cheb_rowbatch", 0)
    xs = eng.analysis(X, yb, d, nb, 1.1, method="matfun")
    assert rel_fro(got, xs.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("name,c", [("c2", 10.0), ("c2m3", 10.0)])
def test_weights_without_eigensolver_vs_reference(eng, golden, monkeypatch, name, c):
    """estimate_weights semantics (letkf.py:145-146) through mia_letkf_weights_matfun_f32: phi(S) as a matrix from the
    Chebyshev recurrence on MFMA tiles.  Weights at the golden sample points and the analysis against the reference;
    strong observations make the kernel decline points, which the eigensolver kernel then redoes WITH weights."""
    g = golden("g7_synthetic_configs.npz")
    st, gx, ox, yb, d = (g[f"{name}_{n}"] for n in ("state", "grid_x", "obs_x", "yb", "d"))
    nb = eng.localize(gx, ox, [c])
    for inf in (1.0, 1.1):
        tag = f"{name}_{str(inf).replace('.', 'p')}"
        xa, W, fl = eng.analysis(dev(st, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, inf,
                                 return_weights=True, return_flags=True)
        f = fl.cpu().numpy()
        assert int((f & 0xff).max()) == 0 and int(((f >> 8) & 0xff).min()) >= 3       # bits 8-15: degree => matfun ran
        assert rel_fro(W.cpu().numpy()[g[f"{name}_widx"]], g[f"{tag}_weights"]) < TOL32
        assert rel_fro(xa.cpu().numpy(), g[f"{tag}_analysis"]) < TOL32
    # declined points: the eigensolver redoes them, weights included
    scale = 12.0
    xa, W, fl = eng.analysis(dev(st, torch.float32), dev(yb * scale, torch.float32), dev(d * scale, torch.float32), nb, 1.1,
                             return_weights=True, return_flags=True)
    ref_xa, ref_w = O.letkf_analysis(st, gx, ox, yb * scale, d * scale, c, 1.1)
    assert rel_fro(W.cpu().numpy(), ref_w) < 5e-5 and rel_fro(xa.cpu().numpy(), ref_xa) < TOL32
    set_option("cheb_dmax", 8)          # force declines on the plain case too
    xa2, W2 = eng.analysis(dev(st, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1, return_weights=True)
    assert rel_fro(W2.cpu().numpy()[g[f"{name}_widx"]], g[f"{name}_1p1_weights"]) < TOL32


@pytest.mark.parametrize("gamma", [None, 0.5])
def test_dense_local_networks_far_beyond_the_ensemble_size(eng, gamma):
    """p >> k (the primal route): 1200-3000 local observations per grid point for k = 24 members.  The member Gram is
    streamed from the packed records on the matrix cores, so no local block has to fit a workgroup's LDS (the staged
    block capped p at a few hundred).  float32 matfun route, the eigensolver route and the float64 kernel against the
    oracle; observation strength chosen so that the matfun kernel declines part of the points (eigensolver redo)."""
    rs = np.random.RandomState(31)
    k, G, P = 24, 24, 3000
    grid, obs = rs.uniform(0.3, 0.7, size=G), rs.uniform(0, 1, size=P)
    state = rs.normal(size=(2, k, G))
    hx = rs.normal(size=(k, P)) * 0.35
    yb, d = hx - hx.mean(axis=0), rs.normal(size=P) * 0.35
    nb = eng.localize(grid, obs, [0.25])
    assert nb.p_max > 1200
    core = O.etkf_weights if gamma is None else (lambda a, b, inf: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, gamma), inf))
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, 0.25, 1.1, core=core)
    kw = dict(rbf_gamma=gamma)
    x64 = eng.analysis(dev(state, torch.float64), dev(yb, torch.float64), dev(d, torch.float64), nb, 1.1, **kw)
    assert rel_fro(x64.cpu().numpy(), ref) < 1e-9
    for method in ("auto", "eig"):
        xa, fl = eng.analysis(dev(state, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1,
                              method=method, return_flags=True, **kw)
        assert int((fl & 0xff).max().item()) == 0
        assert rel_fro(xa.cpu().numpy(), ref) < TOL32, method
    if gamma is None:
        _, W = eng.analysis(dev(state, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1, return_weights=True)
        _, wref = O.letkf_analysis(state[:, :, :3], grid[:3], obs, yb, d, 0.25, 1.1)
        assert rel_fro(W.cpu().numpy()[:3], wref) < 5e-5


@pytest.mark.parametrize("k,gamma", [(80, None), (72, 0.5), (128, None)])
def test_large_ensembles_with_more_than_64_local_observations(eng, monkeypatch, k, gamma):
    """64 < k <= 128 with more local observations than one row per lane can hold (ensemble-space order > 64): the
    two-rows-per-lane primal kernel (letkf_cheb_big_kernel: Gram streamed from the records, S in LDS).  Against the
    oracle, and against the eigensolver route it replaces (option cheb_big = 0)."""
    rs = np.random.RandomState(k)
    G, P = 30, 600
    grid, obs = rs.uniform(0.3, 0.7, size=G), rs.uniform(0, 1, size=P)
    state = rs.normal(size=(2, k, G))
    hx = rs.normal(size=(k, P)) * 0.5
    yb, d = hx - hx.mean(axis=0), rs.normal(size=P) * 0.5
    nb = eng.localize(grid, obs, [0.12])
    assert nb.p_max > 64
    core = O.etkf_weights if gamma is None else (lambda a, b, inf: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, gamma), inf))
    ref, _ = O.letkf_analysis(state, grid, obs, yb, d, 0.12, 1.1, core=core)
    args = (dev(state, torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1)
    xa, fl = eng.analysis(*args, rbf_gamma=gamma, return_flags=True)
    f = fl.cpu().numpy()
    assert int((f & 0xff).max()) == 0 and int(((f >> 8) & 0xff).min()) >= 3       # degree recorded: the matfun kernel ran
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    set_option("cheb_big", 0)
    xe = eng.analysis(*args, rbf_gamma=gamma)
    assert rel_fro(xe.cpu().numpy(), ref) < TOL32


@pytest.mark.gpu
@pytest.mark.parametrize("scale,gamma", [(1.0, None), (4.0, None), (1.0, 0.5)])
def test_coefficient_table_equals_the_in_kernel_transform(eng, monkeypatch, scale, gamma):
    """The Chebyshev coefficients come from a per-device table over the scaled spectral bound (letkf_cheb.hip,
    cheb_coef_table); option cheb_table = 0 computes them in the analysis kernel as before.  Same analysis to float32
    rounding on the dual route, with stronger observations (higher degrees), and on the RBF (primal) route; both within
    the north-star tolerance of the oracle."""
    case = O.synthetic_case(400, 40, 2, seed=7)
    yb, d = case["yb"] * scale, case["d"] * scale
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    args = (dev(case["state"], torch.float32), dev(yb, torch.float32), dev(d, torch.float32), nb, 1.1)
    xa_tab, fl_tab = eng.analysis(*args, return_flags=True, method="matfun", rbf_gamma=gamma)
    set_option("cheb_table", 0)
    xa_ker, fl_ker = eng.analysis(*args, return_flags=True, method="matfun", rbf_gamma=gamma)
    set_option("cheb_table", 1)
    assert int((fl_tab.cpu() & 0xff).max()) == 0 and int((fl_ker.cpu() & 0xff).max()) == 0
    assert rel_fro(xa_tab.cpu().numpy(), xa_ker.cpu().numpy()) < 2e-6
    # the table's interval is the next grid point above the bound: at most one degree more than the exact interval
    dt, dk = (fl_tab.cpu().numpy() >> 8) & 0xff, (fl_ker.cpu().numpy() >> 8) & 0xff
    assert int((dt - dk).min()) >= 0 and int((dt - dk).max()) <= 1
    if gamma is None:
        ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], yb, d, 10.0, 1.1)
        assert rel_fro(xa_tab.cpu().numpy(), ref) < TOL32


@pytest.mark.parametrize("k,stride,c", [(40, 2, 10.0), (36, 2, 11.5), (44, 1, 7.5), (50, 2, 15.5), (72, 2, 9.0),
                                        (96, 3, 16.0), (128, 2, 12.0), (24, 2, 9.5)])
def test_dual_route_orders_20_to_32_in_the_compact_lds_layout(eng, k, stride, c):
    """Scalar-rows matfun kernel on the dual route with 17 .. 32 local observations (order buckets 20 / 24 / 32): list,
    zero row, S, coefficient tables, recurrence vector and x' share the storage of S (cheb_compact_layout), the Gram
    loop reads member pairs when k % 8 == 0 and single members otherwise, k > 64 keeps two members per lane (where the
    compact layout no longer fits and the plain one is used).  Three state rows; against the oracle."""
    case = O.synthetic_case(96, k, stride, seed=k, m=3)
    nb = eng.localize(case["grid_x"], case["obs_x"], [c])
    assert 16 < nb.p_max <= min(32, k)
    xa, fl = eng.analysis(dev(case["state"], torch.float32), dev(case["yb"], torch.float32), dev(case["d"], torch.float32),
                          nb, 1.2, return_flags=True, method="matfun")
    f = fl.cpu().numpy()
    assert int((f & 0xff).max()) == 0 and int(((f >> 8) & 0xff).min()) >= 3
    ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], c, 1.2)
    assert rel_fro(xa.cpu().numpy(), ref) < TOL32
    assert rel_fro(xa.cpu().numpy() - case["state"].mean(axis=1, keepdims=True), ref - case["state"].mean(axis=1, keepdims=True)) < 5e-5
