"""GPU parity for SURVEY.md section 8 row f3: the (localised) IEnKS weight update (core/ienks.py:108-141,
interface/lienks.py:75-118) and the ensemble transform with per-grid-point weights (base.py:257-278), through the
C ABI (mia_lienks_update_*, mia_apply_local_weights_*), against golden vectors produced by the reference's own
IEnKSTransformModule / IEnKSBundleModule (tests/golden/g9_ienks.npz) and against the CPU oracle.

Tolerances: float64 <= 1e-9 on weights (two matrix decompositions per update, three chained updates),
float32 <= 1e-5 relative Frobenius error on the analysis ensemble (BASELINE.json north_star), <= 1e-4 on raw
weights after three chained float32 updates.
"""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mia():
    import torch_assimilate_amd as m
    m.build()
    return m


@pytest.fixture(scope="module")
def eng(mia):
    return mia.LetkfEngine("cuda:0")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)])
def test_ienks_core_modules_vs_reference(mia, golden, dtype, tol):
    g = golden("g9_ienks.npz")
    for bi, (k, p) in enumerate(g["blocks"]):
        yb, d = g[f"yb_{bi}"], g[f"d_{bi}"]
        for tau in (1.0, 0.7):
            ttag = str(tau).replace(".", "p")
            for vname, mod, scale in (("transform", mia.IEnKSTransformModule(tau), 1.0),
                                      ("bundle", mia.IEnKSBundleModule(1e-4, tau), 1e-4)):
                w = torch.eye(int(k), dtype=dtype)
                for it in range(3):
                    w = mod(w, torch.tensor(yb * scale, dtype=dtype), torch.tensor(d, dtype=dtype))
                    assert w.dtype == dtype
                    assert rel_fro(w.cpu().numpy(), g[f"{vname}_{bi}_{ttag}_it{it}"]) < tol, (vname, bi, tau, it)
                got = mod(torch.tensor(g[f"w0_{bi}"], dtype=dtype), torch.tensor(yb * scale, dtype=dtype),
                          torch.tensor(d, dtype=dtype))
                assert rel_fro(got.cpu().numpy(), g[f"{vname}_{bi}_{ttag}_general"]) < tol, (vname, bi, tau)


def test_ienks_module_contract(mia, golden):
    g = golden("g9_ienks.npz")
    mod = mia.IEnKSTransformModule(1.0)
    w0 = torch.tensor(g["w0_4"])
    out = mod(w0, torch.zeros((40, 0), dtype=torch.float64), torch.zeros(0, dtype=torch.float64))
    np.testing.assert_array_equal(out.cpu().numpy(), g["w0_4"])             # core/ienks.py:135
    with pytest.raises(ValueError):                                          # core/base.py:28-39
        mod(torch.eye(10), torch.ones(10, 4), torch.ones(3))
    # first transform update from the prior weights with tau = 1 == the ETKF weights
    yb, d = torch.tensor(g["yb_1"]), torch.tensor(g["d_1"])
    w = mod(torch.eye(40, dtype=torch.float64), yb, d).cpu().numpy()
    assert rel_fro(w, mia.ETKFModule(1.0)(yb, d).cpu().numpy()) < 1e-9
    assert str(mod) == "TransformModule(tau=1.0)" and repr(mia.IEnKSBundleModule(1e-4, 0.5)) == "IEnKSBundle(0.0001, 0.5)"


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-5)])
def test_localised_ienks_vs_reference(mia, golden, dtype, tol):
    g = golden("g9_ienks.npz")
    st, gx, ox, yb, d = g["loc_state"], g["loc_grid_x"], g["loc_obs_x"], g["loc_yb"], g["loc_d"]
    loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())
    for vname, a, scale in (("transform", mia.LocalizedIEnKSTransform(None, loc, tau=0.8, dtype=dtype), 1.0),
                            ("bundle", mia.LocalizedIEnKSBundle(None, loc, tau=1.0, epsilon=1e-3, dtype=dtype), 1e-3)):
        w = a.generate_prior_weights(40)
        for it in range(2):
            w = a.inner_loop_arrays(w, yb * scale, d, grid_coords=gx, obs_coords=ox)
            assert tuple(w.shape) == (64, 40, 40)
            assert rel_fro(w.cpu().numpy()[::8], g[f"loc_{vname}_it{it}_weights"]) < 10 * tol, (vname, it)
        xa = a.apply_weights_arrays(st, w)
        assert rel_fro(xa.cpu().numpy(), g[f"loc_{vname}_analysis"]) < tol, vname


@pytest.mark.parametrize("m,k,G,dtype,tol", [(3, 40, 1000, torch.float64, 1e-14), (1, 7, 333, torch.float32, 1e-6),
                                             (2, 80, 257, torch.float32, 1e-6),
                                             # the tile kernel of csrc/apply_local.hip: every member-block count, ragged tiles, row
                                             # chunks of sixteen with a remainder, k not a multiple of four
                                             (16, 40, 1000, torch.float32, 1e-6), (33, 40, 999, torch.float32, 1e-6),
                                             (5, 96, 300, torch.float32, 1e-6), (20, 48, 517, torch.float32, 1e-6),
                                             (2, 3, 211, torch.float32, 1e-6), (70, 17, 403, torch.float32, 1e-6),
                                             (9, 64, 256, torch.float32, 1e-6), (4, 100, 250, torch.float32, 1e-6)])
def test_apply_local_weights_vs_oracle(eng, m, k, G, dtype, tol):
    rs = np.random.RandomState(k)
    X, W = rs.normal(size=(m, k, G)), rs.normal(size=(G, k, k)) / np.sqrt(k)
    X[0] += 300.0                                # (a variable with a large mean: the transform works on perturbations)
    got = eng.apply_local_weights(torch.tensor(X, dtype=dtype), torch.tensor(W, dtype=dtype))
    assert rel_fro(got.cpu().numpy(), O.apply_weights(X, W)) < tol
    assert rel_fro(got.cpu().numpy()[0] - 300.0, O.apply_weights(X, W)[0] - 300.0) < 100 * tol
    sub = eng.apply_local_weights(torch.tensor(X, dtype=dtype), torch.tensor(W[100:200], dtype=dtype), 100, 200)
    np.testing.assert_array_equal(sub.cpu().numpy(), got.cpu().numpy()[:, :, 100:200])
    with pytest.raises(ValueError):
        eng.apply_local_weights(torch.tensor(X, dtype=dtype), torch.tensor(W[:5], dtype=dtype))


def test_update_state_loop_vs_oracle(mia):
    """VarAssimilation.update_state (variational.py:107-135) with a toy linear model and an identity observation
    operator on every second grid point: the whole loop against the same loop run with the oracle."""
    G, k, s, c = 96, 20, 2, 6.0
    rs = np.random.RandomState(21)
    state = rs.normal(size=(1, k, G))
    ox = np.arange(0, G, s, dtype=np.float64)
    y = rs.normal(size=ox.shape[0])
    var = np.full(ox.shape[0], 0.5)
    gx = np.arange(G, dtype=np.float64)

    def model_np(x):
        return 0.9 * x + 0.1 * np.roll(x, 1, axis=-1)

    for vname, eps, tau in (("transform", None, 0.9), ("bundle", 1e-3, 1.0)):
        # --- oracle loop
        w = np.eye(k)
        for it in range(3):
            if eps is None:
                mw = np.broadcast_to(w, (G, k, k)) if w.ndim == 2 else w
            else:
                wm = (np.broadcast_to(w, (G, k, k)) if w.ndim == 2 else w).mean(axis=-1, keepdims=True)
                mw = eps * np.eye(k) + wm
            pseudo = model_np(O.apply_weights(state, mw))
            yb, d = O.obs_space_uncorr(pseudo[0][:, ::s], y, var)
            w = O.lienks_weights(w, gx, ox, yb, d, c, tau, eps)
        ref = O.apply_weights(state, w)
        # --- product loop (float64 end to end)
        def forward_model(x, it):
            nxt = 0.9 * x + 0.1 * torch.roll(x, 1, dims=-1)
            return nxt, nxt

        def observe(pseudo):
            return [pseudo[0][:, ::s]], [torch.tensor(y)], [torch.tensor(var)], None

        cls = mia.LocalizedIEnKSTransform if eps is None else mia.LocalizedIEnKSBundle
        kw = dict(tau=tau, max_iter=3, dtype=torch.float64)
        if eps is not None:
            kw["epsilon"] = eps
        a = cls(forward_model, mia.GaspariCohn(c, mia.AbsoluteDistance()), **kw)
        xa = a.update_state_arrays(state, observe, grid_coords=gx, obs_coords=ox)
        assert rel_fro(xa.cpu().numpy(), ref) < 1e-8, vname


def test_ienks_argument_errors_and_overflow_flag(mia, eng):
    case = O.synthetic_case(64, 10, 2)
    nb = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    yb, d = torch.tensor(case["yb"]), torch.tensor(case["d"])
    with pytest.raises(ValueError):
        eng.ienks_update(torch.eye(10, dtype=torch.float64), yb, d, nb, tau=1.5)
    with pytest.raises(ValueError):
        eng.ienks_update(torch.eye(10, dtype=torch.float64), yb, d, nb, epsilon=0.0)
    with pytest.raises(ValueError):
        eng.ienks_update(torch.eye(9, dtype=torch.float64), yb, d, nb)
    with pytest.raises(ValueError):
        mia.LocalizedIEnKSTransform(None, tau=-0.1)
    small = mia.NeighbourLists(nb.cnt, nb.idx, nb.w, nb.p_cap, 3, nb.g0, nb.g1)      # lie about p_max
    w, flags = eng.ienks_update(torch.eye(10, dtype=torch.float64), yb, d, small, return_flags=True)
    f = flags.cpu().numpy() & 0xff
    over = nb.cnt.cpu().numpy() > 3
    assert over.any() and np.all(f[over] == 1) and torch.isnan(w[torch.tensor(over)]).all()


def test_weight_save_path_streams_weights_through_disk(mia, golden, tmp_path):
    """filter.py:157-164 with a weight_save_path: estimate -> store (device -> pinned -> netCDF) -> load -> apply;
    the analysis equals the fused route's, the file holds the reference's weights."""
    from torch_assimilate_amd import weights_io as io
    g = golden("g7_synthetic_configs.npz")
    st, gx, ox, yb, d = (g["c2_" + n] for n in ("state", "grid_x", "obs_x", "yb", "d"))
    path = str(tmp_path / "w.nc")
    loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())
    a = mia.LETKF(localization=loc, inf_factor=1.1, dtype=torch.float64, weight_save_path=path)
    xa = a.analyse_arrays(st, yb, d, grid_coords=gx, obs_coords=ox)
    assert rel_fro(xa.cpu().numpy(), g["c2_1p1_analysis"]) < 1e-10
    W, coords = io.load_weights(path)
    assert tuple(W.shape) == (256, 40, 40) and coords["grid"].tolist() == list(range(256))
    assert rel_fro(W.numpy()[g["c2_widx"]], g["c2_1p1_weights"]) < 1e-10
    # several staging chunks, ragged tail, float32 weights
    Wd = torch.randn(1000, 12, 12, device="cuda:0")
    io.store_weights(path, Wd, chunk_points=96)
    back, _ = io.load_weights(path, "cuda:0", torch.float32, chunk_points=130)
    assert back.is_cuda and torch.equal(back, Wd)
    # global ETKF: (k, k) weights through the same path
    e = mia.ETKF(inf_factor=1.1, dtype=torch.float64, weight_save_path=path)
    xg = e.analyse_arrays(g["c1_state"], g["c1_yb"], g["c1_d"])
    assert rel_fro(xg.cpu().numpy(), g["c1_1p1_analysis"]) < 1e-10


@pytest.mark.parametrize("m,k,G,dtype,tol", [(3, 40, 1000, torch.float64, 1e-14), (1, 20, 40, torch.float32, 1e-6),
                                             (16, 40, 1003, torch.float32, 1e-6), (5, 96, 300, torch.float32, 1e-6),
                                             (2, 3, 211, torch.float32, 1e-6), (7, 17, 403, torch.float32, 1e-6),
                                             (4, 64, 256, torch.float32, 1e-6), (3, 100, 250, torch.float32, 1e-6)])
def test_apply_global_weights_vs_oracle(eng, m, k, G, dtype, tol):
    """_apply_weights with ONE weight matrix (the global ETKF's transform, base.py:257-278): the grid-points-as-columns kernel of
    csrc/apply_local.hip for float32 and k <= 96, the round-1 kernel otherwise; sub-ranges of the grid, a variable with a large mean."""
    rs = np.random.RandomState(100 + k)
    X, W = rs.normal(size=(m, k, G)), rs.normal(size=(k, k)) / np.sqrt(k)
    X[0] += 300.0
    got = eng.apply_weights(torch.tensor(X, dtype=dtype), torch.tensor(W, dtype=dtype))
    ref = O.apply_weights(X, W)
    assert rel_fro(got.cpu().numpy(), ref) < tol
    assert rel_fro(got.cpu().numpy()[0] - 300.0, ref[0] - 300.0) < 100 * tol
    if G > 200:
        sub = eng.apply_weights(torch.tensor(X, dtype=dtype), torch.tensor(W, dtype=dtype), 100, 200)
        np.testing.assert_array_equal(sub.cpu().numpy(), got.cpu().numpy()[:, :, 100:200])
