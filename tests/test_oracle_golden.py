"""Pins the CPU oracle (oracle/letkf_oracle.py) against vectors produced by the
reference itself (tools/gen_golden.py imports pytassim's core/localization/kernels/
wrapper modules) and against the closed-form answers of the reference's unit tests."""
import numpy as np
import pytest
import torch

from oracle import letkf_oracle as O
from conftest import rel_fro

TOL = 1e-12


def test_g1_known_answer(golden):
    g = golden("g1_known_answer.npz")
    w_mean, w_perts, pa = O.etkf_weights_parts(torch.tensor(g["yb"]), torch.tensor(g["d"]), 1.0)
    # closed forms quoted in tests/unit_tests/core/test_etkf.py:142-156,181-187
    np.testing.assert_allclose(pa.numpy(), [[0.75, 0.25], [0.25, 0.75]], atol=1e-12)
    np.testing.assert_allclose(w_mean.numpy().ravel(), [0.1, -0.1], atol=1e-12)
    np.testing.assert_allclose((w_perts @ w_perts.T).numpy(), pa.numpy(), atol=1e-12)
    np.testing.assert_allclose(w_mean.numpy(), g["w_mean"], atol=TOL)
    np.testing.assert_allclose(w_perts.numpy(), g["w_perts"], atol=TOL)
    np.testing.assert_allclose(pa.numpy(), g["pa"], atol=TOL)
    w = O.etkf_weights(g["yb"], g["d"], 1.0).numpy()
    np.testing.assert_allclose(w, g["weights"], atol=TOL)
    # mean_j(weights - I) = w_mean  (test_etkf.py:219-225)
    np.testing.assert_allclose((w - np.eye(2)).mean(axis=1), w_mean.numpy().ravel(), atol=1e-12)


def test_g2_prior_and_errors(golden):
    g = golden("g2_prior.npz")
    w = O.etkf_weights(np.ones((10, 0)), np.ones((1, 0)), 1.1).numpy()
    np.testing.assert_allclose(w, g["weights"], atol=TOL)
    np.testing.assert_allclose(w, np.sqrt(1.1) * np.eye(10), atol=1e-15)
    with pytest.raises(ValueError):
        O.etkf_weights(np.ones((10, 4)), np.ones((1, 3)))


def test_g3_g4_blocks(golden):
    g = golden("g3_g4_core_blocks.npz")
    for ci, (k, p) in enumerate(g["cases"]):
        yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
        for inf in (1.0, 1.1):
            tag = f"{ci}_{str(inf).replace('.', 'p')}"
            assert rel_fro(O.etkf_weights(yb, d, inf).numpy(), g[f"etkf_{tag}"]) < TOL
            for name, kern in (("rbf0p5", lambda x, y: O.rbf_kernel(x, y, 0.5)),
                               ("rbf10", lambda x, y: O.rbf_kernel(x, y, 10.0)),
                               ("gauss2", lambda x, y: O.rbf_kernel(x, y, 0.5 / 2.0 ** 2)),
                               ("linear", O.linear_kernel)):
                got = O.ketkf_weights(yb, d, kern, inf).numpy()
                assert rel_fro(got, g[f"ketkf_{name}_{tag}"]) < 1e-10, (name, tag)
            # linear-kernel KETKF == ETKF (tests/unit_tests/interface/test_lketkf.py:109-117)
            assert rel_fro(g[f"ketkf_linear_{tag}"], g[f"etkf_{tag}"]) < 1e-9
    x, y = torch.tensor(g["kern_x"]), torch.tensor(g["kern_y"])
    np.testing.assert_allclose(O.rbf_kernel(x, y, 0.5).numpy(), g["kern_rbf0p5"], atol=1e-14)
    np.testing.assert_allclose(O.rbf_kernel(x, y, 10.0).numpy(), g["kern_rbf10"], atol=1e-14)
    np.testing.assert_allclose(O.rbf_kernel(x, y, 0.125).numpy(), g["kern_gauss2"], atol=1e-14)
    np.testing.assert_allclose(O.linear_kernel(x, y).numpy(), g["kern_linear"], atol=1e-14)


def test_g5_gaspari_cohn(golden):
    g = golden("g5_gaspari_cohn.npz")
    r = g["r"]
    np.testing.assert_array_equal(O.gc_f1(r), g["f1"])
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(O.gc_f2(r[r > 0]), g["f2"])
    for c in (1.0, 10.0, 16.5):
        use, w = O.localize_obs(r * c, c)
        np.testing.assert_array_equal(use, g[f"use_c{c}"])
        np.testing.assert_array_equal(w, g[f"w_c{c}"])
    use, w = O.localize_obs(np.stack([g["dh"], g["dv"]]), (10.0, 1.5))
    np.testing.assert_array_equal(use, g["use_2r"])
    np.testing.assert_array_equal(w, g["w_2r"])
    use, w = O.localize_obs(r * 10.0, 10.0, epsilon=1e-3)
    np.testing.assert_array_equal(use, g["use_eps1e3"])
    # sanity values observed with the reference class
    _, w1 = O.localize_obs(np.array([0, .5, 1, 1.5, 1.8, 1.9, 2.0]), 1.0)
    np.testing.assert_allclose(w1, [1, 0.684896, 0.208333, 0.016493, 4.696e-4, 3.03e-5, 0.0], atol=2e-6)
    assert w1[-1] == 0.0


def test_g6_reference_fixture_letkf(golden):
    g = golden("g6_reference_fixture_letkf.npz")
    state, ti = g["state"], int(g["time_index"])
    yb, d = O.obs_space_corr(state[0, ti], g["obs"][ti], g["cov"])
    np.testing.assert_allclose(yb, g["yb"], atol=1e-13)
    np.testing.assert_allclose(d, g["d"], atol=1e-13)
    for inf in (1.0, 1.1):
        tag = str(inf).replace(".", "p")
        ana, w = O.letkf_analysis(state[:, [ti]], g["grid"], g["obs_grid"], yb, d, 10.0, inf)
        assert rel_fro(w, g[f"weights_{tag}"]) < TOL
        assert rel_fro(ana, g[f"analysis_{tag}"]) < TOL
        ana_g, w_g = O.etkf_analysis(state[:, [ti]], yb, d, inf)
        assert rel_fro(w_g, g[f"weights_global_{tag}"]) < TOL
        assert rel_fro(ana_g, g[f"analysis_global_{tag}"]) < TOL


@pytest.mark.parametrize("name,k,s,c,gamma", [("c2", 40, 2, 10.0, None), ("c4", 80, 1, 16.5, None),
                                              ("c5", 40, 2, 10.0, 0.5), ("c2m3", 40, 2, 10.0, None)])
def test_g7_synthetic(golden, name, k, s, c, gamma):
    g = golden("g7_synthetic_configs.npz")
    st = g[f"{name}_state"]
    G = st.shape[-1]
    case = O.synthetic_case(G, k, s, seed=42, m=st.shape[0])
    np.testing.assert_array_equal(case["state"], st)
    np.testing.assert_allclose(case["yb"], g[f"{name}_yb"], atol=1e-15)
    np.testing.assert_allclose(case["d"], g[f"{name}_d"], atol=1e-15)
    core = O.etkf_weights if gamma is None else (
        lambda a, b, inf: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, gamma), inf))
    for inf in (1.0, 1.1):
        tag = f"{name}_{str(inf).replace('.', 'p')}"
        ana, w = O.letkf_analysis(st, case["grid_x"], case["obs_x"], case["yb"], case["d"], c, inf, core=core)
        assert rel_fro(w[g[f"{name}_widx"]], g[f"{tag}_weights"]) < 1e-11
        assert rel_fro(ana, g[f"{tag}_analysis"]) < 1e-11


def test_g7_c1_global(golden):
    g = golden("g7_synthetic_configs.npz")
    st = g["c1_state"]
    for inf in (1.0, 1.1):
        tag = f"c1_{str(inf).replace('.', 'p')}"
        ana, w = O.etkf_analysis(st, g["c1_yb"], g["c1_d"], inf)
        assert rel_fro(w, g[f"{tag}_weights"][0]) < 1e-10
        assert rel_fro(ana, g[f"{tag}_analysis"]) < 1e-10


def test_g8_gaspari_cohn_inf(golden):
    g = golden("g8_kernels_gcinf.npz")
    r = g["r"]
    with np.errstate(all="ignore"):
        for i, fn in enumerate((O.gc_inf_f1, O.gc_inf_f2, O.gc_inf_f3, O.gc_inf_f4)):
            np.testing.assert_array_equal(fn(r[r > 0]), g[f"inf_f{i + 1}"])
    for c in (1.0, 10.0):
        use, w = O.localize_obs(r * c, c, taper="gc_inf")
        np.testing.assert_array_equal(use, g[f"inf_use_c{c}"])
        np.testing.assert_array_equal(w, g[f"inf_w_c{c}"])
    # C_0(0) = 1, continuity at the three inner knots, compact support
    w = O.gaspari_cohn_inf(np.array([0.0, 0.5 - 1e-12, 0.5, 1 - 1e-12, 1.0, 1.5 - 1e-12, 1.5, 2.0]))
    assert w[0] == 1.0 and w[-1] == 0.0
    np.testing.assert_allclose(w[1::2][:3], w[2::2][:3], atol=1e-10)


def test_g8_kernels_and_ketkf(golden):
    from kernel_cases import oracle_kernels
    g = golden("g8_kernels_gcinf.npz")
    x, y = torch.tensor(g["kern_x"]), torch.tensor(g["kern_y"])
    for name, kern in oracle_kernels().items():
        np.testing.assert_allclose(kern(x, x).numpy(), g[f"kxx_{name}"], rtol=1e-13, atol=1e-13, err_msg=name)
        np.testing.assert_allclose(kern(x, y).numpy(), g[f"kxy_{name}"], rtol=1e-13, atol=1e-13, err_msg=name)
        for bi in range(len(g["blocks"])):
            for inf in (1.0, 1.1):
                tag = f"{name}_{bi}_{str(inf).replace('.', 'p')}"
                got = O.ketkf_weights(g[f"yb_{bi}"], g[f"d_{bi}"], kern, inf).numpy()
                assert rel_fro(got, g[f"ketkf_{tag}"]) < 1e-10, tag


def test_g8_localised(golden):
    from kernel_cases import oracle_kernels
    g = golden("g8_kernels_gcinf.npz")
    st, gx, ox, yb, d = g["loc_state"], g["loc_grid_x"], g["loc_obs_x"], g["loc_yb"], g["loc_d"]
    kerns = oracle_kernels()
    for tag, core, taper in (
            ("lketkf_poly2", lambda a, b, inf: O.ketkf_weights(a, b, kerns["poly2"], inf), "gc"),
            ("lketkf_ornuhl", lambda a, b, inf: O.ketkf_weights(a, b, kerns["ornuhl"], inf), "gc"),
            ("letkf_gcinf", O.etkf_weights, "gc_inf")):
        ana, w = O.letkf_analysis(st, gx, ox, yb, d, 10.0, 1.1, core=core, taper=taper)
        assert rel_fro(w[::8], g[f"{tag}_weights"]) < 1e-10, tag
        assert rel_fro(ana, g[f"{tag}_analysis"]) < 1e-10, tag


def test_g9_ienks_update(golden):
    g = golden("g9_ienks.npz")
    for bi, (k, p) in enumerate(g["blocks"]):
        yb, d = g[f"yb_{bi}"], g[f"d_{bi}"]
        for tau in (1.0, 0.7):
            ttag = str(tau).replace(".", "p")
            for vname, eps in (("transform", None), ("bundle", 1e-4)):
                scale = 1.0 if eps is None else eps
                w = np.eye(k)
                for it in range(3):
                    w = O.ienks_update(w, yb * scale, d, tau, eps).numpy()
                    assert rel_fro(w, g[f"{vname}_{bi}_{ttag}_it{it}"]) < 1e-10, (vname, bi, tau, it)
                got = O.ienks_update(g[f"w0_{bi}"], yb * scale, d, tau, eps).numpy()
                assert rel_fro(got, g[f"{vname}_{bi}_{ttag}_general"]) < 1e-10
    # no observation: weights come back unchanged (core/ienks.py:135; tests/unit_tests/core/test_ienks.py)
    w0 = g["w0_4"]
    np.testing.assert_array_equal(O.ienks_update(w0, np.zeros((40, 0)), np.zeros(0)).numpy(), w0)
    # the first transform iteration from the prior weights with tau = 1 is the ETKF analysis
    yb, d = g["yb_1"], g["d_1"]
    assert rel_fro(O.ienks_update(np.eye(40), yb, d, 1.0).numpy(), O.etkf_weights(yb, d, 1.0).numpy()) < 1e-10


def test_g9_localised_ienks(golden):
    g = golden("g9_ienks.npz")
    st, gx, ox, yb, d = g["loc_state"], g["loc_grid_x"], g["loc_obs_x"], g["loc_yb"], g["loc_d"]
    for vname, tau, eps in (("transform", 0.8, None), ("bundle", 1.0, 1e-3)):
        w = np.eye(40)
        for it in range(2):
            w = O.lienks_weights(w, gx, ox, yb * (1.0 if eps is None else eps), d, 10.0, tau, eps)
            assert rel_fro(w[::8], g[f"loc_{vname}_it{it}_weights"]) < 1e-10, (vname, it)
        assert rel_fro(O.apply_weights(st, w), g[f"loc_{vname}_analysis"]) < 1e-10


def test_oracle_pool_equals_the_oracle_called_directly():
    """tests/oracle_pool.py (the worker pool behind the all-points GPU comparisons): windowed per-point evaluation in forked workers
    == letkf_oracle.letkf_analysis over all observations, ETKF and RBF-KETKF cores."""
    import oracle_pool
    from oracle import letkf_oracle as O
    case = O.synthetic_case(600, 12, 2)
    was = oracle_pool.started()
    oracle_pool.start()
    try:
        pts = np.arange(0, 600, 3)
        got = oracle_pool.oracle_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1, pts, chunk=37)
        ref, _ = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)
        np.testing.assert_allclose(got, ref[:, :, pts], rtol=0, atol=1e-12)
        core = lambda a, b, i: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, 0.5), i)      # noqa: E731
        gk = oracle_pool.oracle_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1, pts[:40], gamma=0.5)
        rk, _ = O.letkf_analysis(case["state"][:, :, pts[:40]], case["grid_x"][pts[:40]], case["obs_x"], case["yb"], case["d"], 10.0, 1.1, core=core)
        np.testing.assert_allclose(gk, rk, rtol=0, atol=1e-12)
        per, fro = oracle_pool.per_point_errors(got, ref[:, :, pts])
        assert per.shape == (len(pts),) and fro < 1e-12
    finally:
        if not was:
            oracle_pool.stop()
