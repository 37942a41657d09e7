#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's hot-path modules.

Runs only in the build container (needs /root/reference); the reference never
travels to the GPU box, only the small .npz vectors this script writes do.

Recipe: the reference's package ``__init__`` files pull in
xarray (absent here), so empty package stubs with the right ``__path__`` are
registered and the five torch/numpy-only modules on the hot path are imported
directly; ``torch.symeig`` (removed from torch 2.x, used at
pytassim/core/utils.py:57) is mapped onto ``torch.linalg.eigh``.  No reference
file is modified or copied.

    python tools/gen_golden.py            # writes tests/golden/*.npz
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MIA_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for name, sub in [("pytassim", ""), ("pytassim.core", "core"), ("pytassim.kernels", "kernels"),
                      ("pytassim.localization", "localization"), ("pytassim.interface", "interface")]:
        m = types.ModuleType(name)
        m.__path__ = [f"{REF}/pytassim/{sub}".rstrip("/")]
        sys.modules[name] = m
    torch.symeig = lambda t, eigenvectors=True, upper=False: torch.linalg.eigh(
        t, UPLO="U" if upper else "L")
    mods = dict(
        etkf=importlib.import_module("pytassim.core.etkf"),
        ketkf=importlib.import_module("pytassim.core.ketkf"),
        gc=importlib.import_module("pytassim.localization.gaspari_cohn"),
        rbf=importlib.import_module("pytassim.kernels.rbf"),
        linear=importlib.import_module("pytassim.kernels.linear"),
        wrapper=importlib.import_module("pytassim.interface.wrapper"),
    )
    mods["ienks"] = importlib.import_module("pytassim.core.ienks")
    for name in ("polynomial", "tanh", "periodic", "rational", "orn_uhl", "scale", "diag", "base_kernels"):
        mods[name] = importlib.import_module("pytassim.kernels." + name)
    return mods


def t64(x):
    return torch.tensor(x, dtype=torch.float64)


def apply_weights(state, weights):
    # einsum form of interface/base.py:267-270 (xr.dot over 'ensemble'); xarray itself is absent here
    mean = state.mean(axis=-2, keepdims=True)
    perts = state - mean
    return mean + np.einsum("...ig,gij->...jg", perts, weights)


def main():
    os.makedirs(OUT, exist_ok=True)
    R = import_reference()
    ETKFModule = R["etkf"].ETKFModule
    KETKFModule = R["ketkf"].KETKFModule
    GaspariCohn = R["gc"].GaspariCohn
    RBFKernel, GaussKernel = R["rbf"].RBFKernel, R["rbf"].GaussKernel
    LinearKernel = R["linear"].LinearKernel
    wrapper_bridge, wrapper_localization = R["wrapper"].wrapper_bridge, R["wrapper"].wrapper_localization

    # ---- G1: the 2-member / 1-obs known-answer case (tests/unit_tests/core/test_etkf.py:47-103)
    hx = np.array([0.5, -0.5]).reshape(2, 1)
    innov = np.array([0.2 - 0.0])
    rcinv = 1.0 / np.sqrt(0.5)
    yb, d = hx * rcinv, (innov * rcinv).reshape(1, 1)
    mod = ETKFModule(t64(1.0))
    w_mean, w_perts, pa = mod._estimate_weights(t64(yb), t64(d))
    np.savez(os.path.join(OUT, "g1_known_answer.npz"), yb=yb, d=d, inf=1.0,
             w_mean=w_mean.numpy(), w_perts=w_perts.numpy(), pa=pa.numpy(),
             weights=mod(t64(yb), t64(d)).numpy())

    # ---- G2: empty observations -> prior (test_etkf.py:227-233)
    mod = ETKFModule(t64(1.1))
    w = mod(torch.ones(10, 0, dtype=torch.float64), torch.ones(1, 0, dtype=torch.float64))
    np.savez(os.path.join(OUT, "g2_prior.npz"), k=10, inf=1.1, weights=w.numpy())

    # ---- G3 / G4: random (k, p) blocks through ETKFModule and KETKFModule
    rnd = np.random.RandomState(42)
    g3 = {}
    cases = [(10, 40), (20, 40), (40, 20), (40, 19), (80, 64), (40, 1), (7, 5), (64, 64), (33, 100)]
    for ci, (k, p) in enumerate(cases):
        yb = rnd.normal(size=(k, p))
        yb -= yb.mean(axis=0)
        dd = rnd.normal(size=(p,))
        g3[f"yb_{ci}"], g3[f"d_{ci}"] = yb, dd
        for inf in (1.0, 1.1):
            tag = f"{ci}_{str(inf).replace('.', 'p')}"
            g3[f"etkf_{tag}"] = ETKFModule(t64(inf))(t64(yb), t64(dd)).numpy()
            for gname, kern in (("rbf0p5", RBFKernel(t64(0.5))), ("rbf10", RBFKernel(t64(10.0))),
                                ("gauss2", GaussKernel(t64(2.0))), ("linear", LinearKernel())):
                g3[f"ketkf_{gname}_{tag}"] = KETKFModule(kern, t64(inf))(t64(yb), t64(dd)).numpy()
    g3["cases"] = np.array(cases)
    # raw kernel matrices (tests/unit_tests/kernels/test_rbf.py:52-96)
    xk, yk = rnd.normal(size=(6, 4)), rnd.normal(size=(3, 4))
    g3["kern_x"], g3["kern_y"] = xk, yk
    g3["kern_rbf0p5"] = RBFKernel(t64(0.5))(t64(xk), t64(yk)).numpy()
    g3["kern_rbf10"] = RBFKernel(t64(10.0))(t64(xk), t64(yk)).numpy()
    g3["kern_gauss2"] = GaussKernel(t64(2.0))(t64(xk), t64(yk)).numpy()
    g3["kern_linear"] = LinearKernel()(t64(xk), t64(yk)).numpy()
    np.savez(os.path.join(OUT, "g3_g4_core_blocks.npz"), **g3)

    # ---- G5: Gaspari-Cohn
    r = np.concatenate([np.linspace(0, 2.5, 251), [0.0, 0.5, 1.0, 1.5, 1.8, 1.9, 1.95, 1.999, 2.0, 2.0001]])
    g5 = {"r": r}
    g5["f1"] = GaspariCohn._f1(r.copy())
    with np.errstate(all="ignore"):
        g5["f2"] = GaspariCohn._f2(r[r > 0].copy())
    for c in (1.0, 10.0, 16.5):
        gc = GaspariCohn(c, dist_func=lambda g, o: np.abs(o - g))
        use, w = gc.localize_obs(0.0, r * c)
        g5[f"use_c{c}"], g5[f"w_c{c}"] = use, w
    # two radii (horizontal x vertical product), distances as a tuple of arrays
    dh, dv = rnd.uniform(0, 25, size=300), rnd.uniform(0, 3, size=300)
    gc2 = GaspariCohn((10.0, 1.5), dist_func=lambda g, o: (o[:, 0], o[:, 1]))
    use, w = gc2.localize_obs(None, np.stack([dh, dv], axis=1))
    g5.update(dh=dh, dv=dv, use_2r=use, w_2r=w)
    gc3 = GaspariCohn(10.0, dist_func=lambda g, o: np.abs(o - g), epsilon=1e-3)
    use, w = gc3.localize_obs(0.0, r * 10.0)
    g5.update(use_eps1e3=use, w_eps1e3=w)
    np.savez(os.path.join(OUT, "g5_gaspari_cohn.npz"), **g5)

    # ---- G6: end-to-end LETKF on the reference's own fixtures (tests/data/*.nc, read with scipy)
    from scipy.io import netcdf_file
    f = netcdf_file(f"{REF}/tests/data/test_state.nc", mmap=False)
    state = np.array(f.variables["__xarray_dataarray_variable__"][:], dtype=np.float64)  # (2,3,10,40)
    grid = np.array(f.variables["grid"][:], dtype=np.float64)
    f.close()
    f = netcdf_file(f"{REF}/tests/data/test_single_obs.nc", mmap=False)
    obs = np.array(f.variables["observations"][:], dtype=np.float64)       # (3, 40)
    cov = np.array(f.variables["covariance"][:], dtype=np.float64)         # (40, 40) = 0.5 I
    obs_grid = np.array(f.variables["obs_grid_1"][:], dtype=np.float64)
    f.close()
    ti = 0
    hx = state[0, ti]                          # dummy_obs_operator: var 'x', identity H (testing/dummy.py:39-66)
    mean = hx.mean(axis=0)
    chol_inv = np.linalg.inv(np.linalg.cholesky(cov).T)   # observation.py:247-252
    yb = (hx - mean) @ chol_inv
    dd = (obs[ti] - mean) @ chol_inv
    g6 = dict(state=state, grid=grid, obs=obs, cov=cov, obs_grid=obs_grid, yb=yb, d=dd, time_index=ti)
    for inf in (1.0, 1.1):
        tag = str(inf).replace(".", "p")
        f_loc = wrapper_localization(
            wrapper_bridge(ETKFModule(t64(inf)), torch.device("cpu"), torch.float64),
            GaspariCohn(10.0, lambda g, o: np.abs(o - g[1])))
        w = np.stack([f_loc(np.array([0.0, x]), yb, dd, obs_info=obs_grid) for x in grid])
        g6[f"weights_{tag}"] = w
        g6[f"analysis_{tag}"] = apply_weights(state[:, [ti]], w)
        # no localisation == global ETKF (interface/test_letkf.py:64-70)
        f_glob = wrapper_localization(
            wrapper_bridge(ETKFModule(t64(inf)), torch.device("cpu"), torch.float64), None)
        wg = f_glob(np.array([0.0, 0.0]), yb, dd, obs_info=obs_grid)
        g6[f"weights_global_{tag}"] = wg
        g6[f"analysis_global_{tag}"] = apply_weights(state[:, [ti]], np.broadcast_to(wg, (40,) + wg.shape))
    # the fixtures' time axes (hours since 1992-12-25 in both files) as unix seconds, and the SMOOTHER-mode case of the
    # same fixture: nothing is sliced (filter.py:150-153 is skipped), all 3 x 40 observations are stacked time-major
    # (base.py:223-241) against the three-time state; weights through the reference's localised core, analysis over
    # every time of the state (base.py:257-278)
    t_unix = (np.datetime64("1992-12-25T00:00:00") - np.datetime64("1970-01-01T00:00:00")) / np.timedelta64(1, "s")
    g6["state_time"] = t_unix + 3600.0 * np.arange(3.0)
    g6["obs_time"] = t_unix + 3600.0 * np.arange(3.0)
    hx_s = state[0].transpose(1, 0, 2)                                  # (ensemble, time, grid)
    mean_s = hx_s.mean(axis=0)
    yb_s = ((hx_s - mean_s) @ chol_inv).reshape(10, 120)
    d_s = ((obs - mean_s) @ chol_inv).reshape(120)
    f_loc = wrapper_localization(
        wrapper_bridge(ETKFModule(t64(1.1)), torch.device("cpu"), torch.float64),
        GaspariCohn(10.0, lambda g, o: np.abs(o - g[1])))
    w_s = np.stack([f_loc(np.array([0.0, x]), yb_s, d_s, obs_info=np.tile(obs_grid, 3)) for x in grid])
    g6.update(yb_smoother=yb_s, d_smoother=d_s, weights_smoother_1p1=w_s, analysis_smoother_1p1=apply_weights(state, w_s))
    np.savez(os.path.join(OUT, "g6_reference_fixture_letkf.npz"), **g6)

    # ---- G7: scaled-down synthetic configs (SURVEY.md §8d generator), G = 256
    def synth(G, k, s, seed=42, m=1):
        rs = np.random.RandomState(seed)
        st = rs.normal(size=(m, k, G))
        ox = np.arange(0, G, s, dtype=np.float64)
        y = rs.normal(size=ox.shape[0])
        hx_ = st[0][:, ::s]
        mu = hx_.mean(axis=0)
        return st, np.arange(G, dtype=np.float64), ox, hx_ - mu, y - mu

    g7 = {}
    for name, (G, k, s, c, kern, m) in dict(
            c2=(256, 40, 2, 10.0, None, 1), c4=(160, 80, 1, 16.5, None, 1),
            c5=(256, 40, 2, 10.0, 0.5, 1), c2m3=(96, 40, 2, 10.0, None, 3),
            c1=(40, 20, 1, None, None, 1)).items():
        st, gx, ox, yb, dd = synth(G, k, s, m=m)
        g7.update({f"{name}_state": st, f"{name}_grid_x": gx, f"{name}_obs_x": ox,
                   f"{name}_yb": yb, f"{name}_d": dd})
        for inf in (1.0, 1.1):
            tag = f"{name}_{str(inf).replace('.', 'p')}"
            core = ETKFModule(t64(inf)) if kern is None else KETKFModule(RBFKernel(t64(kern)), t64(inf))
            loc = None if c is None else GaspariCohn(c, lambda g, o: np.abs(o - g[1]))
            f_loc = wrapper_localization(wrapper_bridge(core, torch.device("cpu"), torch.float64), loc)
            if c is None:
                wg = f_loc(np.array([0.0, 0.0]), yb, dd, obs_info=ox)
                w = np.broadcast_to(wg, (G,) + wg.shape).copy()
            else:
                w = np.stack([f_loc(np.array([0.0, x]), yb, dd, obs_info=ox) for x in gx])
            widx = np.unique(np.clip(np.array([0, 1, 2, 5, 9, 17, 20, G // 2, G // 2 + 1, G - 3, G - 2, G - 1]), 0, G - 1))
            g7[f"{name}_widx"] = widx               # weights kept for a few sample points only (fixture size)
            g7[f"{tag}_weights"] = w[widx]
            g7[f"{tag}_analysis"] = apply_weights(st, w)
    np.savez_compressed(os.path.join(OUT, "g7_synthetic_configs.npz"), **g7)
    # ---- G8: rows f2 / f4 of SURVEY.md section 8: GaspariCohnInf and the remaining kernels + compositions
    GaspariCohnInf = R["gc"].GaspariCohnInf
    g8 = {"r": r}
    with np.errstate(all="ignore"):
        for i, fn in enumerate((GaspariCohnInf._f1, GaspariCohnInf._f2, GaspariCohnInf._f3, GaspariCohnInf._f4)):
            g8[f"inf_f{i + 1}"] = fn(r[r > 0].copy())
    for c in (1.0, 10.0):
        gci = GaspariCohnInf(c, dist_func=lambda g, o: np.abs(o - g))
        use, w = gci.localize_obs(0.0, r * c)
        g8[f"inf_use_c{c}"], g8[f"inf_w_c{c}"] = use, w
    kernels = dict(
        poly2=R["polynomial"].PolyKernel(t64(2.0), t64(1.0)),
        poly3=R["polynomial"].PolyKernel(t64(3.0), t64(0.5)),
        tanh=R["tanh"].TanhKernel(t64(0.05), t64(0.1)),
        periodic=R["periodic"].PeriodicKernel(t64(7.0), t64(1.5)),
        rational=R["rational"].RationalKernel(t64(2.0), t64(1.5)),
        ornuhl=R["orn_uhl"].OrnsteinUhlenbeckKernel(t64(6.0)),
    )
    kernels["rbf_plus_diag"] = RBFKernel(t64(0.5)) + R["diag"].DiagKernel(t64(0.3))
    kernels["scale_times_rbf"] = R["scale"].ScaleKernel(t64(2.5)) * GaussKernel(t64(2.0))
    kernels["linear_plus_scale"] = LinearKernel() + R["scale"].ScaleKernel(t64(0.7))
    kernels["rational_pow_scale"] = R["rational"].RationalKernel(t64(1.0), t64(1.0)) ** R["scale"].ScaleKernel(t64(2.0))
    kernels["poly_plus_ornuhl_times_scale"] = (R["polynomial"].PolyKernel(t64(2.0), t64(1.0))
                                               + R["orn_uhl"].OrnsteinUhlenbeckKernel(t64(4.0)) * R["scale"].ScaleKernel(t64(3.0)))
    rk = np.random.RandomState(7)
    xk, yk = rk.normal(size=(6, 4)), rk.normal(size=(1, 4))
    g8["kern_x"], g8["kern_y"] = xk, yk
    blocks = [(40, 20), (10, 40), (20, 7)]
    for bi, (k, p) in enumerate(blocks):
        yb = rk.normal(size=(k, p)) * 0.6
        yb -= yb.mean(axis=0)
        g8[f"yb_{bi}"], g8[f"d_{bi}"] = yb, rk.normal(size=(p,)) * 0.6
    g8["blocks"] = np.array(blocks)
    for name, kern in kernels.items():
        g8[f"kxx_{name}"] = kern(t64(xk), t64(xk)).numpy()
        g8[f"kxy_{name}"] = kern(t64(xk), t64(yk)).numpy()
        for bi in range(len(blocks)):
            for inf in (1.0, 1.1):
                tag = f"{name}_{bi}_{str(inf).replace('.', 'p')}"
                g8[f"ketkf_{tag}"] = KETKFModule(kern, t64(inf))(t64(g8[f"yb_{bi}"]), t64(g8[f"d_{bi}"])).numpy()
    # localised: LKETKF with a polynomial kernel and LETKF under GaspariCohnInf, G = 64 (synthetic recipe of G7)
    st, gx, ox, yb, dd = synth(64, 40, 2, seed=11)
    g8.update(loc_state=st, loc_grid_x=gx, loc_obs_x=ox, loc_yb=yb, loc_d=dd)
    for tag, core, loc in (
            ("lketkf_poly2", KETKFModule(kernels["poly2"], t64(1.1)), GaspariCohn(10.0, lambda g, o: np.abs(o - g[1]))),
            ("lketkf_ornuhl", KETKFModule(kernels["ornuhl"], t64(1.1)), GaspariCohn(10.0, lambda g, o: np.abs(o - g[1]))),
            ("letkf_gcinf", ETKFModule(t64(1.1)), GaspariCohnInf(10.0, lambda g, o: np.abs(o - g[1])))):
        f_loc = wrapper_localization(wrapper_bridge(core, torch.device("cpu"), torch.float64), loc)
        w = np.stack([f_loc(np.array([0.0, x]), yb, dd, obs_info=ox) for x in gx])
        g8[f"{tag}_weights"] = w[::8]
        g8[f"{tag}_analysis"] = apply_weights(st, w)
    np.savez_compressed(os.path.join(OUT, "g8_kernels_gcinf.npz"), **g8)

    # ---- G9: row f3, IEnKS weight update (core/ienks.py) -- chained iterations from the prior weights
    IEnKSTransformModule, IEnKSBundleModule = R["ienks"].IEnKSTransformModule, R["ienks"].IEnKSBundleModule
    r9 = np.random.RandomState(9)
    g9 = {}
    blocks9 = [(10, 6), (40, 20), (20, 40), (7, 5), (40, 0)]
    g9["blocks"] = np.array(blocks9)
    for bi, (k, p) in enumerate(blocks9):
        yb = r9.normal(size=(k, p))
        yb -= yb.mean(axis=0)
        dd = r9.normal(size=(p,))
        w0 = np.eye(k) + 0.1 * r9.normal(size=(k, k))          # a general, non-symmetric starting point
        g9[f"yb_{bi}"], g9[f"d_{bi}"], g9[f"w0_{bi}"] = yb, dd, w0
        for tau in (1.0, 0.7):
            ttag = str(tau).replace(".", "p")
            for vname, mod, scale in (("transform", IEnKSTransformModule(t64(tau)), 1.0),
                                      ("bundle", IEnKSBundleModule(t64(1e-4), t64(tau)), 1e-4)):
                w = t64(np.eye(k))
                for it in range(3):
                    w = mod(w, t64(yb * scale), t64(dd))
                    g9[f"{vname}_{bi}_{ttag}_it{it}"] = w.numpy()
                g9[f"{vname}_{bi}_{ttag}_general"] = mod(t64(w0), t64(yb * scale), t64(dd)).numpy()
    # localised: two iterations of the per-grid-point loop of interface/lienks.py:88-113 on fixed obs-space input
    st, gx, ox, yb, dd = synth(64, 40, 2, seed=13)
    g9.update(loc_state=st, loc_grid_x=gx, loc_obs_x=ox, loc_yb=yb, loc_d=dd)
    for vname, mod, scale in (("transform", IEnKSTransformModule(t64(0.8)), 1.0),
                              ("bundle", IEnKSBundleModule(t64(1e-3), t64(1.0)), 1e-3)):
        f_loc = wrapper_localization(wrapper_bridge(mod, torch.device("cpu"), torch.float64),
                                     GaspariCohn(10.0, lambda g, o: np.abs(o - g[1])))
        w = np.broadcast_to(np.eye(40), (64, 40, 40))
        for it in range(2):
            w = np.stack([f_loc(np.array([0.0, x]), w[gi], yb * scale, dd, obs_info=ox, args_to_skip=(0,))
                          for gi, x in enumerate(gx)])
            g9[f"loc_{vname}_it{it}_weights"] = w[::8]
        g9[f"loc_{vname}_analysis"] = apply_weights(st, w)
    np.savez_compressed(os.path.join(OUT, "g9_ienks.npz"), **g9)

    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
