"""GPU tests of the host-side mirror of the reference interface; they read like the reference's
tests/unit_tests/{core/test_etkf.py, interface/test_letkf.py, interface/test_lketkf.py}."""
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


def test_core_module_known_answers(mia, golden):
    """core/test_etkf.py:142-240 through the per-block seam."""
    g = golden("g1_known_answer.npz")
    mod = mia.ETKFModule(1.0)
    w = mod(torch.tensor(g["yb"]), torch.tensor(g["d"])).cpu().numpy()
    np.testing.assert_allclose(w, g["weights"], atol=1e-12)
    np.testing.assert_allclose((w - np.eye(2)).mean(axis=1), [0.1, -0.1], atol=1e-12)   # :219-225
    mod.inf_factor = 1.1
    w0 = mod(torch.ones(10, 0, dtype=torch.float64), torch.ones(1, 0, dtype=torch.float64)).cpu().numpy()
    np.testing.assert_allclose(w0, np.sqrt(1.1) * np.eye(10), atol=1e-14)                # :227-233
    with pytest.raises(ValueError):
        mod(torch.ones(10, 4), torch.ones(1, 3))                                          # :235-240
    w32 = mod(torch.tensor(g["yb"], dtype=torch.float32), torch.tensor(g["d"], dtype=torch.float32))
    assert w32.dtype == torch.float32 and w32.is_cuda


def test_ketkf_module_vs_reference(mia, golden):
    g = golden("g3_g4_core_blocks.npz")
    yb, d = g["yb_2"], g["d_2"]                       # (40, 20)
    for name, kern in (("rbf0p5", mia.RBFKernel(0.5)), ("rbf10", mia.RBFKernel(10.0)),
                       ("gauss2", mia.GaussKernel(2.0)), ("linear", mia.LinearKernel())):
        w = mia.KETKFModule(kern, 1.1)(torch.tensor(yb), torch.tensor(d)).cpu().numpy()
        assert rel_fro(w, g[f"ketkf_{name}_2_1p1"]) < 1e-9, name


def test_gaspari_cohn_localize_obs_api(mia, golden):
    g = golden("g5_gaspari_cohn.npz")
    r = g["r"]
    gc = mia.GaspariCohn(10.0, dist_func=lambda grid, obs: np.abs(obs - grid))
    use, w = gc.localize_obs(0.0, r * 10.0)
    np.testing.assert_array_equal(use, g["use_c10.0"])
    np.testing.assert_allclose(w, g["w_c10.0"], rtol=0, atol=4e-15)
    gc2 = mia.GaspariCohn((10.0, 1.5), dist_func=lambda grid, obs: (obs[:, 0], obs[:, 1]))
    use, w = gc2.localize_obs(None, np.stack([g["dh"], g["dv"]], axis=1))
    np.testing.assert_array_equal(use, g["use_2r"])
    np.testing.assert_allclose(w, g["w_2r"], rtol=0, atol=4e-15)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
def test_letkf_localized_right(mia, golden, dtype, tol):
    """interface/test_letkf.py:106-157 on the reference's own fixture, with the built-in metric and
    with an arbitrary python dist_func evaluated on the host."""
    g = golden("g6_reference_fixture_letkf.npz")
    ti = int(g["time_index"])
    state = g["state"][:, [ti]]                       # (var, time, ens, grid)
    for dist_func in (mia.AbsoluteDistance(), lambda grid, obs: np.abs(obs[:, 0] - grid[-1])):
        algo = mia.LETKF(localization=mia.GaspariCohn(10.0, dist_func), dtype=dtype)
        xa = algo.analyse_arrays(state, g["yb"], g["d"], grid_coords=g["grid"][:, None], obs_coords=g["obs_grid"][:, None])
        assert rel_fro(xa.cpu().numpy(), g["analysis_1p0"]) < tol
        W = algo.estimate_weights_arrays(g["yb"], g["d"], grid_coords=g["grid"][:, None], obs_coords=g["obs_grid"][:, None])
        assert rel_fro(W.cpu().numpy(), g["weights_1p0"]) < tol
    algo.inf_factor = 1.1
    xa = algo.analyse_arrays(state, g["yb"], g["d"], grid_coords=g["grid"][:, None], obs_coords=g["obs_grid"][:, None])
    assert rel_fro(xa.cpu().numpy(), g["analysis_1p1"]) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
def test_letkf_without_localization_equals_etkf(mia, golden, dtype, tol):
    """interface/test_letkf.py:64-70 and :79-104 (zero-distance Gaspari-Cohn)."""
    g = golden("g6_reference_fixture_letkf.npz")
    ti = int(g["time_index"])
    state = g["state"][:, [ti]]
    kw = dict(grid_coords=g["grid"][:, None], obs_coords=g["obs_grid"][:, None])
    xa_e = mia.ETKF(dtype=dtype).analyse_arrays(state, g["yb"], g["d"])
    assert rel_fro(xa_e.cpu().numpy(), g["analysis_global_1p0"]) < tol
    xa_l = mia.LETKF(dtype=dtype).analyse_arrays(state, g["yb"], g["d"], **kw)
    assert rel_fro(xa_l.cpu().numpy(), g["analysis_global_1p0"]) < tol
    zero = mia.GaspariCohn((1.0, 1.0), dist_func=lambda grid, obs: np.zeros((2, obs.shape[0])))
    xa_z = mia.LETKF(localization=zero, dtype=dtype).analyse_arrays(state, g["yb"], g["d"], **kw)
    assert rel_fro(xa_z.cpu().numpy(), g["analysis_global_1p0"]) < tol


def test_lketkf_linear_equals_letkf_and_rbf_vs_reference(mia, golden):
    """interface/test_lketkf.py:109-117 (linear kernel == ETKF) and the RBF config."""
    g = golden("g7_synthetic_configs.npz")
    loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())
    kw = dict(grid_coords=g["c5_grid_x"], obs_coords=g["c5_obs_x"])
    st = g["c5_state"]
    a = mia.LKETKF(mia.LinearKernel(), localization=loc, inf_factor=1.1, dtype=torch.float64)
    xa_lin = a.analyse_arrays(st, g["c5_yb"], g["c5_d"], **kw)
    assert rel_fro(xa_lin.cpu().numpy(), g["c2_1p1_analysis"]) < 1e-10       # c5 shares c2's inputs
    a.kernel = mia.RBFKernel(0.5)
    xa_rbf = a.analyse_arrays(st, g["c5_yb"], g["c5_d"], **kw)
    assert rel_fro(xa_rbf.cpu().numpy(), g["c5_1p1_analysis"]) < 1e-9
    a32 = mia.LKETKF(mia.RBFKernel(0.5), localization=loc, inf_factor=1.1)
    assert rel_fro(a32.analyse_arrays(st, g["c5_yb"], g["c5_d"], **kw).cpu().numpy(), g["c5_1p1_analysis"]) < 1e-5
    ke = mia.KETKF(mia.RBFKernel(0.5), inf_factor=1.1, dtype=torch.float64)
    gg = golden("g3_g4_core_blocks.npz")
    w = ke.estimate_weights_arrays(gg["yb_2"], gg["d_2"]).cpu().numpy()
    assert rel_fro(w, gg["ketkf_rbf0p5_2_1p1"]) < 1e-9


def test_sharded_runner_single_gpu_matches_oracle(mia):
    case = O.synthetic_case(2000, 40, 2)
    dev = torch.device("cuda:0")
    runner = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    X = torch.as_tensor(case["state"], dtype=torch.float32, device=dev)
    out = runner.assimilate(X, torch.as_tensor(case["grid_x"], device=dev), torch.as_tensor(case["obs_x"], device=dev),
                            torch.as_tensor(case["yb"], dtype=torch.float32, device=dev),
                            torch.as_tensor(case["d"], dtype=torch.float32, device=dev))
    assert runner.last_flags_ok()
    sel = np.arange(0, 2000, 97)
    for gi in sel:
        dist = O.abs_distance_1d(case["grid_x"][gi], case["obs_x"])
        w = O.localized_weights(dist, case["yb"], case["d"], [10.0], 1.1)
        ref = O.apply_weights(case["state"][:, :, [gi]], w[None])
        assert rel_fro(out[:, :, gi].cpu().numpy(), ref[:, :, 0]) < 1e-5


@pytest.mark.parametrize("G,chunks,strong", [(2000, 4, False), (1999, 3, False), (1000, 4, True)])
def test_overlapped_exchange_path_on_one_gpu(mia, G, chunks, strong):
    """The compute / all-gather overlap route of the multi-GPU runner (side stream, events, one RCCL
    all-gather per chunk, final permute), driven on ONE GPU through a single-rank RCCL group: must give the
    same ensemble as the plain route, twice in a row (second call uses the assumed list bound).  `strong`:
    accurate observations make the matfun kernel decline points, exercising the re-exchange of redone chunks."""
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    own = not dist.is_initialized()
    if own:
        import os, socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        case = O.synthetic_case(G, 40, 2)
        scale = 12.0 if strong else 1.0
        args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
                torch.as_tensor(case["obs_x"], device=dev),
                torch.as_tensor(case["yb"] * scale, dtype=torch.float32, device=dev),
                torch.as_tensor(case["d"] * scale, dtype=torch.float32, device=dev))
        plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
        ref = plain.assimilate(*args)
        over = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, comm_chunks=chunks)
        for _ in range(2):
            out = over._assimilate_overlapped(*args, G, 0, G)
            torch.cuda.synchronize()
            assert out.shape == ref.shape
            assert over.last_flags_ok()
            assert rel_fro(out.cpu().numpy(), ref.cpu().numpy()) < 1e-6
        if strong:
            assert plain.last_retries > 0 and over.last_retries == plain.last_retries
    finally:
        if own:
            dist.destroy_process_group()
