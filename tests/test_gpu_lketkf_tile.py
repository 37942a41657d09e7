"""GPU parity of the RBF-kernelised filter on the tile route (csrc/lketkf_tile.hip, mia_lketkf_rbf_analysis_tiles_f32) through
the C ABI: reference-generated golden blocks (KETKFModule + RBFKernel / GaussKernel, core/ketkf.py:65-94, kernels/rbf.py),
the float64 oracle on synthetic grids (ensemble sizes that fill the lanes' rows and ones that do not), shards, many state
rows, non-finite records, union overflow, and the step driver."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu
TOL32 = 1e-5          # north star: relative Frobenius error of the float32 analysis


@pytest.fixture(scope="module")
def mia():
    import torch_assimilate_amd as m
    m.build()
    return m


@pytest.fixture(scope="module")
def eng(mia):
    return mia.LetkfEngine("cuda:0")


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def rbf_core(gamma):
    return lambda a, b, i, g_=gamma: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, g_), i)


def oracle_analysis(state, gx, ox, yb, d, c, inf, gamma, pts):
    out = []
    for g in pts:
        w = O.localized_weights(O.abs_distance_1d(gx[g], ox), yb, d, [c], inf, core=rbf_core(gamma))
        out.append(O.apply_weights(state[:, :, [g]], w[None])[:, :, 0])
    return np.stack(out, axis=-1)


def test_core_blocks_on_the_tile_route_vs_reference(eng, golden):
    """The reference's own KETKF weights (golden g4: RBF gamma 0.5 / 10, Gauss lengthscale 2 = gamma 0.125; inflation 1.0 / 1.1)
    applied to a random ensemble = the tile kernel's analysis of ONE grid point that sees every observation with weight 1."""
    g = golden("g3_g4_core_blocks.npz")
    seen = 0
    for ci, (k, p) in enumerate(g["cases"]):
        yb, d = g[f"yb_{ci}"], g[f"d_{ci}"]
        X = np.random.RandomState(100 + ci).normal(size=(3, k, 1))
        if (p + 8 + 15) // 16 > 6:                  # (more observations than the tile-list format has slots)
            continue
        tiles = eng.localize_tiles(np.zeros(1), np.zeros(p), [5.0], int(p))
        for gname, gamma in (("rbf0p5", 0.5), ("rbf10", 10.0), ("gauss2", 0.125)):
            for inf, tag in ((1.0, "1p0"), (1.1, "1p1")):
                res = eng.analysis_tiles_rbf(dev(X), dev(yb), dev(d), tiles, inf, gamma)
                if res is None:
                    assert k > 40 or (p + 8 + 15) // 16 > min(4, (k + 15) // 16 + 1), (k, p)     # outside the kernel by its own rule
                    continue
                xa, fl, retry = res
                assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0
                ref = O.apply_weights(X, g[f"ketkf_{gname}_{ci}_{tag}"][None])
                assert rel_fro(xa.cpu().numpy(), ref) < TOL32, (k, p, gname, tag)
                seen += 1
    assert seen >= 24         # (k, p) = (20, 40), (40, 20), (40, 19), (40, 1), (7, 5) x 3 kernels x 2 inflations at least


def test_scaled_down_config5_vs_golden(eng, golden):
    """golden g7, config 5 scaled down (G = 256, k = 40, RBF gamma 0.5): the reference's analysis at every grid point"""
    g = golden("g7_synthetic_configs.npz")
    X, gx, ox, yb, d = g["c5_state"], g["c5_grid_x"], g["c5_obs_x"], g["c5_yb"], g["c5_d"]
    nb = eng.localize(gx, ox, [10.0])
    tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
    for inf, tag in ((1.0, "1p0"), (1.1, "1p1")):
        xa, fl, retry = eng.analysis_tiles_rbf(dev(X), dev(yb), dev(d), tiles, inf, 0.5)
        assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0
        assert rel_fro(xa.cpu().numpy(), g[f"c5_{tag}_analysis"]) < TOL32


@pytest.mark.parametrize("k,stride,c", [(40, 2, 10.0), (32, 1, 4.0), (37, 2, 10.0), (20, 3, 25.0), (8, 2, 6.0), (5, 1, 3.0), (24, 2, 10.0)])
def test_shapes_vs_oracle(eng, k, stride, c):
    """ensemble sizes that fill the four lanes' rows exactly (k = 40, 32, 24, 8) and ones that leave rows empty (37, 20, 5),
    three observation networks, gamma 0.02 / 0.5 / 10, two inflations, three state rows: every grid point against the oracle"""
    G = 150
    case = O.synthetic_case(G, k, stride, seed=7 + k, m=3)
    X, gx, ox, yb, d = case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"]
    nb = eng.localize(gx, ox, [c])
    extra = 0
    tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
    if int(tiles.stats[1].item()):
        extra = 1
        tiles = eng.localize_tiles(gx, ox, [c], nb.p_max, extra_blocks=1)
    assert int(tiles.stats[1].item()) == 0
    for gamma in (0.02, 0.5, 10.0):
        for inf in (1.0, 1.1):
            res = eng.analysis_tiles_rbf(dev(X), dev(yb), dev(d), tiles, inf, gamma)
            assert res is not None, (k, nb.p_max, extra)
            xa, fl, retry = res
            assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0
            ref = oracle_analysis(X, gx, ox, yb, d, c, inf, gamma, range(G))
            got = xa.cpu().numpy()
            assert rel_fro(got, ref) < TOL32, (k, gamma, inf)
            mean = X.mean(axis=1, keepdims=True)
            assert rel_fro(got - mean, ref - mean) < 5e-5, (k, gamma, inf)


def test_shards_determinism_and_the_point_route(eng):
    """two runs agree bit for bit; a shard that cuts the grid anywhere reproduces the full run per point to rounding (the powers
    of two of the split products belong to the tile); the one-point-per-wavefront kernel (letkf_cheb.hip) agrees to rounding"""
    G, k = 5000, 40
    case = O.synthetic_case(G, k, 2, seed=11)
    X, yb, d = dev(case["state"]), dev(case["yb"]), dev(case["d"])
    gx, ox = case["grid_x"], case["obs_x"]
    nb = eng.localize(gx, ox, [10.0])
    tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
    xa, fl, _ = eng.analysis_tiles_rbf(X, yb, d, tiles, 1.1, 0.5)
    assert torch.equal(xa, eng.analysis_tiles_rbf(X, yb, d, tiles, 1.1, 0.5)[0])
    g0, g1 = 1237, 1411
    part = eng.analysis_tiles_rbf(X, yb, d, eng.localize_tiles(gx, ox, [10.0], nb.p_max, g0=g0, g1=g1), 1.1, 0.5)[0]
    ref_part = xa[:, :, g0:g1]
    # (the per-slot / per-point powers of two of the split products depend on the tile: equal to rounding, not bit for bit)
    assert float(((part - ref_part).norm(dim=(0, 1)) / ref_part.norm(dim=(0, 1))).max()) < 2e-6
    xo = eng.analysis(X, yb, d, nb, 1.1, rbf_gamma=0.5, method="matfun")
    assert float(torch.linalg.norm(xa - xo) / torch.linalg.norm(xo)) < 1e-6


def test_nonfinite_record_and_union_overflow(eng):
    """a NaN in one observation's perturbations: every point of the tiles that hold it is handed to the eigensolver kernel
    (MIA_FLAG_RETRY), which works point by point: afterwards no point that does not see the observation is touched by it (they
    equal the point route's result); a union that does not fit its slots is flagged, never truncated"""
    G, k = 640, 40
    case = O.synthetic_case(G, k, 2, seed=3)
    gx, ox = case["grid_x"], case["obs_x"]
    yb_bad = case["yb"].copy()
    yb_bad[5, 100] = np.nan
    X, yb, d = dev(case["state"]), dev(yb_bad), dev(case["d"])
    nb = eng.localize(gx, ox, [10.0])
    tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
    xa, fl, retry = eng.analysis_tiles_rbf(X, yb, d, tiles, 1.1, 0.5)
    n_retry = int(retry.item())
    assert n_retry > 0 and n_retry % 16 == 0
    flagged = ((fl & 0xff) == 8).cpu().numpy()           # MIA_FLAG_RETRY
    assert flagged.sum() == n_retry
    eng.retry_points(X, yb, d, nb, 1.1, xa, fl, rbf_gamma=0.5)
    xo = eng.analysis(X, yb, d, nb, 1.1, rbf_gamma=0.5, method="matfun")
    sees = (np.abs(gx - ox[100]) < 20.0)
    fin = torch.isfinite(xa).all(dim=(0, 1)).cpu().numpy()
    assert not (~fin & ~sees).any()
    # (what the points that see the record get is the eigensolver kernel's business -- the reference raises or returns NaN
    #  there; every other point, also those that share a tile with them, equals the clean analysis)
    ok = torch.as_tensor(~sees, device=xa.device)
    assert float(torch.linalg.norm(xa[:, :, ok] - xo[:, :, ok]) / torch.linalg.norm(xo[:, :, ok])) < TOL32
    # union overflow: lists sized for a bound that the tiles' unions exceed
    small = eng.localize_tiles(gx, ox, [10.0], 4)
    assert int(small.stats[1].item()) > 0
    xa2, fl2, _ = eng.analysis_tiles_rbf(dev(case["state"]), dev(case["yb"]), dev(case["d"]), small, 1.1, 0.5)
    over = ((fl2 & 0xff) == 1).cpu().numpy()             # MIA_FLAG_OVERFLOW
    assert over.any() and bool(torch.isnan(xa2[:, :, torch.as_tensor(over, device=xa2.device)]).all())


def test_step_driver_takes_the_tile_route(mia):
    """ShardedLetkf with rbf_gamma: the exact-list call, the native one-call step and steps in flight all take the tile route and
    agree; a geometry epoch (reused lists) reproduces the full rebuild bit for bit; 64 oracle points"""
    dev0 = torch.device("cuda:0")
    G, k = 20000, 40
    case = O.synthetic_case(G, k, 2, seed=21)
    X, yb, d = dev(case["state"]), dev(case["yb"]), dev(case["d"])
    gx, ox = torch.as_tensor(case["grid_x"], device=dev0), torch.as_tensor(case["obs_x"], device=dev0)
    r = mia.ShardedLetkf(dev0, 0, 1, radii=[10.0], inf_factor=1.1, rbf_gamma=0.5, max_in_flight=3)
    first = r.assimilate(X, gx, ox, yb, d)              # exact lists (entry by entry)
    second = r.assimilate(X, gx, ox, yb, d)             # the one-call step driver
    assert r.native_steps == 1 and r.last_flags_ok() and r.last_retries == 0
    assert r.dominant_kernel_name.startswith("lketkf_tile_kernel")
    assert torch.equal(first, second)
    pend = [r.submit(X, gx, ox, yb, d, geometry_id="net") for _ in range(4)]
    for h in pend:
        assert torch.equal(h.result(), second)
    assert r.reused_steps >= 1
    pts = np.random.RandomState(4).choice(G, 64, replace=False)
    ref = oracle_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1, 0.5, pts)
    got = second[:, :, torch.as_tensor(pts, device=dev0)].double().cpu().numpy()
    assert rel_fro(got, ref) < TOL32
    eng = r.engine
    nb = eng.localize(gx, ox, [10.0])
    xo = eng.analysis(X, yb, d, nb, 1.1, rbf_gamma=0.5, method="matfun")
    assert float(torch.linalg.norm(second - xo) / torch.linalg.norm(xo)) < 1e-6


def test_points_without_local_observations(eng):
    """Observations on one third of the domain only: points that see none get the prior branch of ETKFModule.forward
    (core/etkf.py:91-95: weights sqrt(inf) I, i.e. mean + sqrt(inf) x') from the same kernel, tiles that mix both kinds included."""
    G, k = 400, 40
    case = O.synthetic_case(G, k, 2, seed=17)
    gx, ox = case["grid_x"], case["obs_x"]
    keep = ox < 130.0
    ox, yb, d = ox[keep], case["yb"][:, keep], case["d"][keep]
    X = case["state"]
    nb = eng.localize(gx, ox, [10.0])
    tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
    for inf in (1.0, 1.3):
        xa, fl, retry = eng.analysis_tiles_rbf(dev(X), dev(yb), dev(d), tiles, inf, 0.5)
        assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0
        ref = oracle_analysis(X, gx, ox, yb, d, 10.0, inf, 0.5, range(G))
        got = xa.cpu().numpy()
        assert rel_fro(got, ref) < TOL32
        far = gx > 160.0
        mean = X.mean(axis=1, keepdims=True)
        assert rel_fro(got[:, :, far], (mean + np.sqrt(inf) * (X - mean))[:, :, far]) < TOL32
