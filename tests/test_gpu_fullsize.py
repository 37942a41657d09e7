"""Full-size GPU tests (BASELINE.json configurations 2 - 5 at their stated sizes): EVERY grid point of configurations 2, 4, 5 and
1e5 of configuration 3's 1e6 against the float64 oracle through the step driver with steps in flight (the oracle runs on a pool
of CPU workers forked at session start, tests/oracle_pool.py), plus properties that need no oracle at scale (determinism, shard
invariance, agreement of independent routes).
Inputs are generated on the device (bench.make_case: the distribution of oracle.synthetic_case)."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu
TOL32 = 1e-5


@pytest.fixture(scope="module")
def mia():
    import torch_assimilate_amd as m
    m.build()
    return m


@pytest.fixture(scope="module")
def eng(mia):
    return mia.LetkfEngine("cuda:0")


def oracle_points(X, gx, ox, Yb, d, c, inf, pts, gamma=None):
    st, yb_h, d_h = X.double().cpu().numpy(), Yb.double().cpu().numpy(), d.double().cpu().numpy()
    gxh, oxh = gx.cpu().numpy(), ox.cpu().numpy()
    core = O.etkf_weights if gamma is None else (lambda a, b, i, g_=gamma: O.ketkf_weights(a, b, lambda x, y: O.rbf_kernel(x, y, g_), i))
    out = []
    for g in pts:
        lo, hi = max(0, int(g) - 200), min(len(gxh), int(g) + 200)       # (the taper's support is a few dozen grid steps)
        sel = (oxh >= gxh[lo]) & (oxh <= gxh[hi - 1])
        w = O.localized_weights(O.abs_distance_1d(gxh[g], oxh[sel]), yb_h[:, sel], d_h[sel], [c], inf, core=core)
        out.append(O.apply_weights(st[:, :, [g]], w[None])[:, :, 0])
    return np.stack(out, axis=-1)


def _report(name, n, per_point, fro, extra=""):
    print("\n[all-points parity] %s: %d grid points against the float64 oracle -- relative Frobenius error %.3e, per-point relative "
          "error max %.3e / median %.3e / 99.9th percentile %.3e%s" % (name, n, fro, per_point.max(), np.median(per_point),
                                                                     np.quantile(per_point, 0.999), extra))


@pytest.mark.parametrize("name,k,stride,c,gamma,seed", [("c2", 40, 2, 10.0, None, 42), ("c4", 80, 1, 16.5, None, 43),
                                                        ("c5", 40, 2, 10.0, 0.5, 43)])
def test_every_grid_point_against_the_oracle_on_the_headline_path(mia, name, k, stride, c, gamma, seed):
    """BASELINE configurations 2, 4 and 5 at their full 1e5 grid points through the path `value` is measured on -- the native step
    driver with steps in flight (ShardedLetkf.submit: bucket index + record packing, the analysis wavefronts localising their own
    tiles where the unions fit 32 slots, lists in memory + two wavefronts per tile / the RBF tile kernel otherwise) -- with EVERY
    grid point compared with the float64 oracle (the reference's per-point localize -> mask / sqrt(rho) -> weights -> transform,
    as interface/test_letkf.py:106-157 compares every grid point): relative Frobenius error <= 1e-5 (north star), and the
    LARGEST per-point relative error reported and bounded."""
    import bench
    import oracle_pool
    if not oracle_pool.started():
        pytest.skip("oracle worker pool not running (it is forked at session start for -m gpu runs)")
    dev = torch.device("cuda:0")
    G = 100000
    X, gx, ox, Yb, d = bench.make_case(G, k, stride, dev, seed=seed)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[c], inf_factor=1.1, rbf_gamma=gamma, max_in_flight=4)
    for _ in range(3):
        first = r.assimilate(X, gx, ox, Yb, d).clone()
    assert r.native_steps >= 2 and r.last_flags_ok()
    pend = [r.submit(X, gx, ox, Yb, d) for _ in range(6)]             # steps in flight, as the bench's timed loop has them
    outs = [h.result().clone() for h in pend]
    for o in outs:
        assert torch.equal(o, first)
    assert r.last_flags_ok() and bool(torch.isfinite(first).all())
    kern = r.dominant_kernel_name
    assert kern.startswith({"c2": "letkf_tile2f_kernel<2, 3, 1, false", "c4": "letkf_tile2p_kernel<5, 5", "c5": "lketkf_tile_kernel<10, 2"}[name]), kern
    r.close()
    ref = oracle_pool.oracle_analysis(X.double().cpu().numpy(), gx.cpu().numpy(), ox.cpu().numpy(), Yb.double().cpu().numpy(),
                                      d.double().cpu().numpy(), c, 1.1, np.arange(G), gamma=gamma)
    got = first.double().cpu().numpy()
    per_point, fro = oracle_pool.per_point_errors(got, ref)
    mean = X.double().cpu().numpy().mean(axis=1, keepdims=True)
    _, fro_inc = oracle_pool.per_point_errors(got - mean, ref - mean)
    _report(name + " (" + kern + ")", G, per_point, fro, "; increments (analysis - prior mean) %.3e" % fro_inc)
    assert fro < TOL32
    assert per_point.max() < TOL32, "worst grid point %d: %.3e" % (int(per_point.argmax()), per_point.max())
    assert fro_inc < 5e-5


@pytest.mark.parametrize("name,k,stride,c", [("c2", 40, 2, 10.0), ("c4", 80, 1, 16.5)])
def test_tile_route_at_full_size(eng, name, k, stride, c):
    """Configs 2 and 4 at 1e5 grid points on the tile route: 64 oracle points, bit-for-bit determinism, a shard that cuts the
    grid anywhere reproduces the full run per point to rounding, the round-2 kernel on per-point lists agrees."""
    import bench
    dev = torch.device("cuda:0")
    G = 100000
    X, gx, ox, Yb, d = bench.make_case(G, k, stride, dev)
    nb = eng.localize(gx, ox, [c])
    assert nb.p_max == (20 if name == "c2" else 63)
    tiles = eng.localize_tiles(gx, ox, [c], nb.p_max)
    assert tiles.stats.tolist() == [nb.p_max, 0]
    rec = eng.pack_split(Yb, d)
    xa, fl, retry = eng.analysis_tiles(X, rec, Yb.shape[1], tiles, 1.1)
    assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0 and bool(torch.isfinite(xa).all())
    xa2 = eng.analysis_tiles(X, rec, Yb.shape[1], eng.localize_tiles(gx, ox, [c], nb.p_max), 1.1)[0]
    assert torch.equal(xa, xa2)
    g0, g1 = 30005, 30117
    part = eng.analysis_tiles(X, rec, Yb.shape[1], eng.localize_tiles(gx, ox, [c], nb.p_max, g0=g0, g1=g1), 1.1)[0]
    ref_part = xa[:, :, g0:g1]
    assert float(((part - ref_part).norm(dim=(0, 1)) / ref_part.norm(dim=(0, 1))).max()) < 2e-6
    xo = eng.analysis(X, Yb, d, nb, 1.1, method="matfun")
    assert float(torch.linalg.norm(xa - xo) / torch.linalg.norm(xo)) < 1e-6
    pts = np.random.RandomState(1).choice(G, 64, replace=False)
    ref = oracle_points(X, gx, ox, Yb, d, c, 1.1, pts)
    got = xa[:, :, torch.as_tensor(pts, device=dev)].double().cpu().numpy()
    assert rel_fro(got, ref) < TOL32
    mean = X.double().mean(dim=1, keepdim=True)[:, :, torch.as_tensor(pts, device=dev)].cpu().numpy()
    assert rel_fro(got - mean, ref - mean) < 5e-5


def test_config5_at_full_size(eng):
    """Config 5 (RBF-kernelised filter, gamma 0.5, k = 40) at 1e5 grid points on the tile route: 64 oracle points, determinism,
    the one-point-per-wavefront kernel agrees, the eigensolver route on a 2000-point shard agrees."""
    import bench
    dev = torch.device("cuda:0")
    G = 100000
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev, seed=43)
    nb = eng.localize(gx, ox, [10.0])
    rec = eng.pack_obs(Yb, d, torch.float32)
    tiles = eng.localize_tiles(gx, ox, [10.0], nb.p_max)
    assert tiles.stats.tolist() == [nb.p_max, 0]
    xa, fl, retry = eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, 0.5)           # the tile route (csrc/lketkf_tile.hip)
    assert int(retry.item()) == 0 and int((fl & 0xff).max().item()) == 0 and bool(torch.isfinite(xa).all())
    assert torch.equal(xa, eng.analysis_tiles_rbf(X, Yb, d, tiles, 1.1, 0.5)[0])
    xp = eng.analysis(X, None, None, nb, 1.1, rec=rec, rbf_gamma=0.5, method="matfun")      # one point per wavefront
    assert float(torch.linalg.norm(xa - xp) / torch.linalg.norm(xp)) < 1e-6
    nb_s = eng.localize(gx, ox, [10.0], g0=50000, g1=52000)
    xe = eng.analysis(X, None, None, nb_s, 1.1, rec=rec, rbf_gamma=0.5, method="eig")
    part = xa[:, :, 50000:52000]
    assert float(torch.linalg.norm(part - xe) / torch.linalg.norm(xe)) < TOL32
    pts = np.random.RandomState(2).choice(G, 64, replace=False)
    ref = oracle_points(X, gx, ox, Yb, d, 10.0, 1.1, pts, gamma=0.5)
    got = xa[:, :, torch.as_tensor(pts, device=dev)].double().cpu().numpy()
    assert rel_fro(got, ref) < TOL32


def test_config3_problem_on_one_gpu(mia):
    """Config 3's problem (G = 1e6 grid points, P = 5e5 observations, k = 40) through the native step driver on ONE GPU: steps in
    flight, 64 oracle points, determinism across steps, and rank 3's block of an 8-rank partition computed alone (what that
    rank of the 8-GPU run computes) equals the same columns of the full run per point to rounding."""
    import bench
    dev = torch.device("cuda:0")
    G = 1000000
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3)
    out = None
    for _ in range(3):
        out = r.assimilate(X, gx, ox, Yb, d)
    assert r.native_steps == 2 and r.last_flags_ok() and bool(torch.isfinite(out).all())
    assert r.dominant_kernel_name.startswith("letkf_tile2")
    pend = [r.submit(X, gx, ox, Yb, d) for _ in range(3)]
    for h in pend:
        assert torch.equal(h.result(), out)
    import oracle_pool
    if oracle_pool.started():          # 1e5 of the 1e6 grid points (every tenth tile's worth, random) against the oracle
        pts = np.sort(np.random.RandomState(3).choice(G, 100000, replace=False))
        ref = oracle_pool.oracle_analysis(X.double().cpu().numpy(), gx.cpu().numpy(), ox.cpu().numpy(), Yb.double().cpu().numpy(),
                                          d.double().cpu().numpy(), 10.0, 1.1, pts)
        got = out[:, :, torch.as_tensor(pts, device=dev)].double().cpu().numpy()
        per_point, fro = oracle_pool.per_point_errors(got, ref)
        _report("c3 problem on one GPU (" + r.dominant_kernel_name + ")", len(pts), per_point, fro)
        assert fro < TOL32 and per_point.max() < TOL32
    else:
        pts = np.random.RandomState(3).choice(G, 64, replace=False)
        ref = oracle_points(X, gx, ox, Yb, d, 10.0, 1.1, pts)
        got = out[:, :, torch.as_tensor(pts, device=dev)].double().cpu().numpy()
        assert rel_fro(got, ref) < TOL32
    from torch_assimilate_amd.sharded import block_partition
    g0, g1 = block_partition(G, 8)[3]
    eng = r.engine
    tiles = eng.localize_tiles(gx, ox, [10.0], 20, g0=g0, g1=g1)
    blk = eng.analysis_tiles(X, eng.pack_split(Yb, d), Yb.shape[1], tiles, 1.1)[0]
    full = out[:, :, g0:g1]
    assert float(((blk - full).norm(dim=(0, 1)) / full.norm(dim=(0, 1))).max()) < 2e-6


def test_mesh_2d_and_eight_state_rows_at_full_size(eng):
    """Off the 1-D line: a 316 x 316 mesh in row-major order, observations at every 2nd point in both dimensions,
    Euclidean distance (the reference's arbitrary dist_func over a real mesh, gaspari_cohn.py:124-134) -- whichever route the
    unions of sixteen consecutive points allow -- and config 2 with m = 8 state rows per grid point (n_var * n_time values,
    interface/base.py:257-278): 64 oracle points each within the north star's tolerance, no flags."""
    import bench
    dev = torch.device("cuda:0")
    X, g, o, Yb, d = bench.make_case_2d(316, 316, 40, 2, dev, seed=44)
    rec = bench.tile_route_case(eng, X, g, o, Yb, d, 2.5, 1.1)
    assert rec["rel_frobenius_error_vs_oracle"] < TOL32 and rec["flags"] == 0, rec
    assert 8 <= rec["p_max"] <= 64
    if rec["route"].startswith("tile lists"):
        assert rec["overflowed_tiles"] == 0 and rec["max_union"] <= rec["union_slots"]
    X1, gx, ox, Yb1, d1 = bench.make_case(100000, 40, 2, dev, seed=45)
    X8 = (X1.repeat(8, 1, 1) * torch.linspace(0.5, 2.0, 8, device=dev)[:, None, None]).contiguous()
    rec8 = bench.tile_route_case(eng, X8, gx, ox, Yb1, d1, 10.0, 1.1)
    assert rec8["route"].startswith("tile lists") and rec8["overflowed_tiles"] == 0
    assert rec8["rel_frobenius_error_vs_oracle"] < TOL32 and rec8["flags"] == 0, rec8


def test_weight_transforms_at_config3_size(mia):
    """_apply_weights at config 3's grid (1e6 points, k = 40): per-point weights on a block of 2e5 points of it, and one weight
    matrix for all points -- the tile kernels of csrc/apply_local.hip with their 32-bit lane offsets at that leading dimension;
    against torch on slices."""
    dev = torch.device("cuda:0")
    eng = mia.LetkfEngine(dev)
    G, k, m = 1000000, 40, 3
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    X = torch.randn((m, k, G), generator=gen, device=dev)
    X[1] += 250.0
    g0, g1 = 700000, 900000
    W = torch.randn((g1 - g0, k, k), generator=gen, device=dev) / k ** 0.5
    out = eng.apply_local_weights(X, W, g0, g1)
    assert out.shape == (m, k, g1 - g0)
    for a, b in ((0, 3000), (g1 - g0 - 3001, g1 - g0)):
        xs = X[:, :, g0 + a:g0 + b].double()
        mean = xs.mean(dim=1, keepdim=True)
        ref = torch.einsum("mig,gij->mjg", xs - mean, W[a:b].double()) + mean
        assert float(torch.linalg.norm(out[:, :, a:b].double() - ref) / torch.linalg.norm(ref)) < 1e-6
    outg = eng.apply_weights(X, W[0])
    for a, b in ((0, 2000), (G - 2001, G)):
        xs = X[:, :, a:b].double()
        mean = xs.mean(dim=1, keepdim=True)
        ref = torch.einsum("mig,ij->mjg", xs - mean, W[0].double()) + mean
        assert float(torch.linalg.norm(outg[:, :, a:b].double() - ref) / torch.linalg.norm(ref)) < 1e-6


def test_size_independent_properties_at_config2(mia):
    """Config 2 at full size through the step driver, checked by properties that need no oracle: (1) scaling the state by a power
    of two scales the analysis by it EXACTLY (every scale factor inside the kernels is a power of two); (2) the order of the
    observations does not matter beyond rounding (tiles rank their unions by observation index: a permutation changes the
    summation order only); (3) grid points out of every observation's reach get the prior ensemble with inflated perturbations,
    mean + sqrt(inf) (x - mean) (core/etkf.py:57-66 with no observation); (4) the analysis is linear in the state rows:
    analysing two variables at once equals analysing each alone."""
    import bench
    dev = torch.device("cuda:0")
    G, k = 100000, 40
    X, gx, ox, Yb, d = bench.make_case(G, k, 2, dev, seed=5)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        base = r.assimilate(X, gx, ox, Yb, d).clone()
    assert r.last_flags_ok() and r.native_steps >= 2
    # (1)
    scaled = r.assimilate(X * 8.0, gx, ox, Yb, d)
    assert torch.equal(scaled, base * 8.0)
    # (2)
    perm = torch.randperm(ox.shape[0], generator=torch.Generator().manual_seed(1)).to(dev)
    shuffled = r.assimilate(X, gx, ox[perm].contiguous(), Yb[:, perm].contiguous(), d[perm].contiguous())
    assert float(torch.linalg.norm(shuffled - base) / torch.linalg.norm(base)) < 2e-6
    # (3) observations only in the first half of the domain: points beyond their reach keep the inflated prior
    half = ox < 0.5 * G
    out_h = r.assimilate(X, gx, ox[half].contiguous(), Yb[:, half].contiguous(), d[half].contiguous())
    far = slice(int(0.5 * G) + 50, G)
    mean = X[:, :, far].mean(dim=1, keepdim=True)
    prior = mean + (1.1 ** 0.5) * (X[:, :, far] - mean)
    assert float(torch.linalg.norm(out_h[:, :, far] - prior) / torch.linalg.norm(prior)) < 1e-6
    r.close()
    # (4)
    X2 = torch.cat([X, torch.flip(X, dims=[2]) * 3.0 + 1.0], dim=0).contiguous()
    r2 = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        both = r2.assimilate(X2, gx, ox, Yb, d)
    assert float(torch.linalg.norm(both[0] - base[0]) / torch.linalg.norm(base[0])) < 2e-6
    r3 = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        second = r3.assimilate(X2[1:].contiguous(), gx, ox, Yb, d)
    assert float(torch.linalg.norm(both[1] - second[0]) / torch.linalg.norm(second[0])) < 2e-6
    r2.close(); r3.close()
