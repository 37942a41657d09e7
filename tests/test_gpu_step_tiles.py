"""GPU tests of the native step driver on the tile route (csrc/sharded_step.hip: bucket index -> tile lists + split records ->
letkf_tile2_kernel): same result as the entry-by-entry engine calls, with the bucket index and with the scan-based one, when
observations leave the bounding box a workspace held (box rebuilt, step repeated), when a cell overflows its bucket (scan-based
index from then on), when tiles need more slots, and with steps in flight."""
import numpy as np
import pytest
import torch

from conftest import rel_fro, set_option
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mia():
    import torch_assimilate_amd as m
    m.build()
    return m


def args_of(case, dev, shift=0.0, scale=1.0):
    return (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"] + shift, device=dev),
            torch.as_tensor(case["obs_x"] + shift, device=dev), torch.as_tensor(case["yb"] * scale, dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"] * scale, dtype=torch.float32, device=dev))


def test_bucket_index_equals_scan_index_and_engine_calls(mia):
    """Tile lists do not depend on how the observations were binned (slots follow the rank of the observation index): the step
    driver with the one-kernel bucket index, with the four-kernel scan index and the entry-by-entry engine route agree bit for
    bit, serial steps and steps in flight."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(3000, 40, 2, seed=5)
    a = args_of(case, dev)
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False).assimilate(*a)
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
    assert rel_fro(ref.cpu().numpy(), oracle) < 1e-5
    for bucket in (1, 0):
        set_option("bucket_index", bucket)
        r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3)
        for _ in range(4):
            out = r.assimilate(*a)
        assert r.native_steps == 3 and torch.equal(out, ref) and r.last_flags_ok()
        assert r.dominant_kernel_name.startswith("letkf_tile2")
        pend = []
        for _ in range(7):
            pend.append(r.submit(*a))
            if len(pend) == 3:
                assert torch.equal(pend.pop(0).result(), ref)
        while pend:
            assert torch.equal(pend.pop(0).result(), ref)
        assert r.native_steps == 10 and not r._scan_index and not r._no_tile_lists


def test_observations_leaving_the_stored_box_rebuild_it(mia):
    """The bucket index reuses the bounding box a workspace holds and checks every observation against it: a network that
    moved (here by 500 grid steps, same sizes) is reported (error bit 8), the box rebuilt and the step repeated -- same result
    as a fresh object; small drifts stay inside the box's one-cell margin and cost nothing."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(2000, 40, 2, seed=6)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        r.assimilate(*args_of(case, dev))
    n0 = r.native_steps
    for shift in (3.0, 500.0, -40.0):
        a = args_of(case, dev, shift=shift)
        out = r.assimilate(*a)
        fresh = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False).assimilate(*a)
        assert torch.equal(out, fresh) and r.last_flags_ok()
    assert r.native_steps >= n0 + 3 and not r._scan_index
    # other radii on the same workspace: the cell grid no longer fits them
    r.radii = [6.0]
    r._p_max_hint = None
    a = args_of(case, dev)
    for _ in range(3):
        out = r.assimilate(*a)
    fresh = mia.ShardedLetkf(dev, 0, 1, radii=[6.0], inf_factor=1.1, native_step=False).assimilate(*a)
    assert torch.equal(out, fresh)


def test_overfull_cell_falls_back_to_the_scan_index(mia):
    """A cluster of 150 observations inside one cell (buckets hold at most 64): error bit 16, the object switches to the
    scan-based index and repeats the step; points near the cluster see more than the tile route's 88 local observations and the
    step takes whatever route covers that -- the result must equal the engine route's."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(3)
    G, k = 1500, 40
    grid = np.arange(G, dtype=np.float64)
    obs = np.concatenate([np.arange(0, G, 4.0), 700.0 + rs.uniform(0, 3.0, size=150)])
    state = rs.normal(size=(1, k, G))
    hx = rs.normal(size=(k, len(obs)))
    yb, d = hx - hx.mean(axis=0), rs.normal(size=len(obs))
    a = (torch.as_tensor(state, dtype=torch.float32, device=dev), torch.as_tensor(grid, device=dev), torch.as_tensor(obs, device=dev),
         torch.as_tensor(yb, dtype=torch.float32, device=dev), torch.as_tensor(d, dtype=torch.float32, device=dev))
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[1.5], inf_factor=1.1, native_step=False).assimilate(*a)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[1.5], inf_factor=1.1)
    for _ in range(4):
        out = r.assimilate(*a)
    assert torch.equal(out, ref) and r.last_flags_ok()


def test_dense_network_gets_more_slots(mia):
    """One observation per grid step: sixteen consecutive points see p_max + 15 observations, more than the default slots of a
    short list -- the step reports the tiles, the object adds a row block (MIA_STEP_TILE_EXTRA) and repeats; afterwards the
    steps run on the tile route."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(1000, 24, 1, seed=9)
    a = args_of(case, dev)
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[1.6], inf_factor=1.1, native_step=False).assimilate(*a)
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 1.6, 1.1)[0]
    assert rel_fro(ref.cpu().numpy(), oracle) < 1e-5
    r = mia.ShardedLetkf(dev, 0, 1, radii=[1.6], inf_factor=1.1)
    for _ in range(4):
        out = r.assimilate(*a)
    assert torch.equal(out, ref) and r._tile_extra == 1 and not r._no_tile_lists
    assert r.dominant_kernel_name.startswith("letkf_tile2")


def test_declined_points_are_redone_from_lists_built_then(mia):
    """Strong observations: the tile kernel declines points (MIA_FLAG_RETRY); phase 1 of the step builds float32 records and
    per-point lists over a scan-based index and the eigensolver kernel redoes exactly those points -- same result as the engine
    route, whose retry uses the exact lists."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(1200, 40, 2, seed=11)
    a = args_of(case, dev, scale=12.0)
    plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False)
    ref = plain.assimilate(*a)
    assert plain.last_retries > 0
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"] * 12.0, case["d"] * 12.0, 10.0, 1.1)[0]
    assert rel_fro(ref.cpu().numpy(), oracle) < 1e-5
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(4):
        out = r.assimilate(*a)
    assert torch.equal(out, ref) and r.last_retries == plain.last_retries and r.last_flags_ok()


def test_geometry_epoch_reuses_the_tile_lists(mia):
    """``geometry_id``: steps of one geometry epoch rebuild only the split records (MIA_STEP_REUSE_LISTS) and reproduce the full
    rebuild bit for bit -- serially and with steps in flight (every pipeline slot builds its lists once), with new perturbations
    and innovations every step; a new id (moved observations) rebuilds; declined points are still redone."""
    import bench
    dev_ = torch.device("cuda:0")
    G, k = 40000, 40
    X, gx, ox, Yb, d = bench.make_case(G, k, 2, dev_, seed=9)
    full = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=4)
    epoch = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=4)
    gen = torch.Generator(device=dev_)
    gen.manual_seed(3)
    steps = []
    for i in range(10):
        Yi = Yb * (1.0 + 0.1 * i) + 0.01 * torch.randn(Yb.shape, generator=gen, device=dev_)
        di = d + 0.1 * torch.randn(d.shape, generator=gen, device=dev_)
        if i == 7:
            Yi[:, 100:140] *= 14.0                                    # strong observations: declined points in this step
        steps.append((Yi.contiguous(), di.contiguous()))
    ref = [full.assimilate(X, gx, ox, Yi, di).clone() for Yi, di in steps]
    # serial
    got = [epoch.assimilate(X, gx, ox, Yi, di, geometry_id="a").clone() for Yi, di in steps]
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    # in flight
    pend = [epoch.submit(X, gx, ox, Yi, di, geometry_id="a") for Yi, di in steps[:4]]
    out = [h.result().clone() for h in pend]
    pend = [epoch.submit(X, gx, ox, Yi, di, geometry_id="a") for Yi, di in steps[4:]]
    out += [h.result().clone() for h in pend]
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert epoch.reused_steps >= 10
    # moved observations under a new id: rebuilt, equal to the plain runner on the new geometry
    ox2 = (ox + 0.25).contiguous()
    r2 = full.assimilate(X, gx, ox2, *steps[0]).clone()
    g2 = epoch.assimilate(X, gx, ox2, *steps[0], geometry_id="b").clone()
    g3 = epoch.assimilate(X, gx, ox2, *steps[0], geometry_id="b").clone()
    assert torch.equal(g2, r2) and torch.equal(g3, r2)


def test_results_do_not_depend_on_the_number_of_preparation_streams(mia):
    """Steps in flight rotate through `prep_streams` preparation streams (plain streams: the hardware-queue probing of round 3
    is gone): results do not depend on how many there are."""
    import bench
    dev_ = torch.device("cuda:0")
    X, gx, ox, Yb, d = bench.make_case(20000, 40, 2, dev_, seed=4)
    outs = []
    for n, extra in ((5, {}), (2, {}), (1, {}), (0, dict(analysis_streams=2, fuse_tile_lists=True)), (0, {})):
        # (0: no preparation stream at all -- a step's launches back to back on its analysis stream; with the fused kernel and two
        #  analysis streams: "two step streams")
        r = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=4, prep_streams=n, **extra)
        r.assimilate(X, gx, ox, Yb, d)
        pend = [r.submit(X, gx, ox, Yb, d) for _ in range(4)]
        res = [h.result() for h in pend]
        assert all(torch.equal(res[0], x) for x in res[1:])
        outs.append(res[0].clone())
        r.close()
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_sharded_output_keeps_the_block(mia):
    """gather=False: rank r of a world of 4 (one process, one GPU: no communicator is needed) returns ITS BLOCK of the analysis
    from the exact-list call, the one-call native step (MIA_STEP_NO_GATHER on a partition-only communicator) and steps in
    flight -- equal to the same columns of the single-rank run per point to rounding (the tiles are cut elsewhere)."""
    import bench
    from torch_assimilate_amd.sharded import block_partition
    dev_ = torch.device("cuda:0")
    G = 30011
    X, gx, ox, Yb, d = bench.make_case(G, 40, 2, dev_, seed=9)
    full = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1)
    full.assimilate(X, gx, ox, Yb, d)
    ref = full.assimilate(X, gx, ox, Yb, d)
    for rank in (0, 2, 3):
        g0, g1 = block_partition(G, 4)[rank]
        r = mia.ShardedLetkf(dev_, rank, 4, radii=[10.0], inf_factor=1.1, gather=False, max_in_flight=3)
        first = r.assimilate(X, gx, ox, Yb, d)
        second = r.assimilate(X, gx, ox, Yb, d)
        assert first.shape == (1, 40, g1 - g0) and second.shape == first.shape
        assert r.native_steps == 1 and r.last_flags_ok()
        pend = [r.submit(X, gx, ox, Yb, d) for _ in range(4)]
        for h in pend:
            assert torch.equal(h.result(), second)
        part = ref[:, :, g0:g1]
        for got in (first, second):
            assert float(((got - part).norm(dim=(0, 1)) / part.norm(dim=(0, 1))).max()) < 2e-6
        r.close()


def test_wavefronts_localising_their_own_tiles_equal_lists_in_memory(mia):
    """``fuse_tile_lists`` (option tile_fused, csrc/letkf_tile2f.hip): the analysis wavefronts run the list kernel's code on their own
    tile -- bit for bit the analysis from lists in memory, one step at a time, with steps in flight (the two per-cell count arrays
    of a workspace alternate: the launch that reads one clears the other), when the two kinds of step follow each other on
    the same workspaces, with new observation positions every step, and for 2-D / 3-D coordinates with two radii."""
    import bench
    dev_ = torch.device("cuda:0")
    X, gx, ox, Yb, d = bench.make_case(30000, 40, 2, dev_, seed=12)
    lists = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3, fuse_tile_lists=False)
    fused = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3, fuse_tile_lists=True)
    mixed = mia.ShardedLetkf(dev_, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3)
    assert fused.dominant_kernel_name.startswith("letkf_tile2f_kernel")
    gen = torch.Generator(device=dev_)
    gen.manual_seed(5)
    geoms = [(ox + 0.3 * torch.rand(ox.shape, generator=gen, device=dev_, dtype=ox.dtype)).contiguous() for _ in range(9)]
    ref = [lists.assimilate(X, gx, o, Yb, d).clone() for o in geoms]
    assert lists.last_flags_ok()
    got = [fused.assimilate(X, gx, o, Yb, d).clone() for o in geoms]
    assert fused.last_flags_ok() and fused.native_steps >= 8
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    for runner in (fused, mixed):
        pend, out = [], []
        for o in geoms:
            pend.append(runner.submit(X, gx, o, Yb, d))
            if len(pend) == 3:
                out.append(pend.pop(0).result().clone())
        out += [h.result().clone() for h in pend]
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        # ... and one at a time again on the workspaces the steps in flight used
        assert torch.equal(runner.assimilate(X, gx, geoms[4], Yb, d), ref[4])
        assert torch.equal(runner.assimilate(X, gx, geoms[5], Yb, d), ref[5])
    # counters: the longest list comes from the analysis launch now
    assert fused._p_max_hint == lists._p_max_hint
    for r in (lists, fused, mixed):
        r.close()
    # 2-D and 3-D networks with short lists (unions of at most 32 slots), two radii
    rng = np.random.default_rng(8)
    for nc, shape, radii, groups in ((2, (60, 50), [1.3, 1.1], [0, 1]), (3, (14, 15, 16), [1.2, 1.05], [0, 0, 1])):
        axes = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        grid = np.stack([a.ravel() for a in axes], axis=1)
        Gn, k = grid.shape[0], 24
        pick = rng.permutation(Gn)[: Gn // 3]
        obs = grid[np.sort(pick)] + rng.uniform(-0.2, 0.2, (pick.size, nc))
        Xn = torch.as_tensor(rng.standard_normal((1, k, Gn)), dtype=torch.float32, device=dev_)
        Yn = torch.as_tensor(rng.standard_normal((k, pick.size)), dtype=torch.float32, device=dev_)
        Yn = Yn - Yn.mean(dim=0, keepdim=True)
        dn = torch.as_tensor(rng.standard_normal(pick.size), dtype=torch.float32, device=dev_)
        gt, ot = torch.as_tensor(grid, device=dev_), torch.as_tensor(obs, device=dev_)
        outs = {}
        for fz in (False, True):
            r = mia.ShardedLetkf(dev_, 0, 1, radii=radii, coord_group=groups, inf_factor=1.05, fuse_tile_lists=fz)
            for _ in range(3):
                o = r.assimilate(Xn, gt, ot, Yn, dn)
            outs[fz] = (o.clone(), r._no_tile_lists, r._tile_extra, r.last_flags_ok())
            r.close()
        assert outs[True][1:] == outs[False][1:], (nc, outs[True][1:], outs[False][1:])
        assert torch.equal(outs[True][0], outs[False][0]), nc


@pytest.mark.parametrize("k,stride,radius", [(8, 4, 6.0), (16, 2, 4.0), (24, 2, 10.0), (50, 2, 10.0), (70, 2, 10.0), (96, 2, 10.0),
                                             (40, 7, 10.0), (33, 40, 10.0)])
def test_fused_localisation_over_ensemble_sizes(mia, k, stride, radius):
    """Every member-block count (k = 8 .. 96) and both union sizes (one or two row blocks of slots; hardly any observation at all
    with a stride of 40) of letkf_tile2f_kernel against the oracle and, bit for bit, against lists in memory."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(1500, k, stride, seed=20 + k)
    a = args_of(case, dev)
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], radius, 1.1)[0]
    outs = {}
    for fz in (False, True):
        r = mia.ShardedLetkf(dev, 0, 1, radii=[radius], inf_factor=1.1, fuse_tile_lists=fz)
        for _ in range(3):
            out = r.assimilate(*a)
        assert r.last_flags_ok() and r.native_steps == 2
        outs[fz] = (out.clone(), r._no_tile_lists, r._tile_extra, r.last_p_max)
        r.close()
    assert outs[True][1:] == outs[False][1:]
    assert torch.equal(outs[True][0], outs[False][0])
    assert rel_fro(outs[True][0].cpu().numpy(), oracle) < 1e-5


def test_more_workspaces_than_the_library_keeps_state_for(mia):
    """The library keeps per-workspace state (which of the two per-cell count arrays is current, what the tile lists were built
    for) for a bounded number of workspaces; one that fell out of the table is treated as unknown -- its next step clears the
    index tables whatever the caller says about it -- instead of binning into an array the table no longer knows to be dirty."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(700, 24, 2, seed=31)
    a = args_of(case, dev)
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[8.0], inf_factor=1.1, native_step=False).assimilate(*a)
    runners = [mia.ShardedLetkf(dev, 0, 1, radii=[8.0], inf_factor=1.1, max_in_flight=2) for _ in range(72)]
    for sweep in range(4):
        for r in runners:
            out = r.assimilate(*a)
            assert torch.equal(out, ref), sweep
    assert all(r.last_flags_ok() for r in runners) and runners[0].native_steps >= 3
    for r in runners:
        r.close()


def test_random_geometries_against_the_oracle(mia):
    """Twenty seeded random 1-D problems -- ensemble size, network density, radius, inflation, state rows, magnitudes of state and
    observations, ragged grid lengths -- through the step driver (one step at a time: the fused kernel where it has the shape,
    every other route where it does not) against the ORACLE (not against another route of this library), north-star bound 1e-5."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(2024)
    worst = 0.0
    for case_no in range(20):
        k = int(rs.choice([5, 8, 13, 24, 40, 57]))
        stride = int(rs.choice([1, 2, 3, 7]))
        radius = float(rs.choice([2.5, 6.0, 10.0]))
        inf = float(rs.choice([1.0, 1.1, 1.5]))
        m = int(rs.choice([1, 1, 3]))
        G = int(rs.choice([97, 256, 401]))
        sx, sy = float(10.0 ** rs.uniform(-2, 2)), float(10.0 ** rs.uniform(-1, 1))
        case = O.synthetic_case(G, k, stride, seed=1000 + case_no, m=m)
        state = case["state"] * sx + rs.normal(size=(m, 1, 1)) * 50.0 * sx
        yb, d = case["yb"] * sy, case["d"] * sy
        a = (torch.as_tensor(state, dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
             torch.as_tensor(case["obs_x"], device=dev), torch.as_tensor(yb, dtype=torch.float32, device=dev),
             torch.as_tensor(d, dtype=torch.float32, device=dev))
        r = mia.ShardedLetkf(dev, 0, 1, radii=[radius], inf_factor=inf)
        for _ in range(3):
            out = r.assimilate(*a)
        oracle = O.letkf_analysis(state, case["grid_x"], case["obs_x"], yb, d, radius, inf)[0]
        err = rel_fro(out.cpu().numpy(), oracle)
        # the stricter figure: the increments (analysis minus the prior mean), relative to their own size
        prior_mean = state.mean(axis=1, keepdims=True)
        err_inc = rel_fro(out.cpu().numpy() - prior_mean, oracle - prior_mean)
        worst = max(worst, err)
        assert err < 1e-5 and err_inc < 1e-4, (case_no, k, stride, radius, inf, m, G, sx, sy, err, err_inc)
        r.close()
    assert worst < 1e-5


def test_fused_localisation_with_several_state_rows(mia):
    """Several state rows per grid point through the fused kernel's row loop: bit for bit the analysis from lists in memory, and
    against the oracle."""
    dev = torch.device("cuda:0")
    for k, m in ((40, 5), (24, 2), (70, 3)):
        case = O.synthetic_case(1300, k, 2, seed=40 + k, m=m)
        a = args_of(case, dev)
        oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
        outs = {}
        for fz in (False, True):
            r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, fuse_tile_lists=fz)
            for _ in range(3):
                out = r.assimilate(*a)
            assert r.last_flags_ok() and r.native_steps == 2
            outs[fz] = out.clone()
            r.close()
        assert torch.equal(outs[True], outs[False]), (k, m)
        assert rel_fro(outs[True].cpu().numpy(), oracle) < 1e-5


def test_the_steady_state_submit_path_equals_the_general_one(mia):
    """ShardedLetkf.submit takes a short path (_submit_fast) once a steady state exists: same argument block, same streams, same flags.
    Steps through it -- new input tensors every step, a timed step, inputs the short path must refuse (another dtype, another
    shape, a non-contiguous state), observations that leave the stored box (step repeated on the general path) -- return what the
    general path returns bit for bit, and the short path is really taken."""
    dev = torch.device("cuda:0")
    cases = [O.synthetic_case(4000, 40, 2, seed=60 + i) for i in range(3)]
    args = [args_of(c, dev) for c in cases]
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=4)
    ref._submit_fast = lambda *a, **k: None                       # the general path for every step
    want = [ref.assimilate(*a).clone() for a in args]
    want_shift = ref.assimilate(*args_of(cases[0], dev, shift=5000.0)).clone()
    ref.close()
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=4)
    taken = []
    orig = r._submit_fast

    def spy(f, *a):
        h = orig(f, *a)
        taken.append(h is not None)
        return h
    r._submit_fast = spy
    for a in args:
        r.assimilate(*a)
    pend = []
    for i in range(13):
        if i == 5:
            r.time_next_step()
        pend.append((i % 3, r.submit(*[t.clone() for t in args[i % 3]])))
    for j, h in pend:
        assert torch.equal(h.result(), want[j]), j
    assert sum(taken) >= 8, taken                                  # (the first submissions set the state up)
    assert len(r.kernel_timings) == 1 and r.kernel_timings[0][0].elapsed_time(r.kernel_timings[0][1]) > 0.0
    n0 = len(taken)
    X, g, o, Yb, d = args[1]
    refused = [
        (X.double(), g, o, Yb, d), (X, g.float(), o, Yb, d), (X, g, o, Yb.double(), d),
        (X.transpose(1, 2).contiguous().transpose(1, 2), g, o, Yb, d),
    ]
    for a in refused:
        out = r.submit(*a).result()
        assert float(torch.linalg.norm(out.float() - want[1]) / torch.linalg.norm(want[1])) < 1e-5
    assert not any(taken[n0:]), taken[n0:]
    # a smaller problem, then the first one again: the state is re-recorded, the results stay right
    small = args_of(O.synthetic_case(1500, 40, 2, seed=70), dev)
    oracle = O.letkf_analysis(*[O.synthetic_case(1500, 40, 2, seed=70)[k] for k in ("state", "grid_x", "obs_x", "yb", "d")], 10.0, 1.1)[0]
    for _ in range(3):
        out = r.submit(*small).result()
    assert rel_fro(out.cpu().numpy(), oracle) < 1e-5
    for _ in range(3):
        assert torch.equal(r.submit(*args[2]).result(), want[2])
    # observations outside the box the workspaces hold: noticed at collection, the step is repeated (general path), right result
    outs = [r.submit(*args_of(cases[0], dev, shift=5000.0)) for _ in range(3)]
    for h in outs:
        assert torch.equal(h.result(), want_shift)
    assert r.last_flags_ok()
    r.close()


def test_the_steady_state_serial_path_equals_the_general_one(mia):
    """ShardedLetkf.assimilate takes a short path (_run_fast: one library call per step) once a steady state exists for steps taken one
    at a time: same results bit for bit as the general path, for new input tensors every step; inputs it must refuse, another problem
    size, steps in flight in between, observations leaving the stored box and declined points all fall back to the general path
    and come out right."""
    dev = torch.device("cuda:0")
    cases = [O.synthetic_case(4000, 40, 2, seed=80 + i) for i in range(3)]
    args = [args_of(c, dev) for c in cases]
    ref = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    ref._run_fast = lambda *a, **k: None
    want = [ref.assimilate(*a).clone() for a in args]
    want_shift = ref.assimilate(*args_of(cases[0], dev, shift=5000.0)).clone()
    want_strong = ref.assimilate(*args_of(cases[1], dev, scale=300.0)).clone()        # (strong observations: declined points, float64 redo)
    ref.close()
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    taken = []
    orig = r._run_fast

    def spy(f, *a):
        out = orig(f, *a)
        taken.append(out is not None)
        return out
    r._run_fast = spy
    for rep in range(4):
        for j, a in enumerate(args):
            assert torch.equal(r.assimilate(*[t.clone() for t in a]), want[j]), (rep, j)
    assert sum(taken) >= 9, taken
    n0 = len(taken)
    X, g, o, Yb, d = args[1]
    out = r.assimilate(X.double(), g, o, Yb, d)
    assert float(torch.linalg.norm(out.float() - want[1]) / torch.linalg.norm(want[1])) < 1e-5
    pend = [r.submit(*args[2]) for _ in range(3)]                 # steps in flight, then one at a time again
    for h in pend:
        assert torch.equal(h.result(), want[2])
    assert torch.equal(r.assimilate(*args[0]), want[0])
    assert torch.equal(r.assimilate(*args_of(cases[0], dev, shift=5000.0)), want_shift)      # box rebuilt, step repeated
    assert torch.equal(r.assimilate(*args_of(cases[0], dev, shift=5000.0)), want_shift)
    strong = args_of(cases[1], dev, scale=300.0)
    assert torch.equal(r.assimilate(*strong), want_strong)
    assert torch.equal(r.assimilate(*strong), want_strong)
    assert r.last_retries >= 0 and r.last_flags_ok()
    small = args_of(O.synthetic_case(1500, 40, 2, seed=90), dev)
    oracle = O.letkf_analysis(*[O.synthetic_case(1500, 40, 2, seed=90)[k] for k in ("state", "grid_x", "obs_x", "yb", "d")], 10.0, 1.1)[0]
    for _ in range(3):
        out = r.assimilate(*small)
    assert rel_fro(out.cpu().numpy(), oracle) < 1e-5
    assert torch.equal(r.assimilate(*args[2]), want[2])
    assert any(taken[n0:])
    r.close()


def test_non_finite_points_are_reported_in_the_status_word(mia):
    """The fused kernel says in the step's status word (MIA_STEP_STATUS_NONFINITE) whether any grid point carries MIA_FLAG_NONFINITE, so
    that a caller need not scan the per-point flags after every step: clean inputs -> summary 0; a state column beyond the float
    range -> summary 4 and exactly that point flagged; the drop-in class warns from the summary."""
    import warnings
    dev = torch.device("cuda:0")
    case = O.synthetic_case(3000, 40, 2, seed=95)
    X, g, o, Yb, d = args_of(case, dev)
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        r.assimilate(X, g, o, Yb, d)
    assert r.last_flags_summary == 0 and r.last_flags_ok()
    Xb = X.clone()
    Xb[0, 3, 1234] = 3.0e38
    Xb[0, 4, 1234] = -3.0e38
    for _ in range(2):                                   # (general path, then the short one)
        out = r.assimilate(Xb, g, o, Yb, d)
        assert r.last_flags_summary == 4
        fl = (r._last_flags & 0xff).cpu().numpy()
        assert fl[1234] & 4 and int((fl != 0).sum()) == 1
        assert not bool(torch.isfinite(out[0, :, 1234]).all())
    h = [r.submit(Xb, g, o, Yb, d) for _ in range(3)]
    for x in h:
        x.result()
        assert r.last_flags_summary == 4
    r.assimilate(X, g, o, Yb, d)
    assert r.last_flags_summary == 0
    r.close()
    a = mia.LETKF(mia.GaspariCohn(10.0, mia.AbsoluteDistance()), inf_factor=1.1, dtype=torch.float32)
    gx, ox = g[:, None], o[:, None]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(3):
            a.analyse_arrays(X, Yb, d, grid_coords=gx, obs_coords=ox)
    with pytest.warns(RuntimeWarning, match="non-finite"):
        a.analyse_arrays(Xb, Yb, d, grid_coords=gx, obs_coords=ox)
