"""GPU tests of the host-side mirror of the reference interface; they read like the reference's
tests/unit_tests/{core/test_etkf.py, interface/test_letkf.py, interface/test_lketkf.py}."""
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
    W = mia.LETKF(dtype=dtype).estimate_weights_arrays(g["yb"], g["d"], **kw)          # (grid, ensemble, ensemble_new)
    assert tuple(W.shape) == (40, 10, 10) and rel_fro(W[7].cpu().numpy(), g["weights_global_1p0"]) < 5 * tol
    # ... and for far more observations than one workgroup could hold (the reference repeats the global solve G times)
    rs = np.random.RandomState(2)
    yb_big = rs.normal(size=(10, 6000)) * 0.1
    yb_big -= yb_big.mean(axis=0)
    d_big = rs.normal(size=6000) * 0.1
    xa_b = mia.LETKF(dtype=dtype).analyse_arrays(state, yb_big, d_big, grid_coords=g["grid"][:, None],
                                                obs_coords=np.zeros((6000, 1)))
    ref_b, _ = O.etkf_analysis(state, yb_big, d_big, 1.0)
    assert rel_fro(xa_b.cpu().numpy(), ref_b) < tol
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
    a32 = mia.LKETKF(mia.RBFKernel(0.5), localization=loc, inf_factor=1.1, dtype=torch.float32)
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


def _emulated_peer_comm(mia, runner, rank, ref_full, G, chunks):
    """mia_comm_create_custom communicator that plays a 2-rank world on one GPU: the all-gather callback puts
    this rank's piece into its slot and the peer's piece -- cut from a reference analysis of the whole grid --
    into the other slot, on the stream the driver passes."""
    import ctypes as C
    from torch_assimilate_amd import _cabi
    world = 2
    m, k = ref_full.shape[0], ref_full.shape[1]
    n = (G + world - 1) // world
    nc = ((n + chunks - 1) // chunks + 15) // 16 * 16
    peer = 1 - rank
    pieces = []
    for c in range(chunks):
        lo = min(G, peer * n + c * nc)
        hi = min(G, peer * n + min(n, (c + 1) * nc))
        t = torch.zeros((m, k, nc), dtype=torch.float32, device=ref_full.device)
        t[:, :, :hi - lo] = ref_full[:, :, lo:hi]
        pieces.append(t.contiguous())
    calls = {"n": 0}

    def allgather(ctx, send, recv, nbytes, stream):
        ws = next(sl["ws"] for sl in runner._native["slots"]           # the slot (pipeline stage) this step runs in
                  if sl.get("ws") is not None and sl["ws"].data_ptr() <= send < sl["ws"].data_ptr() + sl["ws"].numel())
        c = calls["n"] % chunks
        calls["n"] += 1
        data = m * k * nc * 4
        assert nbytes == data + 16                         # piece + counter trailer
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            off_r = recv - ws.data_ptr()
            off_s = send - ws.data_ptr()
            slots = ws[off_r:off_r + 2 * nbytes].view(2, nbytes)
            slots[rank].copy_(ws[off_s:off_s + nbytes])
            slots[peer, :data].view(torch.float32).view(m, k, nc).copy_(pieces[c])
            slots[peer, data:].zero_()
        return 0

    def allreduce(ctx, buf, cnt, stream):
        return 0

    cb = (_cabi.ALLGATHER_FN(allgather), _cabi.ALLREDUCE_MAX_I32_FN(allreduce))
    handle = C.c_void_p()
    _cabi.check(_cabi.lib().mia_comm_create_custom(rank, world, C.cast(cb[0], C.c_void_p), C.cast(cb[1], C.c_void_p),
                                                   None, C.byref(handle)), "mia_comm_create_custom")
    return handle, cb, calls


_LAYOUT_CASES = [(G, chunks, strong, rank, signal, split)
                 for signal, split in (("1", 0), ("0", 0), ("1", 1)) for rank in (0, 1)
                 for G, chunks, strong in ((2000, 4, False), (1999, 3, False), (1203, 1, False), (1000, 4, True), (40000, 8, False))
                 if not split or (rank == 1 and G in (1999, 1000))]       # split-precision route: the ragged and the redo case


@pytest.mark.parametrize("G,chunks,strong,rank,signal,split", _LAYOUT_CASES)
def test_native_step_driver_two_rank_layout_on_one_gpu(mia, G, chunks, strong, rank, signal, split, monkeypatch):
    """mia_letkf_sharded_step_f32 as rank 0 and as rank 1 of a 2-rank world (peer emulated, see above): block /
    chunk partition, per-chunk exchange on the side stream, placement into (m, k, G) incl. ragged tails and the
    unaligned scalar path, and the phase-1 redo after declined points.  Must reproduce the single-rank result -- bit for
    bit on the f32 products (split = 0: a point's result does not depend on its tile); to rounding on the split-precision
    products, whose operand scale is the tile's (a rank's block starts its tiles elsewhere than the full run)."""
    set_option("segment_signal", int(signal))   # "1": one segmented launch + device-side segment counters
    set_option("tile_split", split)
    same = torch.equal if not split else (lambda a, b: float((a - b).norm() / b.norm()) < 1e-6)
    dev = torch.device("cuda:0")
    case = O.synthetic_case(G, 40, 2)
    scale = 12.0 if strong else 1.0
    args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
            torch.as_tensor(case["obs_x"], device=dev),
            torch.as_tensor(case["yb"] * scale, dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"] * scale, dtype=torch.float32, device=dev))
    plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False)
    ref = plain.assimilate(*args)
    single = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):                                 # call 1: exact lists; calls 2, 3: the native driver
        out1 = single.assimilate(*args)
    assert single.native_steps == 2 and torch.equal(out1, ref) and single.last_flags_ok()
    if strong:
        assert single.last_retries == plain.last_retries > 0

    runner = mia.ShardedLetkf(dev, rank, 2, radii=[10.0], inf_factor=1.1, comm_chunks=chunks)
    runner._p_max_hint = plain._p_max_hint
    handle, keep, calls = _emulated_peer_comm(mia, runner, rank, ref, G, chunks)
    import ctypes as C
    from torch_assimilate_amd import _cabi as cabi_
    xstream = torch.cuda.Stream(device=dev)     # placement of the gathered pieces off the exchange stream (steps in flight)
    cabi_.check(cabi_.lib().mia_comm_set_place_stream(handle, C.c_void_p(xstream.cuda_stream)), "mia_comm_set_place_stream")
    runner._native = dict(comm=handle, custom=True, stream=torch.cuda.Stream(device=dev), xstream=xstream, slots=[{}, {}, {}])
    try:
        for _ in range(2):
            out = runner.assimilate(*args)
            torch.cuda.synchronize()
            assert same(out, ref)
            assert runner.last_flags_ok()
        assert runner.native_steps == 2
        assert calls["n"] == 2 * chunks * (2 if strong else 1)
        # the same steps software-pipelined (submit i+1 before collecting i): both slots, one exchange stream
        if not strong:          # (the emulated peer's callback serves pieces in submission order)
            pend = []
            for _ in range(6):
                pend.append(runner.submit(*args))
                if len(pend) == 3:                          # three steps in flight
                    assert same(pend.pop(0).result(), ref)
            while pend:
                assert same(pend.pop(0).result(), ref)
            assert runner.last_flags_ok() and runner.native_steps == 8
    finally:
        from torch_assimilate_amd import _cabi
        _cabi.lib().mia_comm_destroy(handle)
        runner._native = None


def test_native_step_driver_through_real_rccl_single_rank(mia):
    """The library-owned RCCL communicator (unique id over torch.distributed, ncclCommInitRank, per-chunk
    ncclAllGather on the side stream, ncclAllReduce of the counters) on a one-rank world."""
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    own = not dist.is_initialized()
    if own:
        import os, socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        G = 3000
        case = O.synthetic_case(G, 40, 2)
        args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
                torch.as_tensor(case["obs_x"], device=dev), torch.as_tensor(case["yb"], dtype=torch.float32, device=dev),
                torch.as_tensor(case["d"], dtype=torch.float32, device=dev))
        ref = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False).assimilate(*args)
        r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, comm_chunks=4)
        r._force_comm = True
        for _ in range(4):
            out = r.assimilate(*args)
        torch.cuda.synchronize()
        assert r.native_steps == 3 and r._native["comm"] is not None
        assert torch.equal(out, ref) and r.last_flags_ok()
        pend = None                                   # pipelined: two steps in flight over the real RCCL calls
        for _ in range(6):
            h = r.submit(*args)
            if pend is not None:
                assert torch.equal(pend.result(), ref)
            pend = h
        assert torch.equal(pend.result(), ref) and r.native_steps == 9
        r.close()
    finally:
        if own:
            dist.destroy_process_group()


def test_pipelined_steps_match_serial_steps(mia):
    """ShardedLetkf.submit: step i+1 is enqueued before step i is collected (two slots, two streams).  Different
    inputs per step, a step whose strong observations make the matfun kernel decline points (phase-1 redo on the
    slot's stream), and a step whose denser observations break the assumed list bound (all in-flight steps are
    drained and the step is redone with exact lists): every result must equal the serial runner's, bit for bit."""
    dev = torch.device("cuda:0")
    G = 3000
    case = O.synthetic_case(G, 40, 2)
    rs = np.random.RandomState(4)
    dense = O.synthetic_case(G, 40, 1, seed=5)           # observation at every grid point: ~39 local obs instead of 20

    def inputs(i):
        c, scale = case, 1.0
        if i == 3:
            scale = 12.0
        if i == 5:
            c = dense
        x = torch.as_tensor(c["state"] + 0.01 * i, dtype=torch.float32, device=dev)
        return (x, torch.as_tensor(c["grid_x"], device=dev), torch.as_tensor(c["obs_x"], device=dev),
                torch.as_tensor(c["yb"] * scale, dtype=torch.float32, device=dev),
                torch.as_tensor(c["d"] * scale, dtype=torch.float32, device=dev))

    serial = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    refs = [serial.assimilate(*inputs(i)).clone() for i in range(8)]
    for depth in (2, 3):
        piped = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=depth)
        outs, pend = [], []
        for i in range(8):
            pend.append(piped.submit(*inputs(i)))
            if len(pend) == depth:
                outs.append(pend.pop(0).result())
        last = pend[-1]
        while pend:
            outs.append(pend.pop(0).result())
        assert last.result() is outs[-1]                     # idempotent
        for i, (a, b) in enumerate(zip(outs, refs)):
            assert torch.equal(a, b), (depth, i)
        assert piped.last_flags_ok() and not piped._in_flight
        assert piped.native_steps >= 4


# ---------------------------------------------------------------- observation-space preparation
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 2e-6)])
@pytest.mark.parametrize("k,P", [(10, 40), (40, 1000), (7, 33), (80, 257)])
def test_obs_space_uncorrelated_vs_oracle(mia, dtype, tol, k, P):
    rnd = np.random.RandomState(k * 1000 + P)
    hx = rnd.normal(size=(k, P)) + 3.0
    y = rnd.normal(size=P) + 3.0
    var = rnd.uniform(0.2, 4.0, size=P)
    yb_ref, d_ref = O.obs_space_uncorr(hx, y, var)
    eng = mia.LetkfEngine("cuda:0")
    yb, d, rec = eng.obs_space(hx, y, var=var, dtype=dtype, want_rec=True)
    assert rel_fro(yb.cpu().numpy(), yb_ref) < tol and rel_fro(d.cpu().numpy(), d_ref) < tol
    # the records emitted by the same kernel are the packed form of (Yb, d), bit for bit
    assert torch.equal(rec, eng.pack_obs(yb, d, dtype))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
@pytest.mark.parametrize("k,P", [(10, 40), (12, 31), (20, 200), (8, 97)])
def test_obs_space_correlated_vs_oracle(mia, dtype, tol, k, P):
    rnd = np.random.RandomState(k * 1000 + P)
    hx = rnd.normal(size=(k, P)) + 1.0
    y = rnd.normal(size=P) + 1.0
    a = rnd.normal(size=(P, P))
    xg = np.arange(P)
    cov = np.exp(-np.abs(xg[:, None] - xg[None, :]) / 3.0) + 0.05 * (a @ a.T) / P + 0.1 * np.eye(P)
    yb_ref, d_ref = O.obs_space_corr(hx, y, cov)
    eng = mia.LetkfEngine("cuda:0")
    yb, d, rec = eng.obs_space(hx, y, cov=cov, dtype=dtype, want_rec=True)
    assert rel_fro(yb.cpu().numpy(), yb_ref) < tol and rel_fro(d.cpu().numpy(), d_ref) < tol
    assert torch.equal(rec, eng.pack_obs(yb, d, dtype))
    bad = cov.copy(); bad[P // 2, P // 2] = -1.0
    with pytest.raises(ValueError):
        eng.obs_space(hx, y, cov=bad, dtype=dtype)


def test_obs_space_reference_fixture_and_stacking(mia, golden):
    """The reference's own correlated-R fixture (yb, d generated by importing the reference, tools/gen_golden.py)
    and the stacking of two subsets (one correlated, one uncorrelated) into one observation axis."""
    g = golden("g6_reference_fixture_letkf.npz")
    ti = int(g["time_index"])
    hx, y, cov = g["state"][0, ti], g["obs"][ti], g["cov"]
    f = mia.ETKF(inf_factor=1.1, dtype=torch.float64)
    d, yb = f.get_obs_space_variables([hx], [y], covariances=[cov])
    np.testing.assert_allclose(yb.cpu().numpy(), g["yb"], atol=1e-12)
    np.testing.assert_allclose(d.cpu().numpy(), g["d"], atol=1e-12)
    rnd = np.random.RandomState(3)
    hx2, y2, var2 = rnd.normal(size=(hx.shape[0], 17)), rnd.normal(size=17), rnd.uniform(0.5, 2.0, size=17)
    d_s, yb_s = f.get_obs_space_variables([hx, hx2], [y, y2], variances=[None, var2], covariances=[cov, None])
    yb2, d2 = O.obs_space_uncorr(hx2, y2, var2)
    np.testing.assert_allclose(yb_s.cpu().numpy(), np.concatenate([g["yb"], yb2], axis=1), atol=1e-12)
    np.testing.assert_allclose(d_s.cpu().numpy(), np.concatenate([g["d"], d2]), atol=1e-12)
    # ... and straight into the global ETKF of the fixture
    w = f.estimate_weights_arrays(yb, d).cpu().numpy()
    assert rel_fro(w, g["weights_global_1p1"]) < 1e-9


def test_step_driver_on_a_scattered_2d_network_vs_oracle(mia):
    """The one-call step (serial and with steps in flight) on a 2-D mesh with randomly scattered observations: list
    lengths from 0 (grid points with no observation in reach -> prior weights) to well above the ensemble size
    (primal route), horizontal x vertical radii.  Spot-checked against the oracle's per-grid-point loop."""
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(8)
    k, G, P = 12, 1500, 900
    grid = np.stack([rs.uniform(0, 1, G), rs.uniform(0, 0.2, G)], axis=1)
    obs = np.stack([np.concatenate([rs.uniform(0, 0.45, P - 40), rs.uniform(0.9, 1.0, 40)]), rs.uniform(0, 0.2, P)], axis=1)
    state = rs.normal(size=(2, k, G))
    hx = rs.normal(size=(k, P))
    yb = hx - hx.mean(axis=0)
    d = rs.normal(size=P)
    radii, groups = [0.05, 0.08], [0, 1]
    args = (torch.as_tensor(state, dtype=torch.float32, device=dev), torch.as_tensor(grid, device=dev),
            torch.as_tensor(obs, device=dev), torch.as_tensor(yb, dtype=torch.float32, device=dev),
            torch.as_tensor(d, dtype=torch.float32, device=dev))
    runner = mia.ShardedLetkf(dev, 0, 1, radii=radii, inf_factor=1.05, coord_group=groups)
    outs = [runner.assimilate(*args) for _ in range(2)]               # exact-list call, then the native driver
    pend = [runner.submit(*args) for _ in range(3)]                   # three steps in flight
    outs += [h.result() for h in pend]
    assert runner.native_steps == 4 and runner.last_flags_ok()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    got = outs[0].cpu().numpy()
    counts = []
    for gi in rs.choice(G, 60, replace=False):
        dist = O.grouped_euclid_distance(grid[gi], obs, groups, 2)
        use, _ = O.localize_obs(dist, radii)
        counts.append(int(use.sum()))
        w = O.localized_weights(dist, yb, d, radii, 1.05)
        ref = O.apply_weights(state[:, :, [gi]], w[None])
        assert rel_fro(got[:, :, gi], ref[:, :, 0]) < 1e-5, (gi, counts[-1])
    assert min(counts) == 0 and max(counts) > k                        # both extremes were among the samples


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-5)])
def test_assimilate_filter_and_smoother_mode_on_the_reference_fixture(mia, golden, dtype, tol):
    """``LETKF.assimilate`` end to end on the reference's fixture with ALL THREE times of state and observations
    (interface/base.py:419-512, filter.py:39-165): filter mode cuts everything to the analysis time (P = 40), smoother
    mode stacks 3 x 40 observations; the obs-space normalisation (correlated R), localisation, weights and transform
    all run on the GPU.  The localisation's dist_func is the reference test's own (test_letkf.py:107-110): it indexes
    the observation table BY COLUMN NAME, so obs_info must be a DataFrame (mixin_local.py:44-47)."""
    g = golden("g6_reference_fixture_letkf.npz")
    F = mia.ModelState, mia.ObsSubset

    def operator(sub, pseudo):                       # testing/dummy.py:39-66
        return np.asarray(pseudo.values)[0].transpose(1, 0, 2)

    def dist_func(x, y):
        diff = x - y
        return diff["obs_grid_1"].abs().values,

    state = mia.ModelState(g["state"], g["state_time"], g["grid"])
    obs = mia.ObsSubset(g["obs"], g["cov"], g["obs_time"], g["obs_grid"], operator)
    mute = mia.ObsSubset(g["obs"], g["cov"], g["obs_time"], g["obs_grid"])            # no operator: dropped
    for df in (dist_func, mia.AbsoluteDistance()):
        algo = mia.LETKF(localization=mia.GaspariCohn((10.0,), df), inf_factor=1.1, dtype=dtype)
        ana = algo.assimilate(state, (mute, obs), analysis_time=g["state_time"][0])
        assert tuple(ana.values.shape) == (2, 1, 10, 40) and ana.values.is_cuda
        assert rel_fro(ana.values.cpu().numpy(), g["analysis_1p1"]) < tol
    algo.smoother = True
    ana = algo.assimilate(state, obs)
    assert tuple(ana.values.shape) == (2, 3, 10, 40)
    assert rel_fro(ana.values.cpu().numpy(), g["analysis_smoother_1p1"]) < tol
    # global ETKF through the same entry point; nearest-time fallback warns and lands on time 0
    with pytest.warns(UserWarning, match="is not within state"):
        ana = mia.ETKF(inf_factor=1.1, dtype=dtype).assimilate(state, obs, analysis_time=g["state_time"][0] + 60.0)
    assert rel_fro(ana.values.cpu().numpy(), g["analysis_global_1p1"]) < tol


def _peer_rank_stub(mia, runner, rank, ref, G, n_slots=3):
    """Rank `rank` of a 2-rank world with the direct (peer-mapped) exchange, the OTHER rank played by plain tensors: its
    result buffers and flag area are attached with mia_comm_peer_attach, its blocks (cut from a reference analysis) and its
    flags (free / ready at a sequence number no step reaches) are put in place beforehand, as a peer that is always ahead
    would have.  What the step under test pushes into the stub's buffers and flag area is then inspected."""
    import ctypes as C
    from torch_assimilate_amd import _cabi
    lib = _cabi.lib()
    dev = ref.device
    m, k = ref.shape[0], ref.shape[1]
    peer = 1 - rank
    cb = (_cabi.ALLGATHER_FN(lambda *a: 1), _cabi.ALLREDUCE_MAX_I32_FN(lambda *a: 1))        # never called on this route
    handle = C.c_void_p()
    _cabi.check(lib.mia_comm_create_custom(rank, 2, C.cast(cb[0], C.c_void_p), C.cast(cb[1], C.c_void_p), None,
                                           C.byref(handle)), "mia_comm_create_custom")
    _cabi.check(lib.mia_comm_peer_alloc(handle, m * k * G * 4, n_slots, None), "mia_comm_peer_alloc")
    stub_bufs = [torch.full((m, k, G), -7.0, dtype=torch.float32, device=dev) for _ in range(n_slots)]
    stub_sync = torch.zeros(8 * 16 * 6, dtype=torch.int32, device=dev)
    ptrs = (C.c_void_p * n_slots)(*[b.data_ptr() for b in stub_bufs])
    _cabi.check(lib.mia_comm_peer_attach(handle, peer, ptrs, C.c_void_p(stub_sync.data_ptr())), "mia_comm_peer_attach")
    mine = [runner._wrap_device(lib.mia_comm_peer_buffer(handle, s), (m, k, G), dev) for s in range(n_slots)]
    my_sync = runner._wrap_device(lib.mia_comm_peer_sync_area(handle), (8 * 16 * 6,), dev).view(torch.int32)
    n = (G + 1) // 2
    lo, hi = min(G, peer * n), min(G, (peer + 1) * n)
    for s in range(n_slots):
        mine[s].fill_(-3.0)
        mine[s][:, :, lo:hi] = ref[:, :, lo:hi]                   # the peer's block has "arrived"
        w0 = s * 96
        my_sync[w0 + peer] = 1 << 20                               # ready[slot][peer]
        my_sync[w0 + 16 + peer] = 1 << 20                          # free[slot][peer]
        my_sync[w0 + 32 + 4 * peer:w0 + 32 + 4 * peer + 4] = 0     # the peer's counters
    torch.cuda.synchronize()
    return handle, cb, mine, stub_bufs, stub_sync, (lo, hi)


@pytest.mark.parametrize("G,strong,rank,split", [(2000, False, 0, 0), (1999, False, 0, 0), (1001, True, 0, 0), (2000, False, 1, 0),
                                                 (1999, False, 1, 0), (1001, True, 1, 0), (1999, False, 1, 1), (1001, True, 1, 1)])
def test_direct_peer_exchange_layout_and_protocol_on_one_gpu(mia, G, strong, rank, split):
    """The peer-write exchange of mia_letkf_sharded_step_streams_f32 (Xa = a library-owned, peer-mapped result buffer): this
    rank's block is analysed straight into its own (m, k, G) buffer, pushed into the peer's buffer at the same place, its
    counters and the free / ready flags land in the peer's flag area with the step's sequence number, the redo decision is
    folded from both ranks' counters, and a phase-1 redo (declined points) exchanges again.  Bit for bit the single-rank
    result (f32 products; to rounding with the split-precision products, see the two-rank layout test), serial and with
    three steps in flight."""
    set_option("tile_split", split)
    same = torch.equal if not split else (lambda a, b: float((a - b).norm() / b.norm()) < 1e-6)
    dev = torch.device("cuda:0")
    case = O.synthetic_case(G, 40, 2)
    scale = 12.0 if strong else 1.0
    args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
            torch.as_tensor(case["obs_x"], device=dev),
            torch.as_tensor(case["yb"] * scale, dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"] * scale, dtype=torch.float32, device=dev))
    plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False)
    ref = plain.assimilate(*args)
    runner = mia.ShardedLetkf(dev, rank, 2, radii=[10.0], inf_factor=1.1, max_in_flight=3)
    runner._p_max_hint = plain._p_max_hint
    handle, keep, mine, stub_bufs, stub_sync, (lo, hi) = _peer_rank_stub(mia, runner, rank, ref, G)
    runner._native = dict(comm=handle, custom=True, stream=torch.cuda.Stream(device=dev), slots=[{}, {}, {}],
                          peer=mine, peer_shape=tuple(ref.shape))
    n = (G + 1) // 2
    b0, b1 = min(G, rank * n), min(G, (rank + 1) * n)
    try:
        for it in range(2):
            out = runner.assimilate(*args)                 # serial steps use slot 0
            torch.cuda.synchronize()
            assert same(out, ref) and runner.last_flags_ok()
            assert same(stub_bufs[0][:, :, b0:b1], ref[:, :, b0:b1])          # pushed into the peer's buffer ...
            assert float(stub_bufs[0][:, :, lo:hi].max()) == -7.0                     # ... and nowhere else
            sync = stub_sync.cpu().numpy()
            n_exch = (it + 1) * (2 if strong else 1)                                  # a redo exchanges a second time
            assert sync[rank] == n_exch and sync[16 + rank] == n_exch                 # ready / free carry the sequence number
            assert sync[32 + 4 * rank] == runner.last_p_max or sync[32 + 4 * rank] <= runner.last_p_max
        if strong:
            assert 0 < runner.last_retries <= plain.last_retries          # (this rank's declined points)
        pend = []
        for _ in range(6):                                  # three steps in flight: slots 0, 1, 2 in turn
            pend.append(runner.submit(*args))
            if len(pend) == 3:
                assert same(pend.pop(0).result(), ref)
        while pend:
            assert same(pend.pop(0).result(), ref)
        assert runner.native_steps == 8 and runner.exchange_route.startswith("direct")
        for s in (1, 2):
            assert same(stub_bufs[s][:, :, b0:b1], ref[:, :, b0:b1])
    finally:
        from torch_assimilate_amd import _cabi
        runner._native = None
        _cabi.lib().mia_comm_destroy(handle)


def test_a_late_peer_is_waited_for_again_and_a_dead_one_reported(mia):
    """Direct exchange, emulated peer: its ready flag for slot 0 has NOT arrived when the waiter's (shortened) bound runs out -- error
    bit 2.  The host waits again (mia_comm_peer_rewait: the flags of that exchange, the counters folded once more) instead of
    failing the run; the peer's flag arrives during the first re-wait here and the step completes with the right analysis.  A
    peer that stays silent through every re-wait is an error that says so."""
    import warnings
    from torch_assimilate_amd import _cabi
    dev = torch.device("cuda:0")
    G, rank = 2000, 0
    case = O.synthetic_case(G, 40, 2)
    args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
            torch.as_tensor(case["obs_x"], device=dev), torch.as_tensor(case["yb"], dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"], dtype=torch.float32, device=dev))
    plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False)
    ref = plain.assimilate(*args)
    runner = mia.ShardedLetkf(dev, rank, 2, radii=[10.0], inf_factor=1.1, max_in_flight=3)
    runner._p_max_hint = plain._p_max_hint
    handle, keep, mine, stub_bufs, stub_sync, (lo, hi) = _peer_rank_stub(mia, runner, rank, ref, G)
    lib = _cabi.lib()
    _cabi.check(lib.mia_comm_peer_wait_bound(handle, 10), "mia_comm_peer_wait_bound")      # ~1 ms instead of a minute
    runner._native = dict(comm=handle, custom=True, stream=torch.cuda.Stream(device=dev), slots=[{}, {}, {}],
                          peer=mine, peer_shape=tuple(ref.shape))
    my_sync = runner._wrap_device(lib.mia_comm_peer_sync_area(handle), (8 * 16 * 6,), dev).view(torch.int32)
    peer = 1 - rank
    try:
        out = runner.assimilate(*args)                         # the peer is on time
        assert torch.equal(out, ref)
        my_sync[peer] = 0                                      # ready[slot 0][peer]: not yet
        torch.cuda.synchronize()
        calls = []

        def late_peer(attempt):                                # the peer's flag lands while this rank prepares to wait again
            calls.append(attempt)
            my_sync[peer] = 1 << 20
            torch.cuda.synchronize()
        runner._peer_rewait_hook = late_peer
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = runner.assimilate(*args)
        assert calls == [0] and any("waiting again" in str(x.message) for x in w)
        assert torch.equal(out, ref) and runner.last_flags_ok()
        my_sync[peer] = 0                                      # ... and a peer that never answers
        torch.cuda.synchronize()
        runner._peer_rewait_hook = lambda attempt: calls.append(attempt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(_cabi.MiaError, match="did not deliver within 4 waits"):
                runner.assimilate(*args)
        assert calls == [0, 0, 1, 2]
    finally:
        runner._native = None
        lib.mia_comm_destroy(handle)


def _peer_ipc_worker(rank, port, G, out_path):
    """One of two PROCESSES sharing cuda:0: real hipIpcGetMemHandle / hipIpcOpenMemHandle mapping of the other process's result
    buffers and flag area, handle exchange over torch.distributed (gloo), the exchange self-test, then real steps of both
    ranks running concurrently (each waits on flags the other process's kernels write)."""
    import ctypes as C
    import datetime
    import os
    import torch.distributed as dist

    def stage(name):            # progress marker: what the parent reports if this process has to be stopped
        with open(out_path + ".stage%d" % rank, "w") as fh:
            fh.write(name)
    stage("rendezvous")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=90))
    stage("reference")
    import torch_assimilate_amd as mia
    from torch_assimilate_amd import _cabi
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    case = O.synthetic_case(G, 40, 2)
    args = (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
            torch.as_tensor(case["obs_x"], device=dev), torch.as_tensor(case["yb"], dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"], dtype=torch.float32, device=dev))
    plain = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, native_step=False)
    ref = plain.assimilate(*args)
    runner = mia.ShardedLetkf(dev, rank, 2, radii=[10.0], inf_factor=1.1, max_in_flight=2, copy_results=True)
    runner._p_max_hint = plain._p_max_hint
    cb = (_cabi.ALLGATHER_FN(lambda *a: 1), _cabi.ALLREDUCE_MAX_I32_FN(lambda *a: 1))
    handle = C.c_void_p()
    _cabi.check(_cabi.lib().mia_comm_create_custom(rank, 2, C.cast(cb[0], C.c_void_p), C.cast(cb[1], C.c_void_p), None,
                                                   C.byref(handle)), "mia_comm_create_custom")
    st = dict(comm=handle, custom=True, stream=torch.cuda.Stream(device=dev, priority=-1), slots=[{}, {}])
    import warnings
    stage("peer setup")
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        bufs = runner._peer_setup(st, 1, 40, G)
    stage("steps")
    result = {"rank": rank, "ipc": bufs is not None, "why": [str(w.message) for w in wlist], "ok": False}
    if bufs is not None:
        st["peer"], st["peer_shape"] = bufs, (1, 40, G)
        runner._native = st
        ok = True
        # (to rounding: the split-precision products scale their operands per tile, and rank 1's tiles start elsewhere than
        #  the single-rank run's; a block that did not arrive would be off by O(1))
        same = lambda a, b: float((a - b).norm() / b.norm()) < 1e-6
        pend = []
        for _ in range(6):                                     # two steps in flight per rank, both ranks concurrently
            pend.append(runner.submit(*args))
            if len(pend) == 2:
                ok = ok and bool(same(pend.pop(0).result(), ref))
        while pend:
            ok = ok and bool(same(pend.pop(0).result(), ref))
        result["ok"] = ok and runner.last_flags_ok() and runner.native_steps == 6
    stage("barrier")
    dist.barrier()
    stage("done")
    runner._native = None
    _cabi.lib().mia_comm_destroy(handle)
    import json
    with open(out_path + ".%d" % rank, "w") as fh:
        json.dump(result, fh)
    dist.destroy_process_group()


def test_direct_peer_exchange_between_two_processes_on_one_gpu(tmp_path):
    import json, os, socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "peer")
    import time
    ctx = mp.spawn(_peer_ipc_worker, args=(port, 3000, out), nprocs=2, join=False)
    t_end = time.time() + 240.0                    # bounded: a rank that stops responding fails the test instead of hanging it
    done = False
    while not done and time.time() < t_end:
        done = ctx.join(timeout=5.0)               # (raises if a rank raised; the other rank is terminated then)
    if not done:
        stages = [open(out + ".stage%d" % r).read() if os.path.exists(out + ".stage%d" % r) else "not started" for r in (0, 1)]
        for proc in ctx.processes:
            if proc.is_alive():
                proc.terminate()
        for proc in ctx.processes:
            proc.join(10.0)
        pytest.fail("two-process exchange did not finish in 240 s; ranks were at: %s" % stages)
    res = [json.load(open(out + ".%d" % r)) for r in (0, 1)]
    if not all(r["ipc"] for r in res):
        pytest.skip("device-memory IPC between processes is not available on this box: %s" % (res[0]["why"] or res[1]["why"]))
    assert all(r["ok"] for r in res), res


def test_interface_classes_reach_the_tile_kernels(mia, golden, monkeypatch):
    """LETKF / LKETKF.analyse_arrays (what assimilate() ends in) with a built-in distance take the tile route -- tile lists, then
    letkf_tile2_kernel (plain) or lketkf_tile_kernel (RBF kernel) -- and a user-defined distance callable does not; both agree
    with the reference's analysis (golden g7)."""
    g = golden("g7_synthetic_configs.npz")
    calls = {"plain": 0, "rbf": 0}
    from torch_assimilate_amd.engine import LetkfEngine
    orig_plain, orig_rbf = LetkfEngine.analysis_tiles, LetkfEngine.analysis_tiles_rbf

    def spy_plain(self, *a, **k):
        calls["plain"] += 1
        return orig_plain(self, *a, **k)

    def spy_rbf(self, *a, **k):
        calls["rbf"] += 1
        return orig_rbf(self, *a, **k)
    monkeypatch.setattr(LetkfEngine, "analysis_tiles", spy_plain)
    monkeypatch.setattr(LetkfEngine, "analysis_tiles_rbf", spy_rbf)
    loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())
    kw2 = dict(grid_coords=g["c2_grid_x"], obs_coords=g["c2_obs_x"])
    xa = mia.LETKF(localization=loc, inf_factor=1.1, dtype=torch.float32).analyse_arrays(g["c2_state"], g["c2_yb"], g["c2_d"], **kw2)
    assert calls == {"plain": 1, "rbf": 0} and rel_fro(xa.cpu().numpy(), g["c2_1p1_analysis"]) < 1e-5
    xa3 = mia.LETKF(localization=loc, inf_factor=1.0, dtype=torch.float32).analyse_arrays(g["c2m3_state"], g["c2m3_yb"], g["c2m3_d"],
                                                                      grid_coords=g["c2m3_grid_x"], obs_coords=g["c2m3_obs_x"])
    assert calls["plain"] == 2 and rel_fro(xa3.cpu().numpy(), g["c2m3_1p0_analysis"]) < 1e-5
    kw5 = dict(grid_coords=g["c5_grid_x"], obs_coords=g["c5_obs_x"])
    xk = mia.LKETKF(mia.RBFKernel(0.5), localization=loc, inf_factor=1.1, dtype=torch.float32).analyse_arrays(g["c5_state"], g["c5_yb"], g["c5_d"], **kw5)
    assert calls["rbf"] == 1 and rel_fro(xk.cpu().numpy(), g["c5_1p1_analysis"]) < 1e-5
    # a user callable (the reference's arbitrary dist_func, gaspari_cohn.py:124-125): host-evaluated distances, per-point lists
    user = mia.GaspariCohn(10.0, lambda grid, obs: np.abs(np.asarray(obs, dtype=np.float64).reshape(-1) - float(np.asarray(grid).reshape(-1)[0])))
    before = dict(calls)
    xu = mia.LETKF(localization=user, inf_factor=1.1, dtype=torch.float32).analyse_arrays(g["c2_state"], g["c2_yb"], g["c2_d"], **kw2)
    assert calls == before and rel_fro(xu.cpu().numpy(), g["c2_1p1_analysis"]) < 1e-5


def test_weights_with_a_carried_list_bound_follow_a_changing_network(mia):
    """estimate_weights_arrays carries the list bound of its last call and builds the per-point lists only when needed: the
    same weights as a fresh object -- for the same network, for a denser one (the bound breaks: exact lists first) and back."""
    import bench
    dev = torch.device("cuda:0")
    X, gx, ox, Yb, d = bench.make_case(6000, 40, 2, dev, seed=21)
    loc = lambda: mia.GaspariCohn(6.0, mia.AbsoluteDistance())
    a = mia.LETKF(loc(), inf_factor=1.1, dtype=torch.float32)
    w1 = a.estimate_weights_arrays(Yb, d, grid_coords=gx[:, None], obs_coords=ox[:, None]).clone()
    w1b = a.estimate_weights_arrays(Yb, d, grid_coords=gx[:, None], obs_coords=ox[:, None]).clone()       # carried bound
    assert torch.equal(w1, w1b) and a._w_hint is not None
    # the same number of observations, packed into half the domain: lists twice as long -- the carried bound does not hold
    ox_dense = (ox * 0.5).contiguous()
    fresh = mia.LETKF(loc(), inf_factor=1.1, dtype=torch.float32).estimate_weights_arrays(Yb, d, grid_coords=gx[:, None], obs_coords=ox_dense[:, None])
    w2 = a.estimate_weights_arrays(Yb, d, grid_coords=gx[:, None], obs_coords=ox_dense[:, None])
    assert torch.equal(w2, fresh)
    w3 = a.estimate_weights_arrays(Yb, d, grid_coords=gx[:, None], obs_coords=ox[:, None])
    assert torch.equal(w3, w1)
