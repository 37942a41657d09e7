"""Host-side logic that needs no GPU: shard partitioning, kernel descriptors, metric descriptors,
the interface classes' constructor surface."""
import numpy as np
import pytest

import torch_assimilate_amd as mia
from oracle import letkf_oracle as O


def test_block_partition_covers_grid():
    for G, w in ((100000, 1), (1000000, 8), (10, 3), (7, 8), (0, 2)):
        parts = mia.block_partition(G, w)
        assert len(parts) == w
        assert parts[0][0] == 0 and parts[-1][1] == G
        for (a0, a1), (b0, b1) in zip(parts[:-1], parts[1:]):
            assert a1 == b0 and a0 <= a1
        sizes = {b - a for a, b in parts if b - a > 0}
        assert len(sizes) <= 2


def test_kernel_descriptors():
    assert mia.RBFKernel(0.5).gamma == 0.5
    assert mia.RBFKernel(10.0).lengthscale == pytest.approx((0.5 / 10.0) ** 0.5)
    assert mia.GaussKernel(1.0).gamma == mia.RBFKernel(0.5).gamma        # tests/unit_tests/kernels/test_rbf.py:112-116
    assert mia.GaussKernel(2.0).gamma == pytest.approx(0.125)
    vec = mia.GaussKernel(np.array([1.0, 2.0, 0.5]))                     # per-observation lengthscales: global KETKF only
    assert vec.gamma == 0.5 and np.allclose(vec.feature_scale, [1.0, 0.5, 2.0]) and mia.GaussKernel(2.0).feature_scale is None
    assert mia.KETKF(vec)._kernel_args()["rbf_gamma"] == 0.5
    with pytest.raises(NotImplementedError):
        mia.LKETKF(vec)._kernel_args()
    assert mia.LinearKernel().gamma is None
    assert str(mia.LinearKernel()) == "LinearKernel" and repr(mia.RBFKernel()) == "RBFKernel"


def test_vector_lengthscale_is_refused_inside_compositions():
    """rbf.py:75-78 divides both arguments by a per-feature lengthscale.  Only a GaussKernel used on its own in the global KETKF
    gets that treatment here; inside a composition (or under localisation) the vector must not be dropped silently."""
    import numpy as np
    from torch_assimilate_amd import kernels as K
    vec = K.GaussKernel(np.array([1.0, 2.0, 0.5]))
    assert K.kernel_route(vec, allow_feature_scale=True) == (0.5, None)
    with pytest.raises(NotImplementedError):
        K.kernel_route(vec)
    for comp in (vec * K.ScaleKernel(0.3), K.LinearKernel() + vec, vec ** K.ScaleKernel(0.0)):
        with pytest.raises(NotImplementedError):
            K.kernel_route(comp, allow_feature_scale=True)
    assert K.kernel_route(K.GaussKernel(2.0) * K.ScaleKernel(0.3))[1] is not None


def test_metric_descriptor_matches_oracle_distance():
    rs = np.random.RandomState(3)
    g, o = rs.normal(size=3), rs.normal(size=(50, 3))
    m = mia.EuclideanMetric([0, 0, 1])
    got = np.stack(m(g, o))
    np.testing.assert_allclose(got, O.grouped_euclid_distance(g, o, [0, 0, 1], 2), rtol=1e-15)
    # tolerate the reference's leading time column in grid_info (mixin_local.py:55-58)
    got2 = np.stack(m(np.concatenate([[123.0], g]), o))
    np.testing.assert_array_equal(got, got2)
    a = mia.AbsoluteDistance()
    np.testing.assert_allclose(a(np.array([2.0]), np.arange(5.0))[0], np.abs(np.arange(5.0) - 2.0))
    assert m.groups(3, 2) == [0, 0, 1]
    with pytest.raises(ValueError):
        m.groups(2, 2)


def test_interface_constructor_surface():
    loc = mia.GaspariCohn(10.0, mia.AbsoluteDistance())
    assert str(loc) == "GaspariCohn(l=[10.])" and loc.epsilon == 1e-5
    a = mia.LETKF(localization=loc, inf_factor=1.1, smoother=False, gpu=False, pre_transform=None,
                  post_transform=None, chunksize=10, weight_save_path=None, forward_model=None)
    assert a.inf_factor == 1.1 and a.chunks == {"grid": 10}
    a.inf_factor = 1.3
    assert a.inf_factor == 1.3
    assert repr(mia.ETKF(1.0)) == "ETKF(1.0)"
    # working precision: the reference's default and its setter's error (interface/base.py:68,73,106-118)
    import torch
    assert a.dtype is torch.float64 and mia.ETKF().dtype is torch.float64 and mia.KETKF(mia.LinearKernel()).dtype is torch.float64
    a.dtype = torch.float32
    assert a.dtype is torch.float32
    with pytest.raises(TypeError):
        a.dtype = "float32"
    with pytest.raises(TypeError):
        mia.LETKF(dtype=np.float64)
    k = mia.LKETKF(mia.RBFKernel(0.5), localization=loc)
    assert k._kernel_args() == dict(rbf_gamma=0.5, kernel_program=None)
    k.kernel = mia.LinearKernel()
    assert k._kernel_args() == dict(rbf_gamma=None, kernel_program=None)
    assert mia.KETKF(mia.GaussKernel(2.0))._kernel_args()["rbf_gamma"] == pytest.approx(0.125)
    k.kernel = mia.PolyKernel(2.0, 1.0)
    assert k._kernel_args()["rbf_gamma"] is None and len(k._kernel_args()["kernel_program"]) == 5
    inf = mia.GaspariCohnInf(10.0, mia.AbsoluteDistance())
    assert str(inf) == "GaspariCohnInf(l=[10.])" and repr(inf) == "GaspariCohnInf" and inf._thres == [2, 1.5, 1, 0.5]
    with pytest.raises(ValueError):
        mia.GaspariCohnInf((10.0, 2.0), mia.AbsoluteDistance())


def run_program(prog, dot, sq, l1, same):
    """Pure-Python evaluator of a kernel expression with the device's semantics (csrc/letkf_wave.hip kprog_eval)."""
    from torch_assimilate_amd import kernels as K
    st = []
    for op, val in prog:
        if op == K.KOP_DOT:
            st.append(dot)
        elif op == K.KOP_SQDIST:
            st.append(sq)
        elif op == K.KOP_L1DIST:
            st.append(l1)
        elif op == K.KOP_CONST:
            st.append(np.full_like(dot, val))
        elif op == K.KOP_DIAG:
            st.append(np.where(same, val, 0.0))
        elif op in (K.KOP_ADD, K.KOP_MUL, K.KOP_POW):
            y, x = st.pop(), st.pop()
            st.append(x + y if op == K.KOP_ADD else (x * y if op == K.KOP_MUL else np.power(x, y)))
        elif op == K.KOP_EXP:
            st.append(np.exp(st.pop()))
        elif op == K.KOP_TANH:
            st.append(np.tanh(st.pop()))
        elif op == K.KOP_SIN:
            st.append(np.sin(st.pop()))
        else:
            raise AssertionError(op)
    assert len(st) == 1
    return st[0]


def test_kernel_programs_reproduce_the_reference_kernels(golden):
    """every kernel descriptor's expression, evaluated on the three pair statistics, gives the kernel matrices the
    reference's kernel classes produced (golden g8: K(x, x) and K(x, y) with a single y row, as in the KETKF)."""
    import torch
    from kernel_cases import product_kernels, oracle_kernels
    from torch_assimilate_amd import kernels as K
    g = golden("g8_kernels_gcinf.npz")
    x, y = g["kern_x"], g["kern_y"]
    for name, kern in product_kernels().items():
        gamma, prog = K.kernel_route(kern)
        assert gamma is None and prog
        for other, same, key in ((x, np.eye(len(x), dtype=bool), "kxx"), (y, np.zeros((len(x), len(y)), bool), "kxy")):
            diff = x[:, None, :] - other[None, :, :]
            got = run_program(prog, x @ other.T, (diff ** 2).sum(-1), np.abs(diff).sum(-1), same)
            np.testing.assert_allclose(got, g[f"{key}_{name}"], rtol=1e-12, atol=1e-13, err_msg=name)
        ora = oracle_kernels()[name](torch.tensor(x), torch.tensor(x)).numpy()
        np.testing.assert_allclose(ora, g[f"kxx_{name}"], rtol=1e-13, atol=1e-13)
    assert K.kernel_route(None) == (None, None) and K.kernel_route(K.LinearKernel()) == (None, None)
    assert K.kernel_route(K.RBFKernel(10.0)) == (10.0, None)
    with pytest.raises(NotImplementedError):
        K.kernel_route(torch.nn.Linear(2, 2))          # ModuleKernel-style user code is not mirrored
    with pytest.raises(NotImplementedError):
        K.kernel_route(K.GaussKernel(torch.ones(3)))    # per-observation lengthscales: global KETKF only ...
    assert K.kernel_route(K.GaussKernel(torch.ones(3)), allow_feature_scale=True) == (0.5, None)
    with pytest.raises(TypeError):
        K.PolyKernel() + 3.0
    deep = K.ScaleKernel(1.0)
    for _ in range(12):
        deep = deep + K.PolyKernel()
    with pytest.raises(ValueError):
        K.kernel_route(deep)                            # more operations than the device evaluates
    assert str(K.PolyKernel(2.0, 1.0) + K.RBFKernel(0.5)) == "PolynomialKernel(2.0, 1.0)+RBFKernel(γ=0.5)"


def test_weight_files_round_trip_and_layout(tmp_path):
    """store_weights / load_weights (base.py:280-325, utilities/xarray.py:36-173) on host tensors: netCDF-3 layout
    of xarray's scipy engine, multi-level grid index flattened with `multidim_levels`, values exact in float64."""
    import torch
    from scipy.io import netcdf_file
    from torch_assimilate_amd import weights_io as io
    rs = np.random.RandomState(0)
    W = torch.tensor(rs.normal(size=(50, 6, 6)))
    path = str(tmp_path / "weights.nc")
    io.store_weights(path, W, grid_levels={"lat": rs.normal(size=50), "lon": np.arange(50)}, ensemble=np.arange(6) * 2)
    f = netcdf_file(path, mmap=False)
    assert f.version_byte == 2
    var = f.variables[io.DATA_VARIABLE]
    assert var.dimensions == ("grid", "ensemble", "ensemble_new") and var.data.dtype == np.dtype(">f8")
    assert f.variables["grid"].multidim_levels == b"lat;lon"                  # encode_multidim, xarray.py:97-102
    np.testing.assert_array_equal(f.variables["ensemble_new"][:], np.arange(6) * 2)
    f.close()
    out, coords = io.load_weights(path)
    np.testing.assert_array_equal(out.numpy(), W.numpy())
    assert coords["multidim_levels"] == ["lat", "lon"] and coords["lon"].tolist() == list(range(50))
    io.store_weights(path, W[0].float())                                      # global filter: (ensemble, ensemble_new)
    out2, coords2 = io.load_weights(path, dtype=torch.float32)
    assert out2.dtype == torch.float32 and sorted(coords2) == ["ensemble", "ensemble_new"]
    np.testing.assert_array_equal(out2.numpy(), W[0].float().numpy())
    with pytest.raises(ValueError):
        io.store_weights(path, W[:, :3])
    with pytest.raises(ValueError):
        io.store_weights(path, W, grid_index=np.arange(49))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE (how the driver calls it) starts N worker processes itself,
    relays rank 0's JSON line and fails when a worker fails.  MIA_BENCH_DRYRUN swaps the GPU workload for the gloo /
    oracle stand-in: launcher, rendezvous, max-over-ranks timing and the line's fields are what is under test."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MIA_BENCH_DRYRUN"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["dryrun"] and line["shape_ok"]
    assert line["value"] > 0 and line["ms_per_step"] > 0 and "2 ranks joined" in line["config"]["parallelism"]
    # a worker that dies (rank 1, before the rendezvous rank 0 then waits in) fails the launcher instead of hanging it
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=dict(env, MIA_BENCH_DRYRUN_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "rank 1 exited with status 3" in bad.stderr


def test_split_half_precision_products_keep_f32_accuracy():
    """The arithmetic of the tile kernel's default route, restated in numpy (no GPU): an f32 operand carried as two
    halves hi = f16(x), lo = f16(x - hi) after a power-of-two scaling into the middle of the f16 range, and a product
    formed as Ah Bh + Ah Bl + Al Bh.  Over eight decades of operand magnitude the result is as close to the float64
    product as a float32 product is (tools/split_emul.py runs the whole recurrence this way)."""
    rs = np.random.RandomState(0)
    f16 = lambda x: x.astype(np.float16).astype(np.float32)

    def split(x):
        m = np.abs(x).max()
        s = np.float32(2.0 ** (9 - np.floor(np.log2(m)))) if m > 0 else np.float32(1.0)
        xs = (x * s).astype(np.float32)
        hi = f16(xs)
        return hi, f16(xs - hi), s

    worst = 0.0
    for scale in (1e-4, 1e-2, 1.0, 30.0, 1e4):
        A = (rs.normal(size=(32, 40)) * scale).astype(np.float32)
        B = (rs.normal(size=(40, 16)) / scale * 3.0).astype(np.float32)
        Ah, Al, sa = split(A)
        Bh, Bl, sb = split(B)
        # products of halves are exact in float32; the accumulation is float32 (numpy float32 matmul)
        got = ((Ah @ Bh) + (Ah @ Bl) + (Al @ Bh)).astype(np.float64) / (float(sa) * float(sb))
        ref = A.astype(np.float64) @ B.astype(np.float64)
        f32 = (A @ B).astype(np.float64)
        e_split = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        e_f32 = np.linalg.norm(f32 - ref) / np.linalg.norm(ref)
        assert e_split < 3e-7 and e_split < 4 * e_f32 + 1e-7, (scale, e_split, e_f32)
        # the representation itself: hi + lo carries 22-23 significant bits
        assert np.abs((Ah + Al).astype(np.float64) / float(sa) - A).max() <= 2.0 ** -21 * np.abs(A).max()
        worst = max(worst, e_split)
    assert worst > 0.0
