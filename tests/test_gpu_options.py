"""Every route option of mia_set_option (include/mia_letkf.h) at its NON-DEFAULT value against the float64 oracle (the
option table and the test matrix one to one).  The options not exercised elsewhere are exercised here; the table at the end of this
file says where each of the fifteen is tested."""
import numpy as np
import pytest
import torch

from conftest import rel_fro, set_option
from oracle import letkf_oracle as O

pytestmark = pytest.mark.gpu
TOL32 = 1e-5

# option -> (non-default value, test that sets it)
OPTION_TESTS = {
    "cheb_dmax": (8, "tests/test_gpu_parity.py::test_matfun_declines_wide_spectra_and_eigensolver_redoes_them, tests/test_gpu_tile2.py"),
    "cheb_table": (0, "tests/test_gpu_parity.py (coefficients computed in the kernel)"),
    "cheb_rowbatch": (0, "tests/test_gpu_parity.py (row-by-row path for many state rows)"),
    "cheb_big": (0, "tests/test_gpu_parity.py (k > 64 with more than 64 local observations: eigensolver)"),
    "tile": (0, "tests/test_gpu_parity.py, tests/test_gpu_tile.py (one grid point per wavefront)"),
    "tile_split": (0, "tests/test_gpu_parity.py, tests/test_gpu_tile.py, tests/test_gpu_interface.py (f32 matrix instructions)"),
    "localize_quad": (0, "tests/test_gpu_options.py::test_localize_one_lane_per_point"),
    "step_hostwait": (0, "tests/test_gpu_options.py::test_step_driver_options[step_hostwait]"),
    "step_lazy_sort": (0, "tests/test_gpu_options.py::test_step_driver_options[step_lazy_sort]"),
    "segment_signal": (0, "tests/test_gpu_interface.py (one launch + event per piece)"),
    "tile_lists": (0, "tests/test_gpu_options.py::test_step_driver_options[tile_lists]"),
    "bucket_index": (0, "tests/test_gpu_step_tiles.py::test_bucket_index_equals_scan_index_and_engine_calls"),
    "tile_pair": (0, "tests/test_gpu_tile2.py (one wavefront per tile for unions of more than 32 slots)"),
    "tile_fused": (0, "tests/test_gpu_options.py::test_step_driver_options[tile_fused]"),
    "step_coalesce": (1, "tests/test_gpu_options.py::test_coalesced_launches_equal_one_launch_per_step"),
}


@pytest.fixture(scope="module")
def mia():
    import torch_assimilate_amd as m
    m.build()
    return m


def test_the_option_table_is_complete(mia):
    """Every name mia_set_option accepts is in the matrix above, and nothing else."""
    from torch_assimilate_amd import _cabi
    import ctypes as C
    lib = _cabi.lib()
    v = C.c_int(0)
    for name in OPTION_TESTS:
        assert lib.mia_get_option(name.encode(), C.byref(v)) == 0, name
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-assimilate_amd", "csrc", "api.cc")).read()
    names = re.search(r"kOptNames\[MIA_OPT_COUNT_\] = \{(.*?)\};", hdr, re.S).group(1)
    assert sorted(re.findall(r'"(\w+)"', names)) == sorted(OPTION_TESTS)


def _args(case, dev):
    return (torch.as_tensor(case["state"], dtype=torch.float32, device=dev), torch.as_tensor(case["grid_x"], device=dev),
            torch.as_tensor(case["obs_x"], device=dev), torch.as_tensor(case["yb"], dtype=torch.float32, device=dev),
            torch.as_tensor(case["d"], dtype=torch.float32, device=dev))


@pytest.mark.parametrize("name", ["step_hostwait", "step_lazy_sort", "tile_lists", "tile_fused"])
def test_step_driver_options(mia, name):
    """The step driver with the option at its non-default value -- serial steps and steps in flight -- against the oracle, and equal
    (to rounding; bit for bit where only the scheduling changes) to the default route."""
    dev = torch.device("cuda:0")
    case = O.synthetic_case(2500, 40, 2, seed=11)
    a = _args(case, dev)
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
    base = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1)
    for _ in range(3):
        ref = base.assimilate(*a).clone()
    base.close()
    if name == "step_lazy_sort":
        set_option("tile_lists", 0)              # (the lazily sorted index belongs to the per-point list route)
    set_option(name, OPTION_TESTS[name][0])
    r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=3)
    for _ in range(3):
        out = r.assimilate(*a).clone()
    pend = [r.submit(*a) for _ in range(5)]
    outs = [h.result().clone() for h in pend]
    assert r.native_steps >= 6 and r.last_flags_ok()
    kern = r.dominant_kernel_name
    r.close()
    assert rel_fro(out.cpu().numpy(), oracle) < TOL32
    for o in outs:
        assert torch.equal(o, out)
    if name in ("step_hostwait", "tile_fused"):
        assert torch.equal(out, ref), kern              # scheduling / where the lists live: the same bits
    else:
        assert float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)) < 2e-6
    if name == "tile_fused":
        assert kern.startswith("letkf_tile2_kernel<"), kern
    if name in ("tile_lists", "step_lazy_sort"):
        assert not kern.startswith("letkf_tile2"), kern


def test_localize_one_lane_per_point(mia):
    """Option localize_quad = 0 (one lane per grid point instead of four for short lists): the same lists bit for bit, and the
    analysis from them against the oracle."""
    dev = torch.device("cuda:0")
    eng = mia.LetkfEngine(dev)
    case = O.synthetic_case(1500, 40, 2, seed=12)
    nb4 = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    set_option("localize_quad", 0)
    nb1 = eng.localize(case["grid_x"], case["obs_x"], [10.0])
    assert nb1.p_max == nb4.p_max and torch.equal(nb1.cnt, nb4.cnt)
    cnt = nb1.cnt.cpu().numpy()
    i1, i4, w1, w4 = nb1.idx.cpu().numpy(), nb4.idx.cpu().numpy(), nb1.w.cpu().numpy(), nb4.w.cpu().numpy()
    for g in range(0, 1500, 7):
        assert sorted(i1[g, :cnt[g]]) == sorted(i4[g, :cnt[g]])
        o1, o4 = np.argsort(i1[g, :cnt[g]]), np.argsort(i4[g, :cnt[g]])
        np.testing.assert_array_equal(w1[g, :cnt[g]][o1], w4[g, :cnt[g]][o4])
    X = torch.as_tensor(case["state"], dtype=torch.float32, device=dev)
    xa = eng.analysis(X, torch.as_tensor(case["yb"], dtype=torch.float32), torch.as_tensor(case["d"], dtype=torch.float32), nb1, 1.1)
    oracle = O.letkf_analysis(case["state"], case["grid_x"], case["obs_x"], case["yb"], case["d"], 10.0, 1.1)[0]
    assert rel_fro(xa.cpu().numpy(), oracle) < TOL32


def test_coalesced_launches_equal_one_launch_per_step(mia):
    """Launch coalescing (option step_coalesce, default off): steps in flight whose preparation has finished share ONE launch of the
    fused kernel.  Thirty-two steps with four different inputs, some of them timed (a timed step opens a launch): some launches hold
    more than one step, every step returns its OWN result, bit for bit what one launch per step (step_coalesce = 0) returns, and the
    result is the oracle's."""
    from torch_assimilate_amd import _cabi
    dev = torch.device("cuda:0")
    G, k, depth = 100000, 40, 8
    cases = [O.synthetic_case(G, k, 2, seed=40 + i) for i in range(4)]
    args = [_args(c, dev) for c in cases]

    def run(timed_every):
        r = mia.ShardedLetkf(dev, 0, 1, radii=[10.0], inf_factor=1.1, max_in_flight=depth)
        for i in range(depth + 2):
            r.assimilate(*args[i % 4])
        l0, s0 = _cabi.step_coalesce_stats()
        pend, outs, batch = [], [], []
        for i in range(32):
            if timed_every and i % timed_every == 0:
                r.time_next_step()
            pend.append(r.submit(*args[i % 4]))
            if len(pend) == depth:
                h = pend.pop(0)
                outs.append(h.result())          # (every step has a result tensor of its own: no copy, the loop stays GPU-bound)
                batch.append(h.batch_n)
        for h in pend:
            outs.append(h.result())
            batch.append(h.batch_n)
        l1, s1 = _cabi.step_coalesce_stats()
        r._note_kernel()
        kern, tb = r.dominant_kernel_name, [r.kernel_batch.get(t, 1) for t in r.kernel_timings]
        assert r.last_flags_ok()
        r.close()
        return outs, batch, (l1 - l0, s1 - s0), kern, tb

    set_option("step_coalesce", OPTION_TESTS["step_coalesce"][0])             # (one launch running at a time: the others wait, and merge)
    for attempt in range(3):          # (whether two steps are ready together is timing: a second and third try before calling it a failure)
        on, b_on, (launches, steps), kern, tb = run(3)
        if max(b_on) >= 2:
            break
    assert steps == 32 and launches <= steps
    assert max(b_on) >= 2 and launches < steps, (b_on, launches)          # (some launch held more than one step ...)
    assert max(b_on) <= 4 and len(tb) == 11 and all(1 <= t <= 4 for t in tb)
    assert kern.startswith("letkf_tile2fb_kernel<"), kern
    set_option("step_coalesce", 0)
    off, b_off, (l_off, s_off), kern_off, _ = run(3)
    assert b_off == [1] * 32 and l_off == 0 and s_off == 0
    assert kern_off.startswith("letkf_tile2f_kernel<"), kern_off
    for i in range(32):
        assert torch.equal(on[i], off[i]), i                               # (... and every step got its own result)
    for i in range(4):
        c = cases[i]
        if i == 0:
            idx = np.arange(0, G, 97)
            oracle = O.letkf_analysis(c["state"][..., idx], c["grid_x"][idx], c["obs_x"], c["yb"], c["d"], 10.0, 1.1)[0]
            assert rel_fro(on[i].cpu().numpy()[..., idx], oracle) < TOL32
        assert not torch.equal(on[i], on[(i + 1) % 4])
