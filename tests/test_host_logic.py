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
    assert mia.LinearKernel().gamma is None
    assert str(mia.LinearKernel()) == "LinearKernel" and repr(mia.RBFKernel()) == "RBFKernel"


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
    k = mia.LKETKF(mia.RBFKernel(0.5), localization=loc)
    assert k._gamma == 0.5
    k.kernel = mia.LinearKernel()
    assert k._gamma is None
    assert mia.KETKF(mia.GaussKernel(2.0))._gamma == pytest.approx(0.125)
