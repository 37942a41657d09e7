"""Kernel descriptors accepted by KETKF / LKETKF (mirror of pytassim.kernels for the kernels the
fused gfx950 path implements).

The reference's kernels are torch modules that evaluate K(x, y) themselves
(pytassim/kernels/rbf.py:75-81, linear.py:66-67).  Here they only carry the parameters: the Gram
matrix is evaluated inside the fused analysis kernel (csrc/letkf_sys.hip, RBF route).  The other
eight reference kernels and kernel compositions are out of scope.
"""
from __future__ import annotations

__all__ = ["RBFKernel", "GaussKernel", "LinearKernel"]


class GaussKernel:
    """K(x, y) = exp(-|x - y|^2 / (2 l^2)); scalar lengthscale only (rbf.py:41-81)."""

    def __init__(self, lengthscale: float = 1.0):
        self.lengthscale = float(lengthscale)

    @property
    def gamma(self) -> float:
        return 0.5 / self.lengthscale ** 2

    def __str__(self):
        return "GaussKernel(l={0})".format(self.lengthscale)

    def __repr__(self):
        return "GaussKernel"


class RBFKernel(GaussKernel):
    """K(x, y) = exp(-gamma |x - y|^2) == GaussKernel(l = (0.5 / gamma) ** 0.5) (rbf.py:84-111)."""

    def __init__(self, gamma: float = 0.5):
        self._gamma = float(gamma)
        super().__init__((0.5 / self._gamma) ** 0.5)

    @property
    def gamma(self) -> float:
        return self._gamma

    def __str__(self):
        return "RBFKernel(γ={0})".format(self._gamma)

    def __repr__(self):
        return "RBFKernel"


class LinearKernel:
    """K(x, y) = x y^T: the KETKF with this kernel is the ETKF (linear.py:41-67,
    tests/unit_tests/interface/test_lketkf.py:109-117) and is routed to the ETKF kernels."""
    gamma = None

    def __str__(self):
        return "LinearKernel"

    def __repr__(self):
        return "Linear"
