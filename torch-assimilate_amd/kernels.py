"""Kernel descriptors accepted by KETKF / LKETKF: mirror of pytassim.kernels (pytassim/kernels/*.py).

The reference's kernels are torch modules that evaluate K(x, y) themselves.  Here they only carry their
parameters: K(Yb, Yb) and K(Yb, d) are evaluated per matrix element INSIDE the fused analysis kernel
(csrc/letkf_wave.hip, ``kprog_eval``), so a kernel compiles itself to a short expression in reverse Polish
form over the three pair statistics every reference kernel is a function of -- x.y, |x - y|_2^2 and |x - y|_1
(include/mia_letkf.h, MIA_KOP_*).  ``+``, ``*`` and ``**`` compose kernels like the reference's
AdditiveKernel / MultiplicativeKernel / PowerKernel (base_kernels.py:61-161).

Fast routes: a plain LinearKernel is the ETKF (dual-space kernels), a plain RBFKernel / GaussKernel has its
own specialised kernels (incl. the eigensolver-free matfun route); everything else runs the expression route.

Not mirrored: ``ModuleKernel`` (module_kernel.py) applies an arbitrary torch.nn.Module to every localised
block -- user code that cannot run inside a HIP kernel.  Per-feature (vector) lengthscales of the Gauss kernel index
the observations: accepted by the global KETKF (inputs pre-divided, see GaussKernel), refused under localisation.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

__all__ = ["BaseKernel", "AdditiveKernel", "MultiplicativeKernel", "PowerKernel", "LinearKernel", "GaussKernel",
           "RBFKernel", "PolyKernel", "TanhKernel", "PeriodicKernel", "RationalKernel",
           "OrnsteinUhlenbeckKernel", "ScaleKernel", "DiagKernel", "kernel_route",
           "KOP_DOT", "KOP_SQDIST", "KOP_L1DIST", "KOP_CONST", "KOP_DIAG", "KOP_ADD", "KOP_MUL", "KOP_POW",
           "KOP_EXP", "KOP_TANH", "KOP_SIN"]

# opcodes of include/mia_letkf.h
KOP_DOT, KOP_SQDIST, KOP_L1DIST, KOP_CONST, KOP_DIAG = 1, 2, 3, 4, 5
KOP_ADD, KOP_MUL, KOP_POW, KOP_EXP, KOP_TANH, KOP_SIN = 6, 7, 8, 9, 10, 11
MAX_OPS, MAX_DEPTH = 24, 6

Program = List[Tuple[int, float]]


def _f32(v: float) -> float:
    """ScaleKernel / DiagKernel build their constant as ``torch.ones(...) * scaling`` with the ones in torch's
    default dtype float32 (scale.py:71-72, diag.py:69-70): the reference's value of c is float32(c)."""
    import struct
    return struct.unpack("f", struct.pack("f", float(v)))[0]


def _scalar(v, what: str) -> float:
    try:
        import torch
        if isinstance(v, torch.Tensor):
            if v.numel() != 1:
                raise NotImplementedError("%s must be a scalar on the gfx950 path (per-feature values index the "
                                          "local observations and have no meaning under localisation)" % what)
            return float(v.item())
    except ImportError:      # pragma: no cover
        pass
    return float(v)


class BaseKernel:
    """base_kernels.py:37-60: kernels compose with ``+``, ``*`` and ``**``."""

    def __add__(self, other: "BaseKernel") -> "BaseKernel":
        return AdditiveKernel(self, other)

    def __mul__(self, other: "BaseKernel") -> "BaseKernel":
        return MultiplicativeKernel(self, other)

    def __pow__(self, other: "BaseKernel") -> "BaseKernel":
        return PowerKernel(self, other)

    def program(self) -> Program:
        """The kernel as [(opcode, value), ...] in reverse Polish form."""
        raise NotImplementedError


class CompKernel(BaseKernel):
    _op = None
    _sym = "?"

    def __init__(self, kernel_1: BaseKernel, kernel_2: BaseKernel):
        for kern in (kernel_1, kernel_2):
            if not isinstance(kern, BaseKernel):
                raise TypeError("kernel compositions take kernels of torch_assimilate_amd.kernels, got %r" % (kern,))
        self.kernel_1 = kernel_1
        self.kernel_2 = kernel_2

    def program(self) -> Program:
        return self.kernel_1.program() + self.kernel_2.program() + [(self._op, 0.0)]

    def __str__(self) -> str:
        return "{0:s}{1:s}{2:s}".format(str(self.kernel_1), self._sym, str(self.kernel_2))

    def __repr__(self) -> str:
        return "{0:s}{1:s}{2:s}".format(repr(self.kernel_1), self._sym, repr(self.kernel_2))


class AdditiveKernel(CompKernel):
    """K1 + K2 (base_kernels.py:98-117)."""
    _op, _sym = KOP_ADD, "+"


class MultiplicativeKernel(CompKernel):
    """K1 * K2 (base_kernels.py:120-139)."""
    _op, _sym = KOP_MUL, "*"


class PowerKernel(CompKernel):
    """K1 ** K2 (base_kernels.py:142-161)."""
    _op, _sym = KOP_POW, "^"


class LinearKernel(BaseKernel):
    """K(x, y) = x y^T: the KETKF with this kernel is the ETKF (linear.py:41-67,
    tests/unit_tests/interface/test_lketkf.py:109-117) and is routed to the ETKF kernels."""
    gamma = None

    def program(self) -> Program:
        return [(KOP_DOT, 0.0)]

    def __str__(self):
        return "LinearKernel"

    def __repr__(self):
        return "Linear"


class GaussKernel(BaseKernel):
    """K(x, y) = exp(-sum_j ((x_j - y_j) / l_j)^2 / 2) (rbf.py:41-81).  A scalar lengthscale works everywhere.  A
    per-feature (vector) lengthscale -- one value per OBSERVATION, rbf.py:75-78 divides both arguments by it -- is accepted
    by the global KETKF, where the observation axis is fixed: the engine divides Yb's columns and d by it and runs the
    unit-lengthscale kernel (``feature_scale``).  Under localisation every grid point sees a different subset of the
    observations, so a per-observation vector has no meaning there: LKETKF raises NotImplementedError."""

    def __init__(self, lengthscale=1.0):
        import numpy as np
        ls = lengthscale.detach().cpu().numpy() if hasattr(lengthscale, "detach") else np.asarray(lengthscale, dtype=np.float64)
        if ls.size == 1:
            self.lengthscale = float(ls.reshape(-1)[0])
            self.feature_scale = None
        else:
            self.lengthscale = 1.0                       # what the kernels see after the inputs were divided by the vector
            self.feature_scale = 1.0 / ls.reshape(-1).astype(np.float64)
            self.lengthscale_vector = ls.reshape(-1).astype(np.float64)

    @property
    def gamma(self) -> float:
        return 0.5 / self.lengthscale ** 2

    def program(self, allow_vector: bool = False) -> Program:
        if self.feature_scale is not None and not allow_vector:
            # (only the top-level kernel of the global KETKF gets its inputs divided by the vector, kernel_route / KETKFModule;
            #  inside a composition the expression route would silently run the unit-lengthscale kernel instead of rbf.py:75-78)
            raise NotImplementedError("a per-observation (vector) lengthscale is supported for a GaussKernel / RBFKernel used on its "
                                      "own in the global KETKF, not inside a kernel composition")
        return [(KOP_SQDIST, 0.0), (KOP_CONST, -self.gamma), (KOP_MUL, 0.0), (KOP_EXP, 0.0)]

    def __str__(self):
        return "GaussKernel(l={0})".format(self.lengthscale if self.feature_scale is None else self.lengthscale_vector)

    def __repr__(self):
        return "GaussKernel"


class RBFKernel(GaussKernel):
    """K(x, y) = exp(-gamma |x - y|^2) == GaussKernel(l = (0.5 / gamma) ** 0.5) (rbf.py:84-111)."""

    def __init__(self, gamma: float = 0.5):
        self._gamma = _scalar(gamma, "gamma")
        super().__init__((0.5 / self._gamma) ** 0.5)

    @property
    def gamma(self) -> float:
        return self._gamma

    def __str__(self):
        return "RBFKernel(γ={0})".format(self._gamma)

    def __repr__(self):
        return "RBFKernel"


class PolyKernel(BaseKernel):
    """K(x, y) = (x y^T + c)^p (polynomial.py:41-82)."""

    def __init__(self, degree: float = 2.0, const: float = 1.0):
        self.degree = _scalar(degree, "degree")
        self.const = _scalar(const, "const")

    def program(self) -> Program:
        return [(KOP_DOT, 0.0), (KOP_CONST, self.const), (KOP_ADD, 0.0), (KOP_CONST, self.degree), (KOP_POW, 0.0)]

    def __str__(self) -> str:
        return "PolynomialKernel({0}, {1})".format(self.degree, self.const)

    def __repr__(self) -> str:
        return "Polynomial({0}, {1})".format(self.degree, self.const)


class TanhKernel(BaseKernel):
    """K(x, y) = tanh(a x y^T + c) (tanh.py:41-87)."""

    def __init__(self, coeff: float = 1.0, const: float = 0.0):
        self.coeff = _scalar(coeff, "coeff")
        self.const = _scalar(const, "const")

    def program(self) -> Program:
        return [(KOP_DOT, 0.0), (KOP_CONST, self.coeff), (KOP_MUL, 0.0), (KOP_CONST, self.const), (KOP_ADD, 0.0),
                (KOP_TANH, 0.0)]

    def __str__(self) -> str:
        return "TanhKernel({0}, {1})".format(self.coeff, self.const)

    def __repr__(self) -> str:
        return "Tanh({0}, {1})".format(self.coeff, self.const)


class PeriodicKernel(BaseKernel):
    """K(x, y) = exp(-2 sin^2(pi |x - y|_1 / p) / l^2) (periodic.py:42-85)."""

    def __init__(self, period: float = 1.0, lengthscale: float = 1.0):
        self.period = _scalar(period, "period")
        self.lengthscale = _scalar(lengthscale, "lengthscale")

    def program(self) -> Program:
        return [(KOP_L1DIST, 0.0), (KOP_CONST, math.pi / self.period), (KOP_MUL, 0.0), (KOP_SIN, 0.0),
                (KOP_CONST, 2.0), (KOP_POW, 0.0), (KOP_CONST, -2.0 / self.lengthscale ** 2), (KOP_MUL, 0.0),
                (KOP_EXP, 0.0)]

    def __str__(self) -> str:
        return "PeriodicKernel({0}, {1})".format(self.period, self.lengthscale)

    def __repr__(self) -> str:
        return "Periodic({0}, {1})".format(self.period, self.lengthscale)


class RationalKernel(BaseKernel):
    """K(x, y) = (1 + |x - y|^2 / (2 a l^2))^-a (rational.py:41-88)."""

    def __init__(self, lengthscale: float = 1.0, weighting: float = 1.0):
        self.lengthscale = _scalar(lengthscale, "lengthscale")
        self.weighting = _scalar(weighting, "weighting")

    def program(self) -> Program:
        return [(KOP_SQDIST, 0.0), (KOP_CONST, 1.0 / (2.0 * self.weighting * self.lengthscale ** 2)), (KOP_MUL, 0.0),
                (KOP_CONST, 1.0), (KOP_ADD, 0.0), (KOP_CONST, -self.weighting), (KOP_POW, 0.0)]

    def __str__(self) -> str:
        return "RationalKernel({0}, {1})".format(self.lengthscale, self.weighting)

    def __repr__(self) -> str:
        return "Rational({0}, {1})".format(self.lengthscale, self.weighting)


class OrnsteinUhlenbeckKernel(BaseKernel):
    """K(x, y) = exp(-|x - y|_1 / l) (orn_uhl.py:41-76)."""

    def __init__(self, lengthscale: float = 1.0):
        self.lengthscale = _scalar(lengthscale, "lengthscale")

    def program(self) -> Program:
        return [(KOP_L1DIST, 0.0), (KOP_CONST, -1.0 / self.lengthscale), (KOP_MUL, 0.0), (KOP_EXP, 0.0)]

    def __str__(self) -> str:
        return "OrnsteinUhlenbeckKernel({0})".format(self.lengthscale)

    def __repr__(self) -> str:
        return "OrnUhlKernel({0})".format(self.lengthscale)


class ScaleKernel(BaseKernel):
    """K(x, y) = c (scale.py:41-74)."""

    def __init__(self, scaling: float = 0.0):
        self.scaling = _scalar(scaling, "scaling")

    def program(self) -> Program:
        return [(KOP_CONST, _f32(self.scaling))]

    def __str__(self) -> str:
        return "ScaleKernel({0})".format(self.scaling)

    def __repr__(self) -> str:
        return repr(self.scaling)


class DiagKernel(BaseKernel):
    """c * I between a sample set and itself, zeros between sets of different size (diag.py:41-73): inside the
    KETKF c on the diagonal of K(Yb, Yb) and 0 everywhere in K(Yb, d)."""

    def __init__(self, scaling: float = 0.0):
        self.scaling = _scalar(scaling, "scaling")

    def program(self) -> Program:
        return [(KOP_DIAG, _f32(self.scaling))]

    def __str__(self) -> str:
        return "DiagKernel({0})".format(self.scaling)

    def __repr__(self) -> str:
        return "Diag({0})".format(self.scaling)


def check_program(prog: Program) -> None:
    """Host-side twin of kernel_program_check (csrc/mia_common.h), for an early, readable error."""
    if not 1 <= len(prog) <= MAX_OPS:
        raise ValueError("kernel expression has %d operations; the device evaluates at most %d" % (len(prog), MAX_OPS))
    depth = 0
    for op, _ in prog:
        if op <= KOP_DIAG:
            depth += 1
            if depth > MAX_DEPTH:
                raise ValueError("kernel expression nests deeper than the %d-entry operand stack" % MAX_DEPTH)
        elif op <= KOP_POW:
            depth -= 1
    if depth != 1:
        raise ValueError("malformed kernel expression")


def kernel_route(kernel, allow_feature_scale: bool = False) -> Tuple[Optional[float], Optional[Program]]:
    """(rbf_gamma, program) for a kernel object: (None, None) = ETKF route, (gamma, None) = specialised RBF
    kernels, (None, program) = kernel-expression route."""
    if kernel is None or type(kernel) is LinearKernel:
        return None, None
    if type(kernel) in (GaussKernel, RBFKernel):
        if getattr(kernel, "feature_scale", None) is not None and not allow_feature_scale:
            raise NotImplementedError("a per-observation (vector) lengthscale has no meaning under localisation, where every "
                                      "grid point sees another subset of the observations; it is accepted by the global KETKF")
        return float(kernel.gamma), None
    if not isinstance(kernel, BaseKernel):
        raise NotImplementedError(
            "kernel %r is not a torch_assimilate_amd.kernels kernel (arbitrary torch modules cannot run inside "
            "the fused gfx950 analysis kernel)" % (kernel,))
    prog = kernel.program()
    check_program(prog)
    return None, prog
