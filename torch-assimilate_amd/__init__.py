"""MI355X-native LETKF local-analysis engine (drop-in for pytassim's ensemble-transform path).

Import name: ``torch_assimilate_amd`` (the directory name carries a hyphen; the repo-root
module ``torch_assimilate_amd.py`` maps the import name onto this directory).
"""
from ._build import build, LIB_PATH            # noqa: F401
from ._cabi import MiaError, EXPORTED_SYMBOLS  # noqa: F401

__version__ = "0.1.0"

_LAZY = {
    "LetkfEngine": "engine", "NeighbourLists": "engine",
    "GaspariCohn": "localization", "EuclideanMetric": "localization", "AbsoluteDistance": "localization",
    "ETKFModule": "core", "KETKFModule": "core",
    "LETKF": "interface", "ETKF": "interface", "LKETKF": "interface", "KETKF": "interface",
    "GaspariCohnInf": "localization",
    "RBFKernel": "kernels", "GaussKernel": "kernels", "LinearKernel": "kernels", "PolyKernel": "kernels",
    "TanhKernel": "kernels", "PeriodicKernel": "kernels", "RationalKernel": "kernels",
    "OrnsteinUhlenbeckKernel": "kernels", "ScaleKernel": "kernels", "DiagKernel": "kernels",
    "AdditiveKernel": "kernels", "MultiplicativeKernel": "kernels", "PowerKernel": "kernels",
    "IEnKSTransformModule": "ienks", "IEnKSBundleModule": "ienks", "IEnKSTransform": "ienks", "IEnKSBundle": "ienks",
    "LocalizedIEnKSTransform": "ienks", "LocalizedIEnKSBundle": "ienks",
    "ModelState": "assim_flow", "ObsSubset": "assim_flow", "StateError": "assim_flow", "ObservationError": "assim_flow",
    "assimilate_arrays": "assim_flow",
    "ShardedLetkf": "sharded", "block_partition": "sharded", "gather_blocks": "sharded",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module("." + _LAZY[name], __name__)
        return getattr(mod, name)
    raise AttributeError(name)
