"""MI355X-native posterior-update hot path behind the BayesianOptimizer API of
Feuermagier/Beyond_Deep_Ensembles (src/algos): SVGD, SWAG, Bayes-by-Backprop
and iVON as hand-written HIP kernels for gfx950 behind a C ABI
(include/bde_hip.h), with drop-in PyTorch optimizer shells.

The kernels have no CPU or PyTorch fallback; see DESIGN.md.
"""
from ._lib import BdeLibraryError, LIB_PATH, is_built  # noqa: F401

__version__ = "0.1.0"

_LAZY = {
    "BayesianOptimizer": ".algo", "LastLayerBayesianOptimizer": ".algo",
    "SVGDOptimizer": ".svgd", "rbf": ".svgd",
    "SwagOptimizer": ".swag",
    "BBBOptimizer": ".bbb", "GaussianPrior": ".bbb", "MixturePrior": ".bbb",
    "GaussianParameter": ".util", "normal_like": ".util", "reset_model_params": ".util",
    "iVONOptimizer": ".ivon",
    "BBBLinear": ".bbb_layers", "BBBConv2d": ".bbb_layers", "make_module_bbb": ".bbb_layers",
    "DeepEnsemble": ".ensemble",
    "HipOps": ".ops",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name], __name__), name)
    raise AttributeError(name)
