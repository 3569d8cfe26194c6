"""archnemesis_dist_amd -- MI355X-native engine for the archNEMESIS radiative-transfer hot path
(ForwardModel_0.nemesisfm / nemesisfmg / jacobian_nemesis -> CIRSrad).  See DESIGN.md.

(The task brief names the package directory `archnemesis-dist_amd`; a hyphen is not importable,
so the package is `archnemesis_dist_amd`.)
"""
from ._lib import AnsfmError, build, load, LIB_PATH  # noqa: F401
from .engine import AnsfmEngine  # noqa: F401

__all__ = ["AnsfmEngine", "AnsfmError", "build", "load", "LIB_PATH"]
