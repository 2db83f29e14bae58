"""ellp_amd — MI355X-native revised-simplex pivot engine behind kehlert/ellp's solver API.

Only the hot path is here: `ellp_amd._engine` binds the C ABI of include/ellp_hip.h
(libellp_hip.so, hand-written HIP for gfx950).  There is no CPU implementation in this package.
"""
from . import _engine  # noqa: F401

__all__ = ["_engine"]
