"""gmmloc_amd -- MI355X-native hot path of GMMLoc behind a C-ABI.

The product is ``libgmmloc_hip.so`` (hand-written HIP for gfx950, built in-tree by
``__graft_entry__.build()``); this package is the thin Python host side used by
the tests and the benchmark: ctypes bindings (``_lib``) and a mirror of the
reference's operator interface (``api``).  There is NO CPU fallback: importing
``api`` without the shared library raises, and every call needs a GPU.
"""
from .api import (  # noqa: F401
    Camera,
    Params,
    Context,
    GMM,
    GLError,
    ASSOC_BRUTE,
    ASSOC_KNN5_EUCLID,
    ASSOC_EXHAUSTIVE,
    optimize_current_pose,
    track_frames,
    track_frames_anchored,
)
