"""morb_slam_amd — MI355X-native implementation of MORB_SLAM's per-frame front end (ORB extract + match) and
local-BA hot path.  The compute lives in libmorb_hip.so (hand-written HIP for gfx950, C ABI in
include/morb_hip.h); this package is the host-side mirror of the reference class surfaces."""
from .capi import KP_DTYPE, MorbError  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher  # noqa: F401
from .optimizer import BAProblem, Optimizer  # noqa: F401
