"""gear_amd — MI355X-native batched collision validation for gear's planning hot path.

Host side (this package): URDF/chain flattening and a mirror of gear's
CollisionChecker / JointPathPlanner predicates over the C ABI in
include/gear_hip.h.  Device side: gear_amd/csrc (hand-written HIP for gfx950).
"""
from . import urdf  # noqa: F401
from .descriptors import (  # noqa: F401
    GH_CHECK_ENV, GH_CHECK_SELF, GH_FLAG_ENV, GH_FLAG_LIMIT, GH_FLAG_SELF,
    CollisionModel, Compound, ShapeSpec, parse_colon_separated_pairs,
)
