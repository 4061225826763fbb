"""MI355X-native engine for the particle-filter hot path of LowLevelParticleFilters.jl.

Host-side mirror of the reference's ParticleFilter / AdvancedParticleFilter API over the C ABI of
libllpf_hip.so (include/llpf.h).  See api.py for the mirrored names.
"""
from . import _structs  # noqa: F401
from . import _capi  # noqa: F401
from .api import *  # noqa: F401,F403
from . import tracing  # noqa: F401
from .tracing import traced_dynamics, ifelse, rk4  # noqa: F401
