"""MI355X-native SEPAIHRD likelihood path (adjo0043/Mathematical-Modeling-Of-Infectious-Diseases-V1).

Layout:
  csrc/     hand-written HIP kernels for gfx950 + the C ABI (include/sepaihrd_hip.h)
  host/     C++ mirror of the reference's plug-in surface above the C ABI
  *.py      ctypes plumbing for tests / bench (no compute, no fallback)

The directory name is not a Python identifier; import it through ``mmid_amd_loader.load()``.
"""
from .problem import (SEPAIHRDProblem, resolve_param_name, widen_age_classes, restrict_age_classes, SOLVER_DOPRI5,
                      SOLVER_CASH_KARP54, CONSTRAINT_CLAMP, CONSTRAINT_REFLECT, ARITH_STRICT, ARITH_FMA,
                      PRECISION_F64, PRECISION_F32)
from . import config_io, hipabi, hostabi, draws, parallel, workloads
from .hipabi import HipObjective, load_library, LIB_PATH, LOWEST
from .hostabi import HostObjective

__all__ = ["SEPAIHRDProblem", "resolve_param_name", "widen_age_classes", "restrict_age_classes", "HipObjective", "load_library",
           "config_io", "hipabi", "hostabi", "draws", "HostObjective", "LIB_PATH", "LOWEST", "SOLVER_DOPRI5", "SOLVER_CASH_KARP54",
           "CONSTRAINT_CLAMP", "CONSTRAINT_REFLECT", "ARITH_STRICT", "ARITH_FMA", "PRECISION_F64", "PRECISION_F32"]
