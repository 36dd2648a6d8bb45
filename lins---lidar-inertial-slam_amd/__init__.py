"""MI355X-native IESKF update path of LINS.

  host   — CPU host pieces (front-end, StatePredictor mirror, synthetic scans)
  ieskf  — binding of liblins_ieskf.so: the HIP kernels behind the C ABI of
           include/lins_ieskf.h.  Importing it never falls back to a CPU path:
           a missing library or GPU raises.
"""
from ._ctypes_defs import (CORR_DTYPE, POSE_DTYPE, Params, Result, ScanPair, default_params)  # noqa: F401
