"""sage355 -- MI355X-native GraphSAGE sample/aggregate/update hot path.

Host-side mirror of graphsage/aggregators.py and graphsage/encoders.py of
zjzijielu/graphsage-simple over a C-ABI HIP library (include/sage355.h).
Importing the package is cheap and GPU-free; the HIP library is loaded on
first use and its absence is an error, never a fallback.
"""
__version__ = "0.1.0"

import os as _os

# RolePipeline runs the stages of consecutive forwards on four HIP streams.  ROCm gives a process 4 hardware queues by
# default and maps all HIP streams onto them round-robin; two role streams on one queue serialise (measured on MI355X:
# 82.6 us per forward with 4 queues, 69.1 us with 6 or more).  The HIP runtime reads this when it initialises, so it is set
# here, at import time, unless the user has chosen a value -- or does not want a library to touch the process environment at all
# (SAGE355_KEEP_ENV=1: then give the process GPU_MAX_HW_QUEUES >= 6 yourself if you use RolePipeline).
if _os.environ.get("SAGE355_KEEP_ENV", "0") != "1":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
