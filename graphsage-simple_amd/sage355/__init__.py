"""sage355 -- MI355X-native GraphSAGE sample/aggregate/update hot path.

Host-side mirror of graphsage/aggregators.py and graphsage/encoders.py of
zjzijielu/graphsage-simple over a C-ABI HIP library (include/sage355.h).
Importing the package is cheap and GPU-free; the HIP library is loaded on
first use and its absence is an error, never a fallback.
"""
__version__ = "0.1.0"
