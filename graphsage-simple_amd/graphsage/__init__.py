"""Drop-in shim: lets the reference's own import lines (graphsage/model.py:12-13)

    from graphsage.encoders import Encoder
    from graphsage.aggregators import MeanAggregator

resolve to the MI355X implementation when this directory precedes the reference
checkout on sys.path (or when these two files replace the reference's).  See
INTEGRATION.md.
"""
