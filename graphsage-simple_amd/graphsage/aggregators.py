from sage355.aggregators import MeanAggregator  # noqa: F401
