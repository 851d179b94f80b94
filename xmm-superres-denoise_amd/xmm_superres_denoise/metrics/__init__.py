from .xmm_metric_collection import XMMMetricCollection, get_in_metrics, get_metrics  # noqa: F401
