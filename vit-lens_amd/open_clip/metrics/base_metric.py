class BaseMetric(object):
    """open_clip/metrics/base_metric.py:1-13."""

    def __init__(self):
        pass

    def initialize(self):
        raise NotImplementedError

    def compute(self, models, sample):
        raise NotImplementedError

    def merge_results(self, output_predict=False):
        raise NotImplementedError
