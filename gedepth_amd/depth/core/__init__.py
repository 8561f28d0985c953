from .evaluation import calculate, eval_metrics, metrics, pre_eval_to_metrics
from .utils import add_prefix

__all__ = ['calculate', 'eval_metrics', 'metrics', 'pre_eval_to_metrics', 'add_prefix']
