"""Drop-in for /root/reference/utils.py as main.py imports it: `from utils import peak_weighted_loss`
(bayesian_ensemble.py:6, temporal_fusion.py) is the path's fused loss, and `from utils import evaluate_predictions`
(main.py:12) is the metric helper with the reference's signature and printing (utils.py:30-40) over the
pipeline's restatement.  The two matplotlib helpers (utils.py:42-76) are unused by main.py and out of scope."""
from mcmc_amd.loss import peak_weighted_loss  # noqa: F401
from mcmc_amd.pipeline import evaluate_predictions as _evaluate_predictions


def evaluate_predictions(y_true, y_pred, model_name=""):
    """utils.py:30-40: {"rmse", "mape"} plus the reference's three printed lines."""
    out = _evaluate_predictions(y_true, y_pred)
    print(f"\n{model_name} Performance:")
    print(f"RMSE: {out['rmse']:.2f}")
    print(f"MAPE: {out['mape']:.2f}%")
    return out
