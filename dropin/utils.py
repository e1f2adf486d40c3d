"""Drop-in for the loss in /root/reference/utils.py:7-28 (`from utils import peak_weighted_loss`,
bayesian_ensemble.py:6, temporal_fusion.py).  Only the path's function is replaced; the
reference's metric/plot helpers (utils.py:30-76) are out of scope and stay in its own utils.py -
import this module as `utils` only in processes that need nothing else from it."""
from mcmc_amd.loss import peak_weighted_loss  # noqa: F401
