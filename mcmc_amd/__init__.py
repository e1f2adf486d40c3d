"""mcmc_amd - MI355X-native Monte Carlo predictive path of the fintkz/mcmc Bayesian ensemble.

Only what the path needs: the C-ABI library (csrc/ -> lib/libmcmc_hip.so), its ctypes binding
(_lib), the chunk-loop driver (engine) and the mirror of the reference's Python interface
(bayesian_ensemble, loss), plus the widened rows of SURVEY.md 8f: train (training step),
pipeline (main.py glue), data (batched generator), temporal_fusion (TFT).  See DESIGN.md and
INTEGRATION.md.
"""
from .bayesian_ensemble import BayesianLinear, BayesianNetwork, GPUBayesianEnsemble
from .engine import PredictResult, mc_predict, pack_members
from .loss import peak_weighted_loss
from .temporal_fusion import TFTModel

__all__ = ["BayesianLinear", "BayesianNetwork", "GPUBayesianEnsemble", "PredictResult", "mc_predict", "pack_members",
           "peak_weighted_loss", "TFTModel"]
