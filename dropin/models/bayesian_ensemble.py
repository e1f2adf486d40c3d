"""Drop-in for /root/reference/models/bayesian_ensemble.py: put `dropin/` ahead of the
reference checkout on sys.path and `from models.bayesian_ensemble import GPUBayesianEnsemble`
(main.py:11) resolves to the MI355X path.  See INTEGRATION.md."""
from mcmc_amd.bayesian_ensemble import BayesianLinear, BayesianNetwork, GPUBayesianEnsemble  # noqa: F401
