"""Drop-in for /root/reference/models/temporal_fusion.py (`from models.temporal_fusion import TFTModel`,
main.py:12): re-exports the MI355X implementation."""
from mcmc_amd.temporal_fusion import GatedResidualNetwork, TemporalFusionTransformer, TFTModel, TimeDistributed  # noqa: F401
