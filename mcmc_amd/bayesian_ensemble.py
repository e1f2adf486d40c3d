"""Host-side mirror of /root/reference/models/bayesian_ensemble.py for the MI355X path.

Same class names, constructor / ``train`` / ``predict`` / ``forward`` signatures, attribute names
(``models``, ``hidden_layers[i].weight_mu`` ...), named-tensor outputs and error behaviour as the
reference, so ``main.py`` (main.py:148-177) runs against it unchanged.  ``predict`` - the hot
path - runs entirely in libmcmc_hip.so (engine.mc_predict); it needs a HIP device and raises
otherwise.  New knobs are keyword-only with reference-preserving defaults.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine


class BayesianLinear(nn.Module):
    """Mean-field Gaussian linear layer (bayesian_ensemble.py:10-49): sigma = exp(rho)."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight_mu = nn.Parameter(torch.zeros(out_features, in_features))
        self.weight_rho = nn.Parameter(torch.zeros(out_features, in_features))
        self.bias_mu = nn.Parameter(torch.zeros(out_features))
        self.bias_rho = nn.Parameter(torch.zeros(out_features))
        self.reset_parameters()

    def reset_parameters(self):
        # bayesian_ensemble.py:27-31
        nn.init.kaiming_normal_(self.weight_mu, mode="fan_in", nonlinearity="relu")
        nn.init.constant_(self.weight_rho, -3)
        nn.init.constant_(self.bias_mu, 0.0)
        nn.init.constant_(self.bias_rho, -3)

    def forward(self, x: torch.Tensor, sample: bool = False) -> torch.Tensor:
        x = x.rename(None)
        if x.size(-1) != self.in_features:
            raise ValueError(f"Expected {self.in_features} input features, got {x.size(-1)}")
        if sample:
            weight = self.weight_mu + torch.randn_like(self.weight_mu) * torch.exp(self.weight_rho)
            bias = self.bias_mu + torch.randn_like(self.bias_mu) * torch.exp(self.bias_rho)
        else:
            weight, bias = self.weight_mu, self.bias_mu
        return F.linear(x, weight, bias)


class BayesianNetwork(nn.Module):
    """input_dim -> 256 -> 128 -> 64 -> 1 with ELU + Dropout(0.1) (bayesian_ensemble.py:52-92).

    This module is the trainable container of a member's (mu, rho); the ensemble's ``predict``
    does not call its ``forward`` - it packs the parameters and runs the fused HIP path."""

    def __init__(self, input_dim: int, hidden_sizes: Sequence[int] = (256, 128, 64)):
        super().__init__()
        if tuple(hidden_sizes) != engine.HIDDEN:
            raise ValueError(f"the HIP kernels are built for hidden sizes {engine.HIDDEN}, got {tuple(hidden_sizes)}")
        self.input_dim = input_dim
        self.hidden_sizes = list(hidden_sizes)
        self.hidden_layers = nn.ModuleList()
        self.hidden_layers.append(BayesianLinear(input_dim, hidden_sizes[0]))
        for i in range(len(hidden_sizes) - 1):
            self.hidden_layers.append(BayesianLinear(hidden_sizes[i], hidden_sizes[i + 1]))
        self.output_layer = BayesianLinear(hidden_sizes[-1], 1)
        self.dropout = nn.Dropout(0.1)

    def forward(self, x: torch.Tensor, sample: bool = False) -> torch.Tensor:
        x = x.rename(None)
        if x.size(-1) != self.input_dim:
            raise ValueError(f"Expected {self.input_dim} input features, got {x.size(-1)}")
        for layer in self.hidden_layers:
            x = layer(x, sample)
            x = F.elu(x)
            x = self.dropout(x)
        output = self.output_layer(x, sample)
        return output.squeeze(-1).refine_names("batch")


class GPUBayesianEnsemble(nn.Module):
    """Drop-in for the reference class (bayesian_ensemble.py:95-256)."""

    def __init__(self, input_dim: int, n_models: int = 5, device: str = None, batch_size: int = 32, *,
                 compute_dtype: str = "fp32", seed: Optional[int] = None):
        super().__init__()
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.input_dim = input_dim
        self.n_models = n_models
        self.batch_size = batch_size
        self.models = []  # plain list like the reference (bayesian_ensemble.py:105)
        self.compute_dtype = compute_dtype
        engine.resolve_dtype(compute_dtype)
        self._seed = seed
        self._calls = 0

    # ------------------------------------------------------------------ train (row f1)
    def train(self, X: torch.Tensor = True, y: torch.Tensor = None, epochs: int = 400, num_samples: int = 10):
        """Train the ensemble (bayesian_ensemble.py:107-209): same initialisation, optimiser (AdamW
        1e-3 / wd 0.01), OneCycleLR schedule, clip_grad_norm_(1.0), per-member shuffles and early
        stopping (patience 25) - but all members advance together, one fused HIP step per batch
        (mcmc_amd.train.EnsembleTrainer) instead of the reference's member-after-member torch graph.
        Weight noise and dropout masks come from the keyed Philox streams.  With ``compute_dtype`` bf16 / bf16_fast and a
        ``batch_size`` of 2,048 rows or more the step runs on the bf16-operand kernels (fp32 master weights and AdamW)."""
        if isinstance(X, bool) and y is None:  # nn.Module.train(mode) - the reference shadows it; keep both usable
            return super().train(X)
        from .train import EnsembleTrainer, fit_ensemble
        X = X.cpu().refine_names("time", "features")
        y = y.cpu().refine_names("time")
        if X.size("features") != self.input_dim:
            raise ValueError(f"Input dimension mismatch. Expected {self.input_dim}, got {X.size('features')}")
        if torch.device(self.device).type != "cuda":
            raise RuntimeError(f"GPUBayesianEnsemble needs a HIP device, got {self.device}; there is no CPU path "
                               "(the reference requires a GPU too: main.py:225-226)")
        self.models = []
        members = [BayesianNetwork(self.input_dim).to(self.device) for _ in range(self.n_models)]
        mu, rho = engine.pack_members(members, torch.device(self.device))
        # bf16 ensembles use the bf16-operand training kernels where those exist: batches of more than one row block
        # (>= 2,048 rows; the reference's batch_size of 32 runs the fp32 one-call step either way)
        trainer = EnsembleTrainer(mu, rho, self.input_dim, num_samples,
                                  compute_dtype="bf16" if self.compute_dtype in ("bf16", "bf16_fast") else "fp32")
        self.epochs_trained = fit_ensemble(trainer, X.rename(None), y.rename(None), epochs=epochs, batch_size=self.batch_size,
                                           seed=self._next_seed())
        trainer.unpack_into(members)
        self.models = members
        print(f"Finished training {self.n_models} models")

    # ------------------------------------------------------------------ predict (the hot path)
    def _next_seed(self) -> int:
        if self._seed is not None:
            self._calls += 1
            return (self._seed + 0x9E3779B97F4A7C15 * self._calls) & 0xFFFFFFFFFFFFFFFF
        # follow torch's global generator like the reference's randn_like does
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def predict_full(self, X: torch.Tensor, num_samples: int = 100, *, dtype: Optional[str] = None,
                     quantiles: Sequence[float] = (0.025, 0.975), nbins: int = 128, noise: str = "philox",
                     eps: Optional[torch.Tensor] = None, seed: Optional[int] = None, group=False,
                     return_samples: bool = False, chunk_pairs: Optional[int] = None, mc_dropout: bool = False,
                     mask_seed: Optional[int] = None) -> engine.PredictResult:
        """predict() plus the extensions (interval quantiles, injected noise, bf16, sharding, MC dropout:
        ``mc_dropout=True`` keeps Dropout(0.1) live with Philox masks; the reference's predict() is eval-mode)."""
        if not self.models:
            raise RuntimeError("no trained members: call train() or fill .models first")
        for model in self.models:
            model.eval()
        X = X.to(self.device).rename(None)
        if X.dim() != 2 or X.size(-1) != self.input_dim:
            raise ValueError(f"Expected {self.input_dim} input features, got {X.size(-1)}")
        mu, rho = engine.pack_members(self.models, X.device)
        with torch.no_grad():
            return engine.mc_predict(X, mu, rho, num_samples, dtype=dtype or self.compute_dtype, noise=noise, eps=eps,
                                     seed=self._next_seed() if seed is None else seed, q_levels=tuple(quantiles), nbins=nbins,
                                     group=group, return_samples=return_samples, chunk_pairs=chunk_pairs, mc_dropout=mc_dropout,
                                     mask_seed=mask_seed)

    def predict(self, X: torch.Tensor, num_samples: int = 100) -> Tuple[torch.Tensor, torch.Tensor]:
        """Generate predictions with uncertainty estimates (bayesian_ensemble.py:212-252):
        (mean, unbiased std) over all n_models x num_samples passes, fp32, named ('batch',)."""
        res = self.predict_full(X, num_samples, quantiles=())
        return res.mean.refine_names("batch"), res.std.refine_names("batch")

    def predict_interval(self, X: torch.Tensor, num_samples: int = 100, level: float = 0.95, **kw):
        """(mean, std, lower, upper): central ``level`` interval from the streaming histogram."""
        a = (1.0 - level) / 2.0
        res = self.predict_full(X, num_samples, quantiles=(a, 1.0 - a), **kw)
        return (res.mean.refine_names("batch"), res.std.refine_names("batch"),
                res.quantiles[0].refine_names("batch"), res.quantiles[1].refine_names("batch"))

    def forward(self, X: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.predict(X)
