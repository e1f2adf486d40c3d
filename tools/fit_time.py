"""Dev tool: wall time per training step inside fit_ensemble (host loop + kernels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.train import EnsembleTrainer, fit_ensemble
from mcmc_amd.engine import param_count
dev = torch.device("cuda", 0)
F, M, N = 11, 5, 365
g = torch.Generator().manual_seed(0)
mu = (torch.randn(M, param_count(F), generator=g) * 0.08).to(dev)
rho = torch.full((M, param_count(F)), -3.0, device=dev)
X = torch.randn(N, F, generator=g).to(dev); y = torch.randn(N, generator=g).to(dev)
tr = EnsembleTrainer(mu, rho, F, 10)
fit_ensemble(tr, X, y, epochs=5, log=lambda *a: None, patience=10**9)
torch.cuda.synchronize(); t0 = time.perf_counter()
ep = 100
fit_ensemble(tr, X, y, epochs=ep, log=lambda *a: None, patience=10**9)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{ep} epochs x 12 steps: {dt:.3f} s = {dt / (ep * 12) * 1e6:.1f} us per step")
