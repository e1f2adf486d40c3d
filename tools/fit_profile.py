"""Dev tool: where the host time of fit_ensemble goes (cProfile) and the kernel time of the same loop."""
import cProfile
import os
import pstats
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
torch.cuda.synchronize()
ep = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pr = cProfile.Profile()
t0 = time.perf_counter(); e0.record()
pr.enable()
fit_ensemble(tr, X, y, epochs=ep, log=lambda *a: None, patience=10**9)
pr.disable()
e1.record(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{ep} epochs x 12 steps: wall {dt / (ep * 12) * 1e6:.1f} us per step, GPU events {e0.elapsed_time(e1) / (ep * 12) * 1e3:.1f} us per step (under cProfile)")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
