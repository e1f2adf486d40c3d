"""Dev tool: host-time profile (cProfile) of one main.py-style Bayesian task (train 400 epochs + predict)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import pipeline  # noqa: E402
from mcmc_amd.data import make_series_batch  # noqa: E402

dev = torch.device("cuda:0")
X, y = make_series_batch(1, seed=7, device="cpu", standardize=False)
feats, temp = X[:, :5].contiguous(), X[:, 5:].contiguous()
combos = pipeline.prepare_feature_combinations(5)[:3]
torch.manual_seed(0)
pipeline.process_bayesian_task(feats, temp, y, combos[0], dev, epochs=20, seed=1, log=lambda *_: None)  # warm-up
for combo in combos[1:]:
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    pipeline.process_bayesian_task(feats, temp, y, combo, dev, epochs=400, seed=1, log=lambda *_: None)
    torch.cuda.synchronize()
    pr.disable()
    print(f"task {combo}: {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
