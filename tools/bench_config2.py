#!/usr/bin/env python3
"""Latency of config 2 (365 rows x 5 members x 100 samples): the reference's own predict() shape."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import mc_predict

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "predict_cfg1.npz"))
d = torch.device("cuda", 0)
X, mu, rho = (torch.from_numpy(g[k]).to(d) for k in ("X", "mu", "rho"))
for dtype in ("fp32", "bf16"):
    for q in ((), (0.025, 0.975)):
        for _ in range(3):
            mc_predict(X, mu, rho, 100, dtype=dtype, seed=1, q_levels=q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for i in range(n):
            r = mc_predict(X, mu, rho, 100, dtype=dtype, seed=i, q_levels=q)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"config2 {dtype} quantiles={bool(q)}: {dt * 1e3:.3f} ms per predict  {365 * 500 / dt:.3e} units/s")

from mcmc_amd.engine import PredictPlan
for dtype in ("fp32", "bf16"):
    for q in ((), (0.025, 0.975)):
        plan = PredictPlan(X.shape[0], X.shape[1], mu.shape[0], 100, device=d, dtype=dtype, q_levels=q)
        for _ in range(3):
            plan(X, mu, rho, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            r = plan(X, mu, rho, i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"config2 {dtype} quantiles={bool(q)} hipGraph plan: {dt * 1e3:.3f} ms per predict  {365 * 500 / dt:.3e} units/s")
