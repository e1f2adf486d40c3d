"""End-to-end wall time of main.py's Bayesian leg at the reference's defaults (config 1 shape):
365-day series, F = 11, 5 members, 400 epochs x 12 batches of 32, 10 train samples, 100 predict
samples.  The reference's work log reports 264-836 s per feature subset for this leg on its GPU
(SURVEY.md section 6).

    python tools/bench_pipeline.py [--epochs 400] [--combos 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import pipeline  # noqa: E402
from mcmc_amd.data import make_series_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=400)
    ap.add_argument("--combos", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    X, y = make_series_batch(1, seed=7, device="cpu", standardize=False)
    feats, temp = X[:, :5].contiguous(), X[:, 5:].contiguous()
    combos = pipeline.prepare_feature_combinations(5)[:a.combos]
    torch.manual_seed(0)
    out = []
    for combo in combos:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        name, rec = pipeline.process_bayesian_task(feats, temp, y, combo, dev, epochs=a.epochs, seed=1, log=lambda *_: None)
        torch.cuda.synchronize()
        out.append({"combo": name, "seconds": round(time.perf_counter() - t0, 3), "mape": round(rec["bayesian"]["metrics"]["mape"], 3),
                    "rmse": round(rec["bayesian"]["metrics"]["rmse"], 2)})
    print(json.dumps({"epochs": a.epochs, "tasks": out}))


if __name__ == "__main__":
    main()
