"""Training-step timing: fused HIP step of all members vs. the reference's
formulation (one member at a time, torch ops + autograd + torch.optim.AdamW) on the same GPU.

    python tools/bench_train.py [--members 5] [--samples 10] [--rows 32] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import engine  # noqa: E402
from mcmc_amd.bayesian_ensemble import BayesianNetwork  # noqa: E402
from mcmc_amd.loss import peak_weighted_loss  # noqa: E402
from mcmc_amd.train import EnsembleTrainer  # noqa: E402


def torch_ops_step(model, opt, xb, yb, S):
    """bayesian_ensemble.py:165-180 as torch ops (what the reference executes per member)."""
    opt.zero_grad()
    preds = torch.stack([model(xb, sample=True).rename(None) for _ in range(S)], dim=0)
    loss = peak_weighted_loss(preds.mean(0), yb)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    return loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=5)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--features", type=int, default=11)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, S, R, F = a.members, a.samples, a.rows, a.features
    torch.manual_seed(0)
    members = [BayesianNetwork(F).to(dev) for _ in range(M)]
    mu, rho = engine.pack_members(members, dev)
    tr = EnsembleTrainer(mu, rho, F, S)
    xb = torch.randn(M, R, F, device=dev)
    yb = torch.randn(M, R, device=dev)

    def timed(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    it = [0]

    def fused():
        it[0] += 1
        tr.step(xb, yb, 1e-3, 0.9, seed=it[0])

    t_fused = timed(fused, a.steps)
    # per-kernel split with events on the same stream
    names = ["sample", "forward", "loss", "backward", "update"]
    lib, h = tr.eng.lib, tr.eng.ctx.handle
    from mcmc_amd.engine import _p, _stream
    Y, delta = tr._work(R)
    calls = [
        lambda: lib.mcmc_train_sample(h, _p(tr.mu), _p(tr.sigma), M, S, F, 2, _p(None), 1, _p(tr.eps), _p(tr.W), _p(tr.WT), _stream()),
        lambda: lib.mcmc_train_forward(h, _p(xb), R, F, M, S, _p(tr.WT), 2, _p(None), _p(None), _p(None), 1, 0.1, _p(Y), _stream()),
        lambda: lib.mcmc_train_loss(h, _p(Y), _p(yb), M, S, R, 2.0, 0.5, _p(delta), _p(tr.loss), _stream()),
        lambda: lib.mcmc_train_backward(h, _p(xb), R, F, M, S, _p(tr.W), _p(tr.WT), 2, _p(None), _p(None), _p(None), 1, 0.1, _p(delta),
                                        _p(tr.dW), _stream()),
        lambda: lib.mcmc_train_update(h, _p(tr.mu), _p(tr.rho), _p(tr.sigma), M, S, F, _p(tr.eps), _p(tr.dW), _p(tr.gmu), _p(tr.grho),
                                      _p(tr.adam_m), _p(tr.adam_v), _p(tr.norm_scratch), _p(None), 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0,
                                      5, _p(tr.gnorm), _stream()),
    ]
    split = {}
    for name, fn in zip(names, calls):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        split[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 50, 2)

    opts = [torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01) for m in members]
    for m in members:
        m.train()

    def eager():
        for m in range(M):
            torch_ops_step(members[m], opts[m], xb[m], yb[m], S)

    t_eager = timed(eager, max(10, a.steps // 10))
    flops = 3 * 2 * (F * 256 + 256 * 128 + 128 * 64 + 64) * R * M * S  # fwd + recompute-free bwd (2x) per step
    print(json.dumps({"members": M, "samples": S, "rows": R, "features": F,
                      "fused_ms_per_step": round(t_fused * 1e3, 4), "torch_ops_ms_per_step": round(t_eager * 1e3, 4),
                      "speedup": round(t_eager / t_fused, 1), "fused_gflops": round(flops / t_fused / 1e9, 1), **split}))


if __name__ == "__main__":
    main()
