"""TFT training-step and predict timing: fused HIP step vs the reference's
formulation (full [B,T,T] attention, torch ops + autograd + torch.optim.AdamW) on the same GPU.

    python tools/bench_tft.py [--steps 200]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.loss import peak_weighted_loss  # noqa: E402
from mcmc_amd.temporal_fusion import TemporalFusionTransformer, TFTEngine  # noqa: E402


def torch_forward(net, x, train):
    """temporal_fusion.py:120-153 as torch ops on the container's parameters."""
    def grn(g, u):
        h = F.elu(F.linear(u, g.fc1.module.weight, g.fc1.module.bias))
        h = F.dropout(h, 0.1, train)
        h = F.elu(F.linear(h, g.fc2.module.weight, g.fc2.module.bias))
        gate = torch.sigmoid(F.linear(torch.cat([u, h], -1), g.gate.module.weight, g.gate.module.bias))
        return g.layer_norm(gate * h + (1 - gate) * u)
    h = F.linear(x, net.input_projection.module.weight, net.input_projection.module.bias)
    h = grn(net.temporal_grn, grn(net.feature_grn, h))
    t = h.permute(1, 0, 2)
    a, _ = net.attention(t, t, t)
    h = grn(net.final_grn, a.permute(1, 0, 2))
    return F.linear(h, net.output_projection.module.weight, net.output_projection.module.bias)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--features", type=int, default=11)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = TemporalFusionTransformer(a.features).to(dev)
    N, T, B = 365, 30, 32
    X = torch.randn(N, a.features, device=dev)
    y = torch.randn(N, device=dev)
    te = TFTEngine(net.flat_parameters(), a.features, T)
    start = torch.randperm(N - T + 1)[:B].to(device=dev, dtype=torch.int32)
    yb = y[start.long() + T - 1]

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
        te.step(X, start, yb, 1e-3, 0.9, seed=it[0])

    t_fused = timed(fused, a.steps)
    all_start = torch.arange(N - T + 1, dtype=torch.int32, device=dev)
    t_pred = timed(lambda: te.forward(X, all_start), a.steps)

    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.01)
    idx = start.long().unsqueeze(1) + torch.arange(T, device=dev).unsqueeze(0)
    xb = X[idx]
    net.train()

    def eager():
        opt.zero_grad()
        out = torch_forward(net, xb, True)
        loss = peak_weighted_loss(out[:, -1, 0], yb)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()

    t_eager = timed(eager, max(20, a.steps // 4))
    net.eval()
    xall = X[all_start.long().unsqueeze(1) + torch.arange(T, device=dev).unsqueeze(0)]

    def eager_pred():
        with torch.no_grad():
            for i in range(0, xall.shape[0], 32):
                torch_forward(net, xall[i:i + 32], False)[:, -1, 0]

    t_epred = timed(eager_pred, 20)
    print(json.dumps({"batch": B, "seq_len": T, "features": a.features, "fused_train_ms_per_step": round(t_fused * 1e3, 4),
                      "torch_ops_train_ms_per_step": round(t_eager * 1e3, 4), "train_speedup": round(t_eager / t_fused, 1),
                      "fused_predict_ms_336_windows": round(t_pred * 1e3, 4), "torch_ops_predict_ms": round(t_epred * 1e3, 4),
                      "predict_speedup": round(t_epred / t_pred, 1)}))


if __name__ == "__main__":
    main()
