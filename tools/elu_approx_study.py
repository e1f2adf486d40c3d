"""CPU study: how much does an ELU whose exponential is built in the bf16 exponent field (no v_exp_f32)
move the bf16 predict's mean / std?  Emulates the bf16 forward kernel (bf16 operands, fp32 accumulate)
with (A) the exact ELU and (B) the exponent-field ELU in the shifted domain h'' = log2e*(elu(z)+1), on the
config-1/2 fixture, against the fp32 golden mean / std.  Test infrastructure only (imports oracle/).

    python tools/elu_approx_study.py [--shift S]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bnn_oracle as orc  # noqa: E402

C = 1.4426950408889634


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def bf_bits(x):
    return x.to(torch.bfloat16).view(torch.int16).to(torch.int32)


def from_bits(b):
    return b.to(torch.int16).view(torch.bfloat16).to(torch.float32)


def act_exact(a):
    """a = log2e*z -> bf16(log2e*elu(z)) (the shipped kernel)."""
    e = torch.exp2(torch.clamp(a, max=0.0))
    return bf(torch.maximum(a, e * np.float32(C) - np.float32(C)))


def act_field(v, sigma):
    """v = a + c -> bf16 bits of h'' = max(v, c*2^min(a,0)) with the exponential built in the exponent field:
    bits = 128*(a + log2(c) + 127) + sigma, capped at bits(c); max on the bit patterns as int16."""
    k1 = np.float32(128.0 / 65535.0)
    k0 = np.float32((128.0 * (127.0 + np.log2(C) - C) + sigma) / 65535.0)
    x = torch.clamp(v * k1 + k0, 0.0, 1.0)
    y = torch.round(x * 65535.0).to(torch.int32)
    cap = int(bf_bits(torch.tensor(C, dtype=torch.float32)))
    y = torch.minimum(y, torch.tensor(cap, dtype=torch.int32))
    p = bf_bits(v)
    p = torch.where(p >= 32768, p - 65536, p)  # int16 view
    return from_bits(torch.maximum(p, y))


def forward(X, w, mode, sigma):
    """w: list of (W [out,in], b [out]) fp32 effective weights of one pair."""
    (W0, b0), (W1, b1), (W2, b2), (W3, b3) = w
    c = np.float32(C)
    xb = bf(X)
    W0s, b0s = bf(W0 * c), bf(b0 * c)  # layer 0 and its bias column travel as bf16, log2e-scaled
    W1b, W2b = bf(W1), bf(W2)
    if mode == "exact":
        h = act_exact(xb @ W0s.T + b0s)
        h = act_exact(h @ W1b.T + b1 * c)
        h = act_exact(h @ W2b.T + b2 * c)
        return (h @ (W3 * np.float32(1 / C)).T + b3).squeeze(-1)
    # shifted domain: accumulators hold a + c; the next layer's bias absorbs -c*rowsum(W)
    h = act_field(xb @ W0s.T + bf(b0 * c + c), sigma)
    h = act_field(h @ W1b.T + (b1 * c + c - c * W1b.sum(1)), sigma)
    h = act_field(h @ W2b.T + (b2 * c + c - c * W2b.sum(1)), sigma)
    w3 = bf(W3 * np.float32(1 / C))  # layer 3 on the MFMA: bf16 weights, bias absorbs the shift
    return (h @ w3.T + (b3 - c * w3.sum(1))).squeeze(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shift", type=float, nargs="*", default=[-5.5, -7.0, -8.0, -9.0, -11.0])
    ap.add_argument("--pairs", type=int, default=500)
    args = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", "predict_cfg1.npz"))
    X, mu, rho = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho"))
    M, S = int(g["M"]), int(g["S"])
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, X.shape[1])  # [M,S,P]
    offs = orc.param_offsets(X.shape[1])
    dims = orc.layer_dims(X.shape[1])
    # activation-function error on a grid
    a = torch.linspace(-12, 3, 200001)
    ex = torch.maximum(a, torch.exp2(torch.clamp(a, max=0)) * C - C)
    for s in args.shift:
        ap_ = act_field(a + np.float32(C), s) - np.float32(C)
        d = ap_ - ex
        print(f"sigma {s:5.1f}: activation error (log2e domain) max {d.abs().max():.4f} mean {d.mean():+.5f} rms {d.pow(2).mean().sqrt():.4f}"
              f"   [exact-bf16 rounding alone: max {(act_exact(a) - ex).abs().max():.4f}]")
    ref_mean, ref_std = torch.from_numpy(g["mean"]), torch.from_numpy(g["std"])
    scale = torch.maximum(ref_mean.abs(), ref_std)
    pairs = [(m, s) for m in range(M) for s in range(S)][: args.pairs]
    for mode, sig in [("exact", 0.0)] + [("field", s) for s in args.shift]:
        ys, y32 = [], []
        for m, s in pairs:
            w = []
            for (wo, bo, _, _), (fi, fo) in zip(offs, dims):
                Wf = (mu[m] + eps[m, s] * torch.exp(rho[m]))
                w.append((Wf[wo:wo + fi * fo].reshape(fo, fi), Wf[bo:bo + fo]))
            ys.append(forward(X, w, mode, sig))
        Y = torch.stack(ys)
        mean, std = Y.mean(0), Y.std(0)
        if len(pairs) == M * S:
            em = ((mean - ref_mean).abs() / scale).max()
            es = ((std - ref_std).abs() / ref_std).max()
            print(f"{mode:6s} sigma {sig:5.1f}: max |mean err|/max(|mean|,std) {em:.2e}   max rel std err {es:.2e}   "
                  f"mean bias {((mean - ref_mean) / scale).mean():+.2e}")
        else:
            print(mode, sig, float(mean.abs().mean()), float(std.mean()))


if __name__ == "__main__":
    main()
