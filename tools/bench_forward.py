#!/usr/bin/env python3
"""Micro-benchmark of the forward kernels alone (kernel tuning / rocprofv3 --pmc target).

    python tools/bench_forward.py --rows 365000 --pairs 1024 --iters 5 [--dtype bf16]
Prints ms per launch, units/s and TFLOP/s (algorithmic 87,680 FLOP/unit at input_dim 11)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import _lib  # noqa: E402
from mcmc_amd.engine import Engine, param_count  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=365_000)
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--F", type=int, default=11)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    from mcmc_amd.engine import resolve_dtype
    dt = resolve_dtype(args.dtype)
    torch.manual_seed(0)
    X = torch.randn(args.rows, args.F, device=dev)
    M = 8
    P = param_count(args.F)
    mu = torch.randn(M, P, device=dev) * 0.1
    rho = torch.full((M, P), -3.0, device=dev)
    S = (args.pairs + M - 1) // M
    xp = eng.pack_x(X, dt)
    wp = torch.empty(args.pairs * eng.wpack_bytes(args.F, dt), dtype=torch.uint8, device=dev)
    eng.sample_weights(mu, rho, S, args.F, 0, args.pairs, _lib.NOISE_PHILOX, None, 1, dt, wp)
    ldY = (args.rows + 63) // 64 * 64
    Y = torch.empty(args.pairs, ldY, device=dev)
    eng.forward(dt, xp, args.rows, args.F, wp, args.pairs, Y, ldY)
    torch.cuda.synchronize()
    evs = []
    for _ in range(args.iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.forward(dt, xp, args.rows, args.F, wp, args.pairs, Y, ldY)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    units = args.rows * args.pairs
    flop = 2 * (args.F * 256 + 256 * 128 + 128 * 64 + 64)
    med = ms[len(ms) // 2]
    print(f"forward {args.dtype}: rows={args.rows} pairs={args.pairs} median {med:.3f} ms (min {ms[0]:.3f})  "
          f"{units / med * 1e3:.4e} units/s  {units * flop / med / 1e9:.1f} TFLOP/s  checksum {float(Y[:, :args.rows].sum()):.6e}")


if __name__ == "__main__":
    main()
