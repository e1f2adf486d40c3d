#!/usr/bin/env python3
"""Micro-benchmark of the fused peak-weighted loss (piece 3), forward + backward, one GPU.

    python tools/bench_loss.py [--n 36500000] [--iters 20]

Times (events on the launch stream) the data-parallel three-call path (mcmc_loss_stats / _forward / _backward, fp64 atomics,
caller-zeroed sums), the single-rank one-call path (mcmc_loss_fused: one persistent launch that keeps part of the slice on chip)
and its three-launch form (mcmc_loss_fused_static) at the config-5 total (100k series x 365) and at the
per-GPU share of it (1/8).  Prints one JSON line per size: ms, GB/s of the 20 B/element algorithmic bytes and
their fraction of the 8 TB/s HBM spec.  rocprofv3 target for profiles/r02_loss_kernel_stats.csv."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import _lib  # noqa: E402
from mcmc_amd.engine import Engine, _p, _stream  # noqa: E402
from mcmc_amd.loss import peak_weighted_loss  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="*", default=[36_500_000, 4_562_500])
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    lib, h = eng.lib, eng.ctx.handle
    for n in args.n:
        g = torch.Generator(device=dev).manual_seed(n)
        t = torch.randn(n, device=dev, generator=g)
        p = t + 0.3 * torch.randn(n, device=dev, generator=g)
        grad = torch.empty_like(p)
        stats = torch.zeros(4, dtype=torch.float64, device=dev)
        part = torch.zeros(4, dtype=torch.float64, device=dev)
        loss = torch.empty(1, device=dev)
        ws = torch.empty(int(lib.mcmc_loss_workspace_bytes()), dtype=torch.uint8, device=dev)

        def three_calls():
            stats.zero_()
            part.zero_()
            _lib.check(lib.mcmc_loss_stats(h, _p(p), _p(t), n, _p(stats), _stream()), "stats")
            _lib.check(lib.mcmc_loss_forward(h, _p(p), _p(t), n, n, _p(stats), 2.0, 0.5, _p(part), _stream()), "fwd")
            _lib.check(lib.mcmc_loss_backward(h, _p(p), _p(t), n, n, _p(stats), _p(part), 2.0, 0.5, 1.0, _p(grad), _stream()), "bwd")

        def fused():
            _lib.check(lib.mcmc_loss_fused(h, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(loss), _p(grad), _p(ws), _stream()), "fused")

        def fused_static():  # the three launches of round 2's first half, on the library's static scratch
            _lib.check(lib.mcmc_loss_fused_static(h, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(loss), _p(grad), _stream()), "fused_static")

        def autograd():
            pr = p.detach().requires_grad_(True)
            peak_weighted_loss(pr, t).backward()

        out = {"n": n, "algorithmic_bytes": 20 * n}
        for name, fn in (("three_calls", three_calls), ("fused", fused), ("fused_static", fused_static), ("autograd_wrapper", autograd)):
            ms = timed(fn, args.iters)
            gbps = 20 * n / ms / 1e6
            out[name] = {"ms": round(ms, 4), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 3)}
        three_calls()
        g3 = grad.clone()
        l3 = float(part[0] / n)
        fused()
        out["fused_vs_three_calls"] = {"loss_rel": abs(float(loss) - l3) / abs(l3), "grad_max_abs": float((grad - g3).abs().max())}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
