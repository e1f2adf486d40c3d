#!/usr/bin/env python3
"""Micro-benchmark of the streaming reduction (reduce_update) and the loss passes: achieved GB/s.

    python tools/bench_reduce.py --rows 3650000 --pairs 1024 --nbins 128     # a pair chunk over all rows
    python tools/bench_reduce.py --rows 262144 --pairs 8192 --no-loss        # the bench's shape: a row block of all pairs
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.engine import Engine  # noqa: E402


def timeit(fn, iters):
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
    ap.add_argument("--rows", type=int, default=3_650_000)
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--nbins", type=int, default=128)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--loss-n", type=int, default=36_500_000)
    ap.add_argument("--no-loss", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    B, C = args.rows, args.pairs
    ldY = (B + 63) // 64 * 64
    Y = torch.randn(C, ldY, device=dev) * 0.5 + 0.3
    pilot = eng.new_state(B)
    eng.reduce_update(Y, ldY, min(C, 64), B, pilot)
    st = eng.new_state(B, args.nbins)
    eng.set_edges(B, pilot.n, pilot.sum, pilot.sumsq, st, 6.0)

    def upd_hist():
        st.hist.zero_()
        eng.reduce_update(Y, ldY, C, B, st)

    def zero_only():
        st.hist.zero_()

    st2 = eng.new_state(B)
    ms_h = timeit(upd_hist, args.iters) - timeit(zero_only, args.iters)
    ms_m = timeit(lambda: eng.reduce_update(Y, ldY, C, B, st2, with_hist=False), args.iters)
    sample_bytes = 4.0 * C * B
    state_bytes = 2.0 * B * (16 + 8 / 2 + 2 * args.nbins)  # read+write moments and histogram, read lo/inv_w
    print(f"reduce_update hist  : {ms_h:.3f} ms  samples {sample_bytes / ms_h / 1e6:.0f} GB/s  samples+state {(sample_bytes + state_bytes) / ms_h / 1e6:.0f} GB/s")
    print(f"reduce_update moments: {ms_m:.3f} ms  samples {sample_bytes / ms_m / 1e6:.0f} GB/s")
    ms_f = timeit(lambda: eng.finalize(B, st, (0.025, 0.975)), args.iters)
    print(f"finalize            : {ms_f:.3f} ms  state {(B * (16 + 8 + 2 * args.nbins + 16)) / ms_f / 1e6:.0f} GB/s")
    # copy ceiling on this device for reference
    Z = torch.empty_like(Y)
    ms_c = timeit(lambda: Z.copy_(Y), args.iters)
    print(f"torch copy (r+w)    : {ms_c:.3f} ms  {2 * sample_bytes / ms_c / 1e6:.0f} GB/s")
    del Z, Y
    if args.no_loss:
        return
    # loss passes
    from mcmc_amd import peak_weighted_loss
    n = args.loss_n
    t = torch.randn(n, device=dev)
    p = (t + 0.3 * torch.randn(n, device=dev)).requires_grad_(True)

    def fwd_bwd():
        p.grad = None
        peak_weighted_loss(p, t).backward()

    ms_l = timeit(fwd_bwd, args.iters)
    print(f"loss fwd+bwd N={n}: {ms_l:.3f} ms  {20.0 * n / ms_l / 1e6:.0f} GB/s at 20 B/element (algorithmic), {28.0 * n / ms_l / 1e6:.0f} GB/s at 28 B/element (moved)")


if __name__ == "__main__":
    main()
