#!/usr/bin/env python3
"""Dev check of the bf16 large-batch training kernels against the fp32 path: per-layer gradient errors, timing.
    python tools/check_train_bf16.py [--rows 4200] [--F 11] [--M 2] [--S 2] [--mask host|philox|none]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.train import EnsembleTrainer  # noqa: E402
from oracle import bnn_oracle as orc  # noqa: E402  (layout helper only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4200)
    ap.add_argument("--F", type=int, default=11)
    ap.add_argument("--M", type=int, default=2)
    ap.add_argument("--S", type=int, default=2)
    ap.add_argument("--mask", default="philox")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--act-gib", type=float, default=0.0, help="bound of the saved-activation buffer in GiB (default: the trainer's)")
    ap.add_argument("--nb", type=int, default=0, help="bf16: pin the row blocks per pair")
    ap.add_argument("--stamps", action="store_true", help="library built with -DMCMC_TB_STAMP: print the backward kernel's phase times")
    ap.add_argument("--no-save", action="store_true", help="bf16: recompute the forward pass in the backward kernel")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    P = orc.n_params(a.F)
    mu = torch.zeros(a.M, P)
    rho = torch.full((a.M, P), -3.0)
    for m in range(a.M):
        for (wo, bo, k_in, k_out) in orc.param_offsets(a.F):
            mu[m, wo:wo + k_in * k_out] = torch.randn(k_in * k_out, generator=g) * (2.0 / k_in) ** 0.5
            mu[m, bo:bo + k_out] = 0.1 * torch.randn(k_out, generator=g)
    xb = torch.randn(a.M, a.rows, a.F, generator=g).to(dev)
    yb = (torch.randn(a.M, a.rows, generator=g) + (torch.rand(a.M, a.rows, generator=g) > 0.85).float() * 2.0).to(dev)
    res = {}
    for dt in ("fp32", "bf16"):
        tr = EnsembleTrainer(mu.to(dev), rho.to(dev), a.F, a.S, compute_dtype=dt, act_max_bytes=0 if (a.no_save and dt == "bf16") else (int(a.act_gib * 2 ** 30) if a.act_gib else None))
        tr.bf16_row_blocks = a.nb
        kw = dict(seed=7, dropout=(a.mask == "philox"))
        if a.mask == "host":
            gm = torch.Generator().manual_seed(3)
            kw["masks"] = [(torch.rand(a.M * a.S, a.rows, w, generator=gm) >= 0.1).to(torch.uint8) for w in (256, 128, 64)]
        loss = tr.step(xb, yb, 1e-3, 0.9, update=True, **kw).clone()
        torch.cuda.synchronize()
        res[dt] = (loss.cpu(), tr.gmu.cpu().clone(), tr.grho.cpu().clone(), tr._Y.cpu().clone())
        if a.time:
            evs = []
            for it in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tr.step(xb, yb, 1e-3, 0.9, seed=8 + it, dropout=(a.mask != "none"))
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(x.elapsed_time(y) for x, y in evs)[2]
            flop = 3 * 2 * (a.F * 256 + 256 * 128 + 128 * 64 + 64) * a.M * a.S * a.rows
            if dt == "bf16":
                print("row blocks per pair:", tr._bf16_row_blocks(a.rows))
                if a.stamps:
                    st = tr.dW[0, :18].cpu().view(torch.int64).tolist()
                    sf = tr._Y.flatten()[:14].cpu().view(torch.int64).tolist()
                    print("forward phases x-store, L0, L1, L2, L3, act copy (shader-clock cycles):", [sf[i + 1] - sf[i] for i in range(6)], "tile:", sf[6] - sf[0])
                    print("backward phases A..G, barrier (shader-clock cycles):", [st[i + 1] - st[i] for i in range(8)], "tile:", st[8] - st[0])
            print(f"{dt}: {ms:.3f} ms per step  {flop / ms / 1e9:.1f} TFLOP/s (forward + 2x backward, algorithmic)")
    lf, gmf, grf, Yf = res["fp32"]
    lb, gmb, grb, Yb = res["bf16"]
    print("loss fp32", lf.tolist(), "bf16", lb.tolist(), "rel", ((lb - lf).abs() / lf.abs()).tolist())
    print("Y max abs err / max|Y|:", float((Yb - Yf).abs().max() / Yf.abs().max()))
    names = ["W0", "b0", "W1", "b1", "W2", "b2", "w3", "b3"]
    for m in range(a.M):
        i = 0
        for (wo, bo, k_in, k_out) in orc.param_offsets(a.F):
            for nm, lo, hi in ((names[i], wo, wo + k_in * k_out), (names[i + 1], bo, bo + k_out)):
                ref, got = gmf[m, lo:hi], gmb[m, lo:hi]
                print(f"member {m} {nm:3s}: max|ref| {float(ref.abs().max()):.3e}  max err {float((got - ref).abs().max()):.3e}  "
                      f"= {float((got - ref).abs().max() / ref.abs().max()):.3e} of max")
            i += 2
        print(f"member {m} all gmu: {float((gmb[m] - gmf[m]).abs().max() / gmf[m].abs().max()):.3e} of max; "
              f"grho: {float((grb[m] - grf[m]).abs().max() / grf[m].abs().max()):.3e}")


if __name__ == "__main__":
    main()
