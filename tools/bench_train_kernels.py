#!/usr/bin/env python3
"""Large-batch training kernels alone: forward / backward of 80 pairs x 65,536 rows, with the forward pass's activations
recomputed by the backward kernel (mcmc_train_forward / mcmc_train_backward_blocks) or saved and reloaded
(mcmc_train_forward_save / mcmc_train_backward_saved).  Timing and PMC target (tools/pmc_train.sh).

    python tools/bench_train_kernels.py [--rows 65536]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import _lib  # noqa: E402
from mcmc_amd.engine import _p, _stream, param_count  # noqa: E402
from mcmc_amd.train import EnsembleTrainer  # noqa: E402



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--mask-mode", type=int, default=2, help="0: no dropout, 2: Philox masks")
    ap.add_argument("--variants", default="recompute,saved", help="which backward forms to run (PMC passes: one)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    M, S, F, R = 8, 10, 11, args.rows
    P = param_count(F)
    g = torch.Generator().manual_seed(1)
    mu = (torch.randn(M, P, generator=g) * 0.08).to(dev)
    rho = torch.full((M, P), -3.0, device=dev)
    gr = torch.Generator(device=dev).manual_seed(100)
    xb = torch.randn(M, R, F, device=dev, generator=gr)
    yb = torch.randn(M, R, device=dev, generator=gr)
    tr = EnsembleTrainer(mu, rho, F, S)
    tr.step(xb, yb, 1e-3, 0.9, seed=1, update=False)  # fills W / WT and sizes the buffers
    lib = tr.eng.lib
    h = tr.eng.ctx.handle
    nb = int(lib.mcmc_train_row_blocks(R))
    Y = torch.empty(M * S, R, device=dev)
    delta = torch.randn(M, R, device=dev, generator=gr) * 1e-4
    dW = torch.empty(M * S * nb, P, device=dev)
    act = torch.empty(M * S * int(lib.mcmc_train_act_floats(R)), device=dev)
    out = {"mask_mode": args.mask_mode, "rows": R, "row_blocks": nb, "act_GB": act.numel() * 4 / 1e9}

    mm = args.mask_mode

    def forward(a):
        if a is None:
            _lib.check(lib.mcmc_train_forward(h, _p(xb), R, F, M, S, _p(tr.WT), mm, None, None, None, 7, 0.1, _p(Y), _stream()), "forward")
        else:
            _lib.check(lib.mcmc_train_forward_save(h, _p(xb), R, F, M, S, _p(tr.WT), mm, None, None, None, 7, 0.1, _p(Y), _p(a), _stream()), "forward_save")

    def backward(a):
        if a is None:
            _lib.check(lib.mcmc_train_backward_blocks(h, _p(xb), R, F, M, S, _p(tr.W), _p(tr.WT), mm, None, None, None, 7, 0.1, _p(delta), nb,
                                                      _p(dW), _stream()), "backward_blocks")
        else:
            _lib.check(lib.mcmc_train_backward_saved(h, _p(xb), R, F, M, S, _p(tr.W), _p(tr.WT), mm, None, None, None, 7, 0.1, _p(delta), nb,
                                                     _p(dW), _p(a), _stream()), "backward_saved")

    def run(a):
        forward(a)
        backward(a)

    ref = None
    for name, a in [(n, {"recompute": None, "saved": act}[n]) for n in args.variants.split(",")]:
        run(a)
        torch.cuda.synchronize()
        if ref is None:
            ref = dW.clone()
        else:
            out["dW_equal"] = bool(torch.equal(ref, dW))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        forward(a)
        ev[1].record()
        backward(a)
        ev[2].record()
        torch.cuda.synchronize()
        out[name] = {"forward_ms": ev[0].elapsed_time(ev[1]), "backward_ms": ev[1].elapsed_time(ev[2])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
