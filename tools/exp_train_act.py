#!/usr/bin/env python3
"""Experiment: large-batch training step with the forward pass's activations SAVED to HBM and reloaded by the backward kernel
(what the 32-row step does through L2) instead of recomputed.  Calls the library's internal launchers by their mangled
names - a measurement aid, not an interface.

    python tools/exp_train_act.py [--rows 65536]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.engine import _p, _stream, param_count  # noqa: E402
from mcmc_amd.train import EnsembleTrainer  # noqa: E402

FWD = "_ZN4mcmc20launch_train_forwardEPKflliiiS1_iPKhS3_S3_mfPfS4_P12ihipStream_t"
BWD = "_ZN4mcmc21launch_train_backwardEPKflliiiS1_S1_iPKhS3_S3_mfS1_iPfS1_PKNS_13TrainLossFuseEP12ihipStream_t"


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
    fwd, bwd = getattr(lib, FWD), getattr(lib, BWD)
    vp, i64, i32, u64, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_float
    fwd.argtypes = [vp, i64, i64, i32, i32, i32, vp, i32, vp, vp, vp, u64, f32, vp, vp, vp]
    bwd.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp, i32, vp, vp, vp, u64, f32, vp, i32, vp, vp, vp, vp]
    fwd.restype = bwd.restype = i32
    nb = int(lib.mcmc_train_row_blocks(R))
    Y = torch.empty(M * S, R, device=dev)
    delta = torch.randn(M, R, device=dev, generator=gr) * 1e-4
    dW = torch.empty(M * S * nb, P, device=dev)
    act = torch.empty(M * S * int(lib.mcmc_train_act_floats(R)), device=dev)
    out = {"mask_mode": args.mask_mode, "rows": R, "row_blocks": nb, "act_GB": act.numel() * 4 / 1e9}

    def run(a):
        assert fwd(_p(xb), R * F, R, F, M, S, _p(tr.WT), args.mask_mode, None, None, None, 7, 0.1, _p(Y), _p(a), _stream()) == 0
        assert bwd(_p(xb), R * F, R, F, M, S, _p(tr.W), _p(tr.WT), args.mask_mode, None, None, None, 7, 0.1, _p(delta), nb, _p(dW), _p(a), None, _stream()) == 0

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
        assert fwd(_p(xb), R * F, R, F, M, S, _p(tr.WT), args.mask_mode, None, None, None, 7, 0.1, _p(Y), _p(a), _stream()) == 0
        ev[1].record()
        assert bwd(_p(xb), R * F, R, F, M, S, _p(tr.W), _p(tr.WT), args.mask_mode, None, None, None, 7, 0.1, _p(delta), nb, _p(dW), _p(a), None, _stream()) == 0
        ev[2].record()
        torch.cuda.synchronize()
        out[name] = {"forward_ms": ev[0].elapsed_time(ev[1]), "backward_ms": ev[1].elapsed_time(ev[2])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
