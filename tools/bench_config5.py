#!/usr/bin/env python3
"""Large-batch (config 5) training step: rows of every member's batch split over the ranks, global-batch
peak-weighted loss, gradient all-reduce, clipped AdamW (EnsembleTrainer.step(group=None)).

    python tools/bench_config5.py --rows 262144                  # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_config5.py --rows 4562500

--rows is per rank.  Prints one JSON line: ms per step, pair-rows/s (member x sample x row), algorithmic
TFLOP/s at 3 x 87,680 FLOP per pair-row (forward + the two backward GEMM families; the backward kernel's
recompute of the forward is not counted)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.engine import param_count  # noqa: E402
from mcmc_amd.train import EnsembleTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262_144)
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--features", type=int, default=11)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="fp32", choices=("fp32", "bf16"), help="GEMM operand type of the large-batch kernels (EnsembleTrainer compute_dtype)")
    args = ap.parse_args()
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ngpu = torch.cuda.device_count()
    dev = torch.device("cuda", local % ngpu)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl" if ngpu >= world else "gloo")
    M, S, F, R = args.members, args.samples, args.features, args.rows
    P = param_count(F)
    g = torch.Generator().manual_seed(1)
    mu = (torch.randn(M, P, generator=g) * 0.08).to(dev)
    rho = torch.full((M, P), -3.0, device=dev)
    gr = torch.Generator(device=dev).manual_seed(100 + rank)
    xb = torch.randn(M, R, F, device=dev, generator=gr)
    yb = torch.randn(M, R, device=dev, generator=gr)
    tr = EnsembleTrainer(mu, rho, F, S, compute_dtype=args.dtype)
    group = None if world > 1 else False
    kw = dict(group=group, global_rows=R * world, mask_seed=7 + rank)
    tr.step(xb, yb, 1e-3, 0.9, seed=1, **kw)  # warm-up
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = tr.step(xb, yb, 1e-3, 0.9, seed=2 + it, **kw)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    assert torch.isfinite(loss).all()
    if rank == 0:
        pair_rows = M * S * R * world
        bf16 = args.dtype == "bf16" and int(tr.eng.lib.mcmc_train_row_blocks(R)) > 1
        print(json.dumps({"world": world, "dtype": args.dtype, "rows_per_rank": R, "members": M, "samples": S, "ms_per_step": dt * 1e3,
                          "pair_rows_per_s": pair_rows / dt, "algorithmic_tflops": 3 * 87680 * pair_rows / dt / 1e12,
                          "row_blocks": tr._bf16_row_blocks(R) if bf16 else int(tr.eng.lib.mcmc_train_row_blocks(R)),
                          "saved_activations": (getattr(tr, "_act16", None) is not None) if bf16 else bool(tr._save_act), "loss": [round(float(x), 5) for x in loss.cpu()]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
