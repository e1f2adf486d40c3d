#!/usr/bin/env python3
"""Benchmark of the MC predictive path (BASELINE.json metric: member x sample x timestep / s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], SURVEY.md 8d "Config 3"): 10,000 synthetic series x 365
steps (X [3,650,000, 11], standardised, resident in HBM before the timed region), 8 members x
1024 MC samples, bf16 MFMA operands / fp32 accumulate, Philox weight noise, mean + unbiased std
+ 95 % interval from the streaming histogram.  One step = one full predict = 2.99e10 units.
With N > 1 the 8,192 (member, sample) pairs are sharded over the ranks; the partial (sum, sumsq,
histogram) are reduce-scattered over rows on RCCL, each rank finalizes its row slice and the
results are all-gathered: total work is fixed -> "strong" scaling.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (forward_bf16_tiles_kernel):
algorithmic FLOPs (87,680 per unit) / launch duration measured with events on
the launch stream.  `cpu_baseline` times the CPU oracle (reference predict() op sequence in
torch-CPU, oracle/bnn_oracle.py) on a bounded sample of the same workload on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_UNIT = 2 * (11 * 256 + 256 * 128 + 128 * 64 + 64)  # 87,680 at input_dim 11
PEAK_BF16_DENSE = 2.5e15   # FLOP/s, MI355X_MICROARCH.md chip table (dense, not the 2:1-sparse headline)
PEAK_HBM = 8.0e12          # B/s spec


def make_members(n_members: int, input_dim: int, seed: int, device):
    """Random-init members of the reference architecture (kaiming mu, rho = -3), biases nudged
    off zero so no term vanishes."""
    from mcmc_amd import BayesianNetwork
    torch.manual_seed(seed)
    members = []
    for _ in range(n_members):
        net = BayesianNetwork(input_dim)
        with torch.no_grad():
            for l in list(net.hidden_layers) + [net.output_layer]:
                l.bias_mu.add_(0.1 * torch.randn_like(l.bias_mu))
        members.append(net.to(device))
    return members


def cpu_baseline(n_series: int, members: int, samples: int, seed: int):
    """Reference predict() semantics on the host CPU (oracle, 'port'), bounded sample."""
    from oracle import bnn_oracle as orc
    from mcmc_amd.data import make_series_batch
    X, _ = make_series_batch(n_series, seed=seed, device="cpu")
    nets = make_members(members, 11, 1, "cpu")
    mu, rho = orc.pack_members(nets)
    g = torch.Generator().manual_seed(0)
    orc.predict_reference_semantics(X[:3650], mu, rho, 2, g)  # warm-up
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        orc.predict_reference_semantics(X, mu, rho, samples, g)
        best = min(best, time.perf_counter() - t0)
    units = X.shape[0] * members * samples
    return {"value": units / best, "unit": "units/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_series} series x 365 steps x {members} members x {samples} samples = {units} units, "
                      f"best of 2, torch-CPU restatement of the reference predict() ({best:.2f} s)"}


def load_traffic(units_per_launch: float):
    """HBM bytes per forward launch: bytes/unit from the committed PMC passes (profiles/
    pmc_forward_bf16.json: FETCH_SIZE x2 + WRITE_SIZE, separate passes) x this run's units per
    launch, or None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "pmc_forward_bf16.json")
    try:
        with open(path) as f:
            return json.load(f)["hbm_bytes_per_unit"] * units_per_launch
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--series", type=int, default=10_000)
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--nbins", type=int, default=128)
    ap.add_argument("--chunk-pairs", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if os.environ.get("MCMC_BENCH_HANG_S"):  # debugging aid: a hang dumps every thread's stack and exits
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MCMC_BENCH_HANG_S"]), exit=True)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        args.gpus = world
    # MCMC_BENCH_BACKEND=gloo: rehearsal of the multi-rank host logic on a box with fewer GPUs than ranks
    backend = os.environ.get("MCMC_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # communicator set-up (RCCL connects its rings over xGMI on the first collective of each kind): not part of any
        # step, so it is done here whatever --warmup says
        probe = torch.zeros(world, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if backend == "nccl":
            dist.all_to_all_single(torch.empty_like(probe), probe)
            dist.all_gather_into_tensor(torch.empty(world * world, device=device), probe)
            torch.cuda.synchronize()

    from mcmc_amd import mc_predict, pack_members
    from mcmc_amd.data import make_series_batch

    X, _ = make_series_batch(args.series, seed=2025, device=device)  # identical on every rank
    members = make_members(args.members, X.shape[1], 1, device)
    mu, rho = pack_members(members, device)
    B = X.shape[0]
    units_per_step = B * args.members * args.samples
    kw = dict(dtype=args.dtype, noise="philox", q_levels=(0.025, 0.975), nbins=args.nbins,
              chunk_pairs=args.chunk_pairs or None)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        mc_predict(X, mu, rho, args.samples, seed=1000 + i, **kw)
    sync()
    timers = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = mc_predict(X, mu, rho, args.samples, seed=i, timers=timers, **kw)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(res.mean).all() and torch.isfinite(res.std).all()

    if rank == 0:
        fwd = timers.get("forward", [])
        fwd_ms = [a.elapsed_time(b) for a, b, _ in fwd]
        fwd_units = sum(u for _, _, u in fwd)
        fwd_s = sum(fwd_ms) / 1e3
        achieved = FLOP_PER_UNIT * fwd_units / fwd_s / 1e12 if fwd_s > 0 else 0.0
        red = timers.get("reduce_update", [])
        red_s = sum(a.elapsed_time(b) for a, b, _ in red) / 1e3
        red_units = sum(u for _, _, u in red)
        out = {
            "metric": "MC predictive samples/sec (member x sample x timestep)",
            "value": units_per_step * args.steps / elapsed,
            "unit": "units/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.series} series x 365 steps x {args.members} members x {args.samples} MC samples, "
                                   f"input_dim {X.shape[1]}, Philox weight noise, mean+std+95% interval ({args.nbins}-bin histogram)",
                       "units_per_step": units_per_step, "pairs_sharded_over": world},
            "roofline": {"bound": "mfma", "kernel": "forward_bf16_tiles_kernel", "achieved": achieved, "peak": PEAK_BF16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": achieved * 1e12 / PEAK_BF16_DENSE,
                         "traffic": load_traffic(fwd_units / max(len(fwd_ms), 1)),
                         "launches": len(fwd_ms), "avg_launch_ms": (sum(fwd_ms) / len(fwd_ms)) if fwd_ms else None,
                         "flop_per_unit": FLOP_PER_UNIT},
            "reduce": {"bound": "hbm", "kernel": "reduce_update4_kernel",
                       "achieved_GBps": (red_units * 4 / red_s / 1e9) if red_s > 0 else None, "peak_GBps": PEAK_HBM / 1e9,
                       "note": "sample bytes only (4 B/unit), launches inside the timed steps"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(100, args.members, 64, seed=2025)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
