#!/usr/bin/env python3
"""Benchmark of the MC predictive path (BASELINE.json metric: member x sample x timestep / s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], SURVEY.md 8d "Config 3"): 10,000 synthetic series x 365
steps (X [3,650,000, 11], standardised, resident in HBM before the timed region), 8 members x
1024 MC samples, bf16 MFMA operands / fp32 accumulate, Philox weight noise, mean + unbiased std
+ 95 % interval from the streaming histogram.  One step = one full predict = 2.99e10 units.
With N > 1 the 8,192 (member, sample) pairs are sharded over the ranks (BASELINE configs[3]); the
partial (sum, sumsq, histogram) are reduce-scattered over rows on RCCL, merged in the library, each
rank finalizes its row slice and the results are all-gathered: total work is fixed -> "strong".

Rank 0 prints ONE JSON line.  `value` is measured with --dtype bf16_fast (bf16 operands, the ELU's
exponential built in the bf16 exponent field: within the north star's 2e-2 of the reference; mean / std
within 6e-3 at config 2, the bound tests/test_gpu_parity.py holds it to); `exact_elu` holds the same step with the exact-ELU bf16 kernel, measured in
the same run.  `roofline` is for the dominant kernel (forward): algorithmic FLOPs (87,680 per unit,
SURVEY.md 8d) / launch duration measured with events on the launch stream; `roofline_reduce` the
same for the streaming reduction in sample bytes (4 B per unit).  `cpu_baseline` times the CPU
oracle (reference predict() op sequence in torch-CPU, oracle/bnn_oracle.py) on the first 100 series
of the SAME X on this host.
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
MFMA_CEILING_MEASURED = 2.1e15  # FLOP/s of a bare v_mfma_f32_32x32x16_bf16 stream at the clock the chip holds (profiles/r01_forward_ablation.md)
# the same stream over operands that toggle like data (random bf16, A fragment from LDS once per two MFMAs as in the kernel),
# 2 s back to back: the clock is power-managed and gives back more when the operand bits move (profiles/r02_mfma_data_ceiling.txt;
# round 3, another device: 1.64e15, and 1.28e15 once the stream carries the kernel's own VALU mix - profiles/r03_mfma_data_ceiling.txt)
MFMA_CEILING_DATA = 1.562e15
MFMA_CEILING_DATA_VALU_MIX = 1.278e15
PEAK_HBM = 8.0e12          # B/s spec
def _newest_profile(stem: str) -> str:
    """profiles/r03_<stem> when this round's PMC digest is committed, else the previous round's."""
    for tag in ("r03", "r02"):
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{tag}_{stem}")):
            return f"{tag}_{stem}"
    return f"r02_{stem}"


PMC_FORWARD, PMC_REDUCE = _newest_profile("pmc_forward.json"), _newest_profile("pmc_reduce.json")
KERNEL_OF = {"bf16": "forward_bf16_exact_kernel", "bf16_fast": "forward_bf16_fast_kernel", "fp32": "forward_f32_kernel"}


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


def cpu_baseline(X_cpu: torch.Tensor, n_series: int, members: int, samples: int):
    """Reference predict() semantics on the host CPU (oracle, 'port') on the first n_series series of the bench's own
    X: 2 warm-ups, best of 5."""
    from oracle import bnn_oracle as orc
    X = X_cpu[:n_series * 365].contiguous()
    nets = make_members(members, X.shape[1], 1, "cpu")
    mu, rho = orc.pack_members(nets)
    g = torch.Generator().manual_seed(0)
    for _ in range(2):
        orc.predict_reference_semantics(X, mu, rho, max(1, samples // 8), g)
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        orc.predict_reference_semantics(X, mu, rho, samples, g)
        best = min(best, time.perf_counter() - t0)
    units = X.shape[0] * members * samples
    return {"value": units / best, "unit": "units/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n_series} series of the bench's X x 365 steps x {members} members x {samples} samples = {units} units, "
                      f"best of 5 after 2 warm-ups, torch-CPU restatement of the reference predict() ({best:.2f} s)"}


def pmc_bytes_per_unit(name: str):
    """HBM bytes per unit from the committed PMC passes (profiles/<name>: FETCH_SIZE x2 + WRITE_SIZE, separate passes,
    MI355X_MICROARCH.md's gfx950 correction), or None when the file is absent.  Counters cannot be read inside this run."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)["hbm_bytes_per_unit"]
    except Exception:
        return None


def loss_block(device):
    """Piece 3 at the config-5 total (100k series x 365 = 3.65e7 elements) through the one-call C entry point."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine, _p, _stream
    eng = Engine(device)
    n = 36_500_000
    g = torch.Generator(device=device).manual_seed(1)
    t = torch.randn(n, device=device, generator=g)
    p = t + 0.3 * torch.randn(n, device=device, generator=g)
    grad, loss = torch.empty_like(p), torch.empty(1, device=device)
    ws = torch.empty(int(eng.lib.mcmc_loss_workspace_bytes()), dtype=torch.uint8, device=device)

    def run():
        _lib.check(eng.lib.mcmc_loss_fused(eng.ctx.handle, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(loss), _p(grad), _p(ws), _stream()), "mcmc_loss_fused")

    run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[5]
    return {"bound": "hbm", "kernel": "fused_stats + fused_forward + fused_backward (mcmc_loss_fused)", "n": n, "ms": ms,
            "achieved": 20 * n / ms / 1e6, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": 20 * n / ms / 1e6 / (PEAK_HBM / 1e9),
            "bytes_model": "20 B/element algorithmic (SURVEY.md 8d); 28 B/element moved by the three dependent passes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--series", type=int, default=10_000)
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16_fast")
    ap.add_argument("--nbins", type=int, default=128)
    ap.add_argument("--chunk-pairs", type=int, default=0)
    ap.add_argument("--chunk-gib", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-balance", action="store_true", help="N > 1: equal pair shards instead of shards by the speed measured in the last warm-up step")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-ELU step and the loss block (profiling runs)")
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

    # one X for every leg: drawn on the host (Philox-keyed, so a device would draw the same series), then uploaded
    X_cpu, _ = make_series_batch(args.series, seed=2025, device="cpu")
    X = X_cpu.to(device)
    members = make_members(args.members, X.shape[1], 1, device)
    mu, rho = pack_members(members, device)
    B = X.shape[0]
    units_per_step = B * args.members * args.samples
    kw = dict(noise="philox", q_levels=(0.025, 0.975), nbins=args.nbins, chunk_pairs=args.chunk_pairs or None,
              chunk_bytes=int(args.chunk_gib * (1 << 30)), group=None if world > 1 else False)

    def kernel_rate(timers, name):
        ev = timers.get(name, [])
        ms = [a.elapsed_time(b) for a, b, _ in ev]
        return sum(ms) / 1e3, sum(u for _, _, u in ev), len(ms)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(dtype, warmup, steps, timers):
        for i in range(warmup):
            tw = {}
            mc_predict(X, mu, rho, args.samples, dtype=dtype, seed=1000 + i, timers=tw, **kw)
            if world > 1 and i == warmup - 1 and not args.no_balance:
                # shard the timed steps by measured device speed: devices sustain different clocks under this kernel, and
                # the step takes as long as the slowest rank (engine.shard_pairs; the statistics do not depend on the split)
                torch.cuda.synchronize()
                secs, units, _ = kernel_rate(tw, "forward")
                rate = torch.tensor([units / secs if secs > 0 else 1.0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
                rates = [torch.zeros_like(rate) for _ in range(world)]
                dist.all_gather(rates, rate)
                kw["shard_weights"] = [float(r.item()) for r in rates]
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            res = mc_predict(X, mu, rho, args.samples, dtype=dtype, seed=i, timers=timers, **kw)
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert torch.isfinite(res.mean).all() and torch.isfinite(res.std).all()
        return elapsed

    timers = {}
    elapsed = timed_steps(args.dtype, args.warmup, args.steps, timers)

    if rank == 0:
        fwd_s, fwd_units, fwd_n = kernel_rate(timers, "forward")
        red_s, red_units, red_n = kernel_rate(timers, "reduce_update")
        achieved = FLOP_PER_UNIT * fwd_units / fwd_s / 1e12 if fwd_s > 0 else 0.0
        fwd_bpu = pmc_bytes_per_unit(PMC_FORWARD)
        red_bpu = pmc_bytes_per_unit(PMC_REDUCE)
        red_gbps = red_units * 4 / red_s / 1e9 if red_s > 0 else 0.0
        phases = {k: sum(a.elapsed_time(b) for a, b, _ in v) / args.steps for k, v in timers.items()}
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
            # the kernel's own name for what it computes in: "bf16_fast" = bf16 MFMA operands, fp32 accumulate, exponent-field ELU
            # (not comparable with an exact-ELU "bf16" line: that one is in `exact_elu`, measured in the same run)
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.series} series x 365 steps x {args.members} members x {args.samples} MC samples, "
                                   f"input_dim {X.shape[1]}, Philox weight noise, mean+std+95% interval ({args.nbins}-bin histogram)",
                       "units_per_step": units_per_step, "pairs_sharded_over": world,
                       "kernel_dtype": args.dtype,
                       "elu": "exponent-field exponential (MCMC_DT_BF16_FAST)" if args.dtype == "bf16_fast" else "exact (v_exp_f32)",
                       "sample_buffer_GiB": args.chunk_gib},
            "roofline": {"bound": "mfma", "kernel": KERNEL_OF.get(args.dtype, args.dtype), "achieved": achieved, "peak": PEAK_BF16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": achieved * 1e12 / PEAK_BF16_DENSE,
                         "traffic": fwd_bpu * fwd_units / max(fwd_n, 1) if fwd_bpu else None,
                         "traffic_source": f"profiles/{PMC_FORWARD} (PMC bytes per unit of a separate rocprofv3 run) x this run's units per launch",
                         "launches": fwd_n, "avg_launch_ms": fwd_s * 1e3 / fwd_n if fwd_n else None, "flop_per_unit": FLOP_PER_UNIT,
                         "measured_bare_mfma_ceiling": MFMA_CEILING_MEASURED / 1e12,
                         "frac_of_measured_ceiling": achieved * 1e12 / MFMA_CEILING_MEASURED,
                         "measured_data_mfma_ceiling": MFMA_CEILING_DATA / 1e12,
                         "frac_of_data_ceiling": achieved * 1e12 / MFMA_CEILING_DATA,
                         # issued MFMA work (92 instructions per 85.6 algorithmic) against the bare stream + this kernel's VALU mix
                         "issued_frac_of_valu_mix_ceiling": achieved * 1e12 * (92 / 85.625) / MFMA_CEILING_DATA_VALU_MIX},
            "roofline_reduce": {"bound": "hbm", "kernel": "reduce_update4_kernel", "achieved": red_gbps, "peak": PEAK_HBM / 1e9,
                                "unit": "GB/s", "frac": red_gbps / (PEAK_HBM / 1e9),
                                "traffic": red_bpu * red_units / max(red_n, 1) if red_bpu else None,
                                "launches": red_n, "avg_launch_ms": red_s * 1e3 / red_n if red_n else None,
                                "bytes_model": "4 B per unit (the fp32 sample); the per-row state is written once per predict"},
            "phases_ms_per_step_rank0": phases,
            "dist": {"world": world, "backend": backend if world > 1 else None,
                     "group_ranks": dist.get_world_size() if world > 1 else 1,
                     "shard_weights": kw.get("shard_weights")},
        }
        if world == 1 and not args.no_extras:
            if args.dtype == "bf16_fast":
                t2 = {}
                e2 = timed_steps("bf16", 1, 1, t2)
                s2, u2, _ = kernel_rate(t2, "forward")
                out["exact_elu"] = {"value": units_per_step / e2, "ms_per_step": e2 * 1e3, "kernel": KERNEL_OF["bf16"],
                                    "roofline_frac": FLOP_PER_UNIT * u2 / s2 / PEAK_BF16_DENSE}
            out["loss"] = loss_block(device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(X_cpu, 100, args.members, 32)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
