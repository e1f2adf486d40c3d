"""Multi-rank rehearsal of the sharded predictive path and the data-parallel loss.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/check_multirank.py

One rank per GPU over RCCL when the node has enough GPUs; on a one-GPU box the ranks share the
card and talk over gloo (rehearsal of the host logic with real kernels).  Every rank compares the
sharded result with its own single-rank run of the same job (Philox noise is keyed by global pair
id, so only the fp64 association and the pilot's histogram edges may differ).
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import engine  # noqa: E402
from mcmc_amd.loss import peak_weighted_loss  # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("MCMC_CHECK_TIMEOUT", "100")), exit=True)  # a hang names its line
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    ngpu = torch.cuda.device_count()
    backend = "nccl" if ngpu >= world else "gloo"
    dev = torch.device("cuda", local % ngpu)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)

    def from_rank0(t):
        """Rank 0's copy of a result, on the host (RCCL moves device tensors, gloo host tensors)."""
        ref = t.detach().clone() if backend == "nccl" else t.detach().cpu()
        dist.broadcast(ref, src=0)
        return ref.cpu()

    solo = None
    for r in range(world):  # every rank must take part in every new_group call
        g = dist.new_group([r])
        if r == rank:
            solo = g
    g = torch.Generator().manual_seed(1)
    B, F, M, S = int(os.environ.get("MCMC_CHECK_ROWS", "1003")), 11, 3, 40
    P = engine.param_count(F)
    X = torch.randn(B, F, generator=g).to(dev)
    mu = (torch.randn(M, P, generator=g) * 0.08).to(dev)
    rho = (-3.0 + 0.3 * torch.randn(M, P, generator=g)).to(dev)
    q = (0.025, 0.5, 0.975)
    for dtype, tol in (("fp32", 2e-6), ("bf16", 2e-6)):
        shard = engine.mc_predict(X, mu, rho, S, dtype=dtype, seed=11, q_levels=q, chunk_pairs=16, group=None)
        alone = engine.mc_predict(X, mu, rho, S, dtype=dtype, seed=11, q_levels=q, chunk_pairs=16, group=solo)
        assert shard.n_samples == alone.n_samples == M * S
        assert torch.allclose(shard.mean, alone.mean, rtol=tol, atol=tol), (dtype, float((shard.mean - alone.mean).abs().max()))
        assert torch.allclose(shard.std, alone.std, rtol=10 * tol, atol=tol), (dtype, float((shard.std - alone.std).abs().max()))
        # the pilot is global pairs q < n_pilot whichever rank owns them, and the histogram merge is exact: the sharded run
        # bins and counts exactly as the single-rank run does
        assert torch.equal(shard.bin_width, alone.bin_width), dtype
        assert torch.equal(shard.quantiles, alone.quantiles), (dtype, float((shard.quantiles - alone.quantiles).abs().max()))
        # all ranks hold the same full result
        assert torch.equal(from_rank0(shard.mean), shard.mean.cpu())
    # row blocks (one pilot all-reduce per block) in the sharded run against one block in the single-rank run
    Xb = torch.randn(40_001, F, generator=g).to(dev)
    shard = engine.mc_predict(Xb, mu, rho, 14, dtype="bf16_fast", seed=3, q_levels=q, chunk_rows=16384, group=None)
    alone = engine.mc_predict(Xb, mu, rho, 14, dtype="bf16_fast", seed=3, q_levels=q, group=solo)
    assert torch.allclose(shard.mean, alone.mean, rtol=2e-6, atol=2e-6) and torch.equal(shard.quantiles, alone.quantiles)
    # shards proportional to per-rank weights (what bench.py does with measured device speeds): same statistics
    skew = engine.mc_predict(Xb, mu, rho, 14, dtype="bf16_fast", seed=3, q_levels=q, group=None, shard_weights=[1.0] + [0.6] * (world - 1))
    assert torch.allclose(skew.mean, alone.mean, rtol=2e-6, atol=2e-6) and torch.equal(skew.quantiles, alone.quantiles)
    # loss, rows split over ranks, global-batch statistics
    N = 4099
    t = torch.randn(N, generator=g)
    p = t + 0.3 * torch.randn(N, generator=g)
    per = (N + world - 1) // world
    sl = slice(rank * per, min(N, (rank + 1) * per))
    p_loc = p[sl].to(dev).requires_grad_(True)
    loss = peak_weighted_loss(p_loc, t[sl].to(dev), group=None)
    loss.backward()
    p_all = p.to(dev).requires_grad_(True)
    full = peak_weighted_loss(p_all, t.to(dev), group=False)
    full.backward()
    assert abs(float(loss) - float(full)) <= 2e-6 * abs(float(full)), (float(loss), float(full))
    assert torch.allclose(p_loc.grad, p_all.grad[sl], rtol=1e-5, atol=1e-9)
    # data-parallel training step (BASELINE config 5): every member's batch rows split over the ranks ==
    # one process on the whole batch (same injected noise and dropout masks)
    from mcmc_amd.train import EnsembleTrainer
    M2, S2, rows, lr = 2, 3, 45, 0.01
    mu0 = (torch.randn(M2, P, generator=g) * 0.08).to(dev)
    rho0 = (-3.0 + 0.3 * torch.randn(M2, P, generator=g)).to(dev)
    xb = torch.randn(M2, rows, F, generator=g).to(dev)
    yb = torch.randn(M2, rows, generator=g).to(dev)
    eps = torch.randn(M2, S2, P, generator=g).to(dev)
    masks = [(torch.rand(M2 * S2, rows, w, generator=g) < 0.9).to(torch.uint8).to(dev) for w in (256, 128, 64)]
    per = (rows + world - 1) // world
    sl = slice(rank * per, min(rows, (rank + 1) * per))
    whole, shard = EnsembleTrainer(mu0, rho0, F, S2), EnsembleTrainer(mu0, rho0, F, S2)
    steps = 3
    for _ in range(steps):
        lw = whole.step(xb, yb, lr, 0.9, noise="injected", eps=eps, masks=masks, group=False)
        ls = shard.step(xb[:, sl], yb[:, sl], lr, 0.9, noise="injected", eps=eps, masks=[m[:, sl] for m in masks], group=None,
                        global_rows=rows)
        assert torch.allclose(lw, ls, rtol=2e-5, atol=1e-7), (lw.tolist(), ls.tolist())
        gscale = float(whole.grad.abs().max())
        assert float((whole.grad - shard.grad).abs().max()) <= 1e-4 * gscale, float((whole.grad - shard.grad).abs().max()) / gscale
        assert torch.allclose(whole.gnorm, shard.gnorm, rtol=1e-5)
    for a, b in ((whole.mu, shard.mu), (whole.rho, shard.rho)):
        diff = (a - b).abs()
        assert float((diff <= 2e-6).float().mean()) >= 0.999 and float(diff.max()) <= 2.1 * lr * steps, (float(diff.max()),)
    assert torch.equal(from_rank0(shard.mu), shard.mu.cpu())  # every rank applied the same update
    dist.barrier()
    if rank == 0:
        print(f"multirank OK: world={world} backend={backend} gpus={ngpu}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
