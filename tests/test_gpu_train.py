"""GPU parity of the training step against oracle.TrainOracle and against the
trajectory the reference's own train() produced (tests/golden/train_steps.npz)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import bnn_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _members(F, M, seed):
    g = torch.Generator().manual_seed(seed)
    P = orc.n_params(F)
    mu = torch.zeros(M, P)
    rho = torch.full((M, P), -3.0)
    for m in range(M):
        for (wo, bo, k_in, k_out) in orc.param_offsets(F):
            mu[m, wo:wo + k_in * k_out] = torch.randn(k_in * k_out, generator=g) * (2.0 / k_in) ** 0.5
            mu[m, bo:bo + k_out] = 0.1 * torch.randn(k_out, generator=g)
        rho[m] += 0.5 * torch.randn(P, generator=g)
    return mu, rho


def _batch(M, rows, F, seed):
    g = torch.Generator().manual_seed(seed)
    xb = torch.randn(M, rows, F, generator=g)
    yb = torch.randn(M, rows, generator=g) + (torch.rand(M, rows, generator=g) > 0.85).float() * 2.0
    return xb, yb


def _masks(M, S, rows, seed, p=0.1):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(M * S, rows, w, generator=g) >= p).to(torch.uint8) for w in orc.HIDDEN]


def _assert_params_close(got, want, lr_sum, what):
    """Adam's update lr*m/(sqrt(v)+eps) is sign-like: where |g| ~ adam_eps (1e-8) an fp32-rounding
    difference in g moves the update by a sizeable part of lr.  Hence: a hard bound of the total
    step budget on every element, and a tight bound on 99.9 % of them.  (The update kernel itself is
    checked tightly below against torch.optim.AdamW fed with the device's own gradients.)"""
    d = (got - want).abs()
    assert float(d.max()) <= 2.0 * lr_sum + 1e-6, f"{what}: max diff {float(d.max()):.3e}"
    assert float(torch.quantile(d.flatten()[:1_000_000], 0.999)) <= 2e-6, what


class _ShadowAdamW:
    """torch.optim.AdamW + clip_grad_norm_ on the host, fed with the DEVICE's gradients."""

    def __init__(self, mu, rho):
        self.p = [torch.nn.Parameter(torch.cat([mu[m], rho[m]]).clone()) for m in range(mu.shape[0])]
        self.opt = [torch.optim.AdamW([q], lr=0.001, weight_decay=0.01) for q in self.p]

    def step(self, gmu, grho, lr, beta1):
        for m, (q, o) in enumerate(zip(self.p, self.opt)):
            o.param_groups[0]["lr"] = lr
            o.param_groups[0]["betas"] = (beta1, 0.999)
            q.grad = torch.cat([gmu[m], grho[m]]).clone()
            torch.nn.utils.clip_grad_norm_([q], 1.0)
            o.step()


# 2500 rows: two row blocks per pair; 4200: four, grid-wide loss kernels; 65,536: the config-5 regime (64 row blocks per pair)
@pytest.mark.parametrize("F,rows,S", [(11, 32, 3), (7, 13, 2), (11, 45, 2), (24, 1, 2), (11, 2500, 2), (7, 4200, 1), (11, 65536, 1)])
def test_train_step_matches_oracle(F, rows, S):
    from mcmc_amd.train import EnsembleTrainer
    M = 2
    mu, rho = _members(F, M, 5 + F)
    tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    oracles = [orc.TrainOracle(mu[m], rho[m]) for m in range(M)]
    shadow = _ShadowAdamW(mu, rho)
    P = orc.n_params(F)
    lrs, b1s = [4e-4, 8e-3, 2.8e-3], [0.95, 0.87, 0.92]
    for it in range(3):
        xb, yb = _batch(M, rows, F, 100 + it)
        eps = torch.randn(M, S, P, generator=torch.Generator().manual_seed(200 + it))
        masks = _masks(M, S, rows, 300 + it)
        loss = tr.step(xb.to(DEV), yb.to(DEV), lrs[it], b1s[it], noise="injected", eps=eps, masks=masks).cpu()
        gmu, grho, gnorm = tr.gmu.cpu(), tr.grho.cpu(), tr.gnorm.cpu()
        shadow.step(gmu, grho, lrs[it], b1s[it])
        for m in range(M):
            got = torch.cat([tr.mu[m], tr.rho[m]]).cpu()
            np.testing.assert_allclose(got.numpy(), shadow.p[m].detach().numpy(), rtol=1e-6, atol=1e-6, err_msg=f"AdamW kernel, step {it}")
            mk = [mm.view(M, S, rows, -1)[m] for mm in masks]
            L, omu, orho, total = oracles[m].step(xb[m], yb[m], eps[m], mk, lr=lrs[it], beta1=b1s[it])
            if rows > 1:  # rows == 1: target std is NaN in the reference as well; loss still finite
                np.testing.assert_allclose(loss[m].item(), L.item(), rtol=2e-5)
            scale = float(omu.abs().max()) + 1e-12
            assert float((gmu[m] - omu).abs().max()) <= 2e-5 * scale + 1e-7, (it, m)
            assert float((grho[m] - orho).abs().max()) <= 2e-5 * max(float(orho.abs().max()), 1e-6) + 1e-7, (it, m)
            np.testing.assert_allclose(gnorm[m].item(), total.item(), rtol=2e-5)
            _assert_params_close(tr.mu[m].cpu(), oracles[m].mu.detach(), sum(lrs[:it + 1]), f"mu step {it} member {m}")
            _assert_params_close(tr.rho[m].cpu(), oracles[m].rho.detach(), sum(lrs[:it + 1]), f"rho step {it} member {m}")
        np.testing.assert_allclose(tr.sigma.cpu().numpy(), np.exp(tr.rho.cpu().numpy()), rtol=2e-6)


@pytest.mark.parametrize("rows", [96, 2500, 8192])
def test_saved_activations_equal_recompute(rows):
    """A step whose backward kernel reloads the forward pass's activations (the default while they fit ACT_MAX_BYTES) ==
    the step whose backward kernel recomputes the forward tile: loss, gradients and parameters, bit for bit, at one
    row block (32-row tiles, the one-call step) and at several (64-row tiles), Philox noise and masks live."""
    from mcmc_amd.train import EnsembleTrainer
    F, M, S = 11, 2, 3
    mu, rho = _members(F, M, 3)
    xb, yb = _batch(M, rows, F, 77)
    out = []
    for max_bytes in (EnsembleTrainer.ACT_MAX_BYTES, 0):
        tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
        tr.act_max_bytes = max_bytes
        losses = [tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, seed=5 + it).clone() for it in range(2)]
        assert tr._save_act == (max_bytes > 0)
        out.append((torch.stack(losses), tr.gmu.clone(), tr.grho.clone(), tr.mu.clone(), tr.rho.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("F,rows,S,mask", [(11, 2500, 2, "host"), (7, 4200, 1, "host"), (24, 2100, 2, "philox"), (11, 65536, 1, "philox")])
def test_bf16_large_batch_step_within_2e2(F, rows, S, mask):
    """Opt-in bf16-operand variant of the large-batch step (mcmc_train_forward_bf16 / mcmc_train_backward_bf16: bf16 MFMA
    operands, fp32 accumulation, fp32 master weights and AdamW) on the same noise and the same dropout masks: loss within 2e-2 of,
    every gradient tensor within 2e-2 of its largest entry of, TrainOracle (autograd on the reference's ops, host masks) - and,
    at the config-5 size with Philox masks (which the oracle's restatement would take minutes to draw), of the fp32 kernels,
    which the oracle pins at that size in test_train_step_matches_oracle."""
    from mcmc_amd.train import EnsembleTrainer
    M = 2
    mu, rho = _members(F, M, 50 + F)
    P = orc.n_params(F)
    xb, yb = _batch(M, rows, F, 400)
    eps = torch.randn(M, S, P, generator=torch.Generator().manual_seed(401))
    kw = dict(noise="injected", eps=eps)
    if mask == "host":
        kw["masks"] = _masks(M, S, rows, 402)
    else:
        kw.update(dropout=True, mask_seed=403)
    tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S, compute_dtype="bf16")
    loss = tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, **kw).cpu()
    gmu, grho = tr.gmu.cpu(), tr.grho.cpu()
    if mask == "host":
        ref = []
        for m in range(M):
            mk = [mm.view(M, S, rows, -1)[m] for mm in kw["masks"]]
            L, omu, orho, _ = orc.TrainOracle(mu[m], rho[m]).step(xb[m], yb[m], eps[m], mk, lr=1e-3, beta1=0.9)
            ref.append((L.detach(), omu, orho))
    else:
        t32 = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
        l32 = t32.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, **kw).cpu()
        ref = [(l32[m], t32.gmu[m].cpu(), t32.grho[m].cpu()) for m in range(M)]
    for m, (L, omu, orho) in enumerate(ref):
        assert abs(float(loss[m]) - float(L)) <= 2e-2 * abs(float(L)), (m, float(loss[m]), float(L))
        for got, want, what in ((gmu[m], omu, "dmu"), (grho[m], orho, "drho")):
            for (wo, bo, k_in, k_out) in orc.param_offsets(F):  # every tensor against its own largest entry
                for lo, hi in ((wo, wo + k_in * k_out), (bo, bo + k_out)):
                    err = float((got[lo:hi] - want[lo:hi]).abs().max())
                    assert err <= 2e-2 * float(want[lo:hi].abs().max()) + 1e-9, (what, m, lo, err)
    # the update ran on the bf16 gradients: parameters moved, and by no more than lr per element (AdamW's first step)
    assert not torch.equal(tr.mu.cpu(), mu) and float((tr.mu.cpu() - mu).abs().max()) <= 1.1e-3
    with pytest.raises(ValueError):
        EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S, compute_dtype="fp16")


@pytest.mark.parametrize("rows,mask", [(2500, "philox"), (4128, "none")])
def test_bf16_saved_activations_equal_recompute(rows, mask):
    """bf16 variant: the backward kernel that reloads the forward kernel's bf16 activation tiles (act) == the one that
    recomputes them (act NULL): the same bf16 operands reach the same MFMAs, so loss, gradients and parameters agree bit for bit."""
    from mcmc_amd.train import EnsembleTrainer
    F, M, S = 11, 2, 3
    mu, rho = _members(F, M, 8)
    xb, yb = _batch(M, rows, F, 78)
    kw = dict(dropout=True, mask_seed=9) if mask == "philox" else dict(dropout=False)
    out = []
    for max_bytes in (None, 0):
        tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S, compute_dtype="bf16", act_max_bytes=max_bytes)
        losses = [tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, seed=5 + it, **kw).clone() for it in range(2)]
        assert (getattr(tr, "_act16", None) is not None) == (max_bytes is None)
        out.append((torch.stack(losses), tr.gmu.clone(), tr.grho.clone(), tr.mu.clone(), tr.rho.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_bf16_step_does_not_depend_on_row_blocks():
    """The bf16 kernels take any number of row blocks per pair (EnsembleTrainer picks it from a makespan model): 1, 3, 7 and 64
    blocks over 2,100 rows - 33 tiles of 64 rows, so with 64 blocks 31 of them are empty and write zero partials - give the same
    loss (same forward tiles: bit for bit) and the same gradients up to the fp32 summation order of the partials."""
    from mcmc_amd.train import EnsembleTrainer
    F, M, S, rows = 11, 2, 2, 2100
    mu, rho = _members(F, M, 21)
    xb, yb = _batch(M, rows, F, 79)
    out = []
    for nb in (1, 3, 7, 64):
        tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S, compute_dtype="bf16")
        tr.bf16_row_blocks = nb
        loss = tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, seed=11, mask_seed=12).clone()
        out.append((loss.cpu(), tr.gmu.cpu().clone(), tr.grho.cpu().clone()))
    for loss, gmu, grho in out[1:]:
        assert torch.equal(loss, out[0][0])
        for got, want in ((gmu, out[0][1]), (grho, out[0][2])):
            assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # the model's own choice is a valid count
    nb = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S, compute_dtype="bf16")._bf16_row_blocks(65536)
    assert 1 <= nb <= 64


def test_train_replays_reference_trajectory(golden):
    """Both golden members stepped together with the eps / masks / lr / beta1 the reference used."""
    from mcmc_amd.train import EnsembleTrainer
    z = golden("train_steps.npz")
    M, S, spm = int(z["M"]), int(z["S"]), int(z["steps_per_member"])
    mu0 = torch.stack([torch.from_numpy(z[f"mu0_{m}"]) for m in range(M)])
    rho0 = torch.stack([torch.from_numpy(z[f"rho0_{m}"]) for m in range(M)])
    F = z["X"].shape[1]
    tr = EnsembleTrainer(mu0.to(DEV), rho0.to(DEV), F, S)
    for j in range(spm):
        idx = [m * spm + j for m in range(M)]
        assert all(z[f"xb_{i}"].shape == z[f"xb_{idx[0]}"].shape for i in idx)
        assert all(z["lr"][i] == z["lr"][idx[0]] for i in idx)
        xb = torch.stack([torch.from_numpy(z[f"xb_{i}"]) for i in idx]).to(DEV)
        yb = torch.stack([torch.from_numpy(z[f"yb_{i}"]) for i in idx]).to(DEV)
        eps = torch.stack([torch.from_numpy(z[f"eps_{i}"]) for i in idx])
        masks = [torch.cat([torch.from_numpy(z[f"mask{l}_{i}"]) for i in idx]) for l in range(3)]
        loss = tr.step(xb, yb, float(z["lr"][idx[0]]), float(z["beta1"][idx[0]]), noise="injected", eps=eps, masks=masks).cpu()
        np.testing.assert_allclose(loss.numpy(), z["loss"][idx], rtol=3e-5)
    lr_sum = float(z["lr"][:spm].sum())
    for m in range(M):
        _assert_params_close(tr.mu[m].cpu(), torch.from_numpy(z[f"mu1_{m}"]), lr_sum, f"final mu member {m}")
        _assert_params_close(tr.rho[m].cpu(), torch.from_numpy(z[f"rho1_{m}"]), lr_sum, f"final rho member {m}")


def test_train_philox_noise_and_masks_match_oracle_streams():
    """Device Philox eps / keep-masks equal the oracle's restatement of the same keying."""
    from mcmc_amd.train import EnsembleTrainer
    F, M, S, rows = 9, 2, 2, 20
    mu, rho = _members(F, M, 77)
    tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    xb, yb = _batch(M, rows, F, 78)
    seed, mseed = 0x1234567890ABCDEF, 987654321
    tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, seed=seed, mask_seed=mseed, update=False)
    P = orc.n_params(F)
    eps_ref = orc.philox_eps(seed, list(range(M * S)), P)
    np.testing.assert_allclose(tr.eps.cpu().numpy(), eps_ref, rtol=0, atol=3e-6)
    Y = tr.predictions().cpu()
    eps_dev = tr.eps.cpu()
    for pair in range(M * S):
        m = pair // S
        mk = [torch.from_numpy(k) for k in orc.philox_keep_masks(mseed, pair, rows)]
        want = orc.forward_pair(xb[m], mu[m], rho[m], eps_dev[pair], mk)
        np.testing.assert_allclose(Y[pair].numpy(), want.numpy(), rtol=2e-5, atol=2e-6)


def test_train_inactive_member_is_frozen():
    from mcmc_amd.train import EnsembleTrainer
    F, M, S, rows = 11, 3, 2, 32
    mu, rho = _members(F, M, 9)
    tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    xb, yb = _batch(M, rows, F, 10)
    active = torch.tensor([1, 0, 1], dtype=torch.uint8, device=DEV)
    tr.step(xb.to(DEV), yb.to(DEV), 1e-3, 0.9, seed=3, active=active)
    assert torch.equal(tr.mu[1].cpu(), mu[1]) and torch.equal(tr.rho[1].cpu(), rho[1])
    assert not torch.equal(tr.mu[0].cpu(), mu[0]) and not torch.equal(tr.mu[2].cpu(), mu[2])
    assert float(tr.adam_v[1].abs().max()) == 0.0


def test_fit_ensemble_learns_and_stops_early():
    from mcmc_amd.train import EnsembleTrainer, fit_ensemble
    from mcmc_amd.data import make_series_batch
    F, M, S = 11, 3, 4
    g = torch.Generator().manual_seed(0)
    X = torch.randn(365, F, generator=g)
    w = torch.randn(F, generator=g)
    y = torch.tanh(X @ w) + 0.05 * torch.randn(365, generator=g)
    mu, rho = _members(F, M, 21)
    tr = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    logs = []
    first = None
    trained = fit_ensemble(tr, X, y, epochs=12, seed=5, log=logs.append, generator=torch.Generator().manual_seed(1))
    assert trained == [12] * M
    # evaluate: deterministic forward (no noise, no dropout) through the same train kernels
    xb = X.to(DEV).expand(M, -1, -1).contiguous()
    yb = y.to(DEV).expand(M, -1).contiguous()
    tr0 = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    l0 = tr0.step(xb, yb, 0.0, 0.9, noise="none", dropout=False, update=False).cpu()
    l1 = tr.step(xb, yb, 0.0, 0.9, noise="none", dropout=False, update=False).cpu()
    assert (l1 < 0.5 * l0).all(), (l0, l1)
    # patience 1 with an impossible-to-improve setting (lr schedule still moves; just exercise the path)
    tr2 = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    trained2 = fit_ensemble(tr2, X, y, epochs=30, seed=5, patience=1, log=logs.append, generator=torch.Generator().manual_seed(1))
    assert all(1 <= t <= 30 for t in trained2) and any("Early stopping" in s for s in logs)
    # the look-ahead loop (device-side early stopping, losses settled epochs later) == the epoch-synchronous loop
    for pat in (1, 3):
        runs = []
        for look in (0, 8):
            t3 = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
            lg = []
            tr_epochs = fit_ensemble(t3, X, y, epochs=30, seed=5, patience=pat, log=lg.append, generator=torch.Generator().manual_seed(1),
                                     lookahead=look)
            runs.append((tr_epochs, lg, t3.t, t3.mu.cpu(), t3.rho.cpu(), t3.adam_v.cpu()))
        assert runs[0][:3] == runs[1][:3], (runs[0][:3], runs[1][:3])
        assert all(torch.equal(a, b) for a, b in zip(runs[0][3:], runs[1][3:]))


def test_train_step_strided_batches_and_epoch_loss_equal_separate_calls():
    """mcmc_train_step on slices of the epoch tensors == the five separate entry points on packed copies, bit for bit;
    loss_acc is the fp64 sum of the per-step fp32 losses (bayesian_ensemble.py:181)."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import _p, _stream
    from mcmc_amd.train import EnsembleTrainer
    F, M, S, N = 11, 3, 4, 100
    mu, rho = _members(F, M, 31)
    g = torch.Generator().manual_seed(32)
    Xe = torch.randn(M, N, F, generator=g).to(DEV)
    ye = torch.randn(M, N, generator=g).to(DEV)
    fused = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    sep = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    lib, h = sep.eng.lib, sep.eng.ctx.handle
    acc = torch.zeros(M, dtype=torch.float64, device=DEV)
    want_acc = torch.zeros(M, dtype=torch.float64)
    for it, (b0, b1) in enumerate([(0, 32), (32, 64), (64, 96), (96, 100)]):  # ragged last batch like drop_last=False
        rows, seed, lr, b1m = b1 - b0, 1000 + it, 1e-3 * (it + 1), 0.9 - 0.01 * it
        loss = fused.step(Xe[:, b0:b1], ye[:, b0:b1], lr, b1m, seed=seed, loss_acc=acc)
        xb, yb = Xe[:, b0:b1].contiguous(), ye[:, b0:b1].contiguous()
        Y, delta = sep._work(rows)
        st = _stream()
        _lib.check(lib.mcmc_train_sample(h, _p(sep.mu), _p(sep.sigma), M, S, F, 2, _p(None), seed, _p(sep.eps), _p(sep.W), _p(sep.WT), st), "s")
        _lib.check(lib.mcmc_train_forward(h, _p(xb), rows, F, M, S, _p(sep.WT), 2, _p(None), _p(None), _p(None), seed, 0.1, _p(Y), st), "f")
        _lib.check(lib.mcmc_train_loss(h, _p(Y), _p(yb), M, S, rows, 2.0, 0.5, _p(delta), _p(sep.loss), st), "l")
        _lib.check(lib.mcmc_train_backward(h, _p(xb), rows, F, M, S, _p(sep.W), _p(sep.WT), 2, _p(None), _p(None), _p(None), seed, 0.1,
                                           _p(delta), _p(sep.dW), st), "b")
        _lib.check(lib.mcmc_train_update(h, _p(sep.mu), _p(sep.rho), _p(sep.sigma), M, S, F, _p(sep.eps), _p(sep.dW), _p(sep.gmu),
                                         _p(sep.grho), _p(sep.adam_m), _p(sep.adam_v), _p(sep.norm_scratch), _p(None), lr, b1m, 0.999,
                                         1e-8, 0.01, 1.0, it + 1, _p(sep.gnorm), st), "u")
        assert torch.equal(loss, sep.loss), it
        want_acc += sep.loss.cpu().double()
    for name in ("mu", "rho", "sigma", "adam_m", "adam_v", "gnorm"):
        assert torch.equal(getattr(fused, name), getattr(sep, name)), name
    assert torch.equal(acc.cpu(), want_acc)
    # a batch the kernels cannot address in place (row stride != F) is packed by the host side: same numbers
    wide = torch.zeros(M, 32, F + 3, device=DEV)
    wide[:, :, :F] = Xe[:, :32]
    a = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    b = EnsembleTrainer(mu.to(DEV), rho.to(DEV), F, S)
    la = a.step(wide[:, :, :F], ye[:, :32], 1e-3, 0.9, seed=5)
    lb = b.step(Xe[:, :32].contiguous(), ye[:, :32].contiguous(), 1e-3, 0.9, seed=5)
    assert torch.equal(la, lb) and torch.equal(a.mu, b.mu)


def test_early_stop_kernel_matches_reference_rule():
    """mcmc_train_early_stop == the bookkeeping of bayesian_ensemble.py:183-193 applied member by member (ties and NaN
    losses count as 'no improvement'; a stopped member is never touched again)."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine, _p, _stream
    eng = Engine(torch.device(DEV))
    lib, h = eng.lib, eng.ctx.handle
    M, epochs, patience = 37, 40, 3
    g = torch.Generator().manual_seed(4)
    losses = torch.rand(epochs, M, generator=g, dtype=torch.float64).mul(4).round().div(4)  # many ties
    losses[5, 3] = float("nan")
    losses[:, 7] = torch.arange(epochs, 0, -1, dtype=torch.float64)                        # always improving: never stops
    best = torch.full((M,), float("inf"), dtype=torch.float64, device=DEV)
    bad = torch.zeros(M, dtype=torch.int64, device=DEV)
    active = torch.ones(M, dtype=torch.uint8, device=DEV)
    w_best, w_bad, w_alive = [float("inf")] * M, [0] * M, [True] * M
    dl = losses.to(DEV)
    for e in range(epochs):
        _lib.check(lib.mcmc_train_early_stop(h, _p(dl[e]), _p(best), _p(bad), _p(active), M, patience, _stream()), "early_stop")
        for m in range(M):
            if not w_alive[m]:
                continue
            if float(losses[e, m]) < w_best[m]:
                w_best[m], w_bad[m] = float(losses[e, m]), 0
            else:
                w_bad[m] += 1
            if w_bad[m] >= patience:
                w_alive[m] = False
        assert active.cpu().tolist() == [int(a) for a in w_alive], e
        assert bad.cpu().tolist() == w_bad and best.cpu().tolist() == w_best, e
    assert w_alive[7] and not all(w_alive)


def test_train_capi_rejects_bad_arguments():
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine
    eng = Engine(torch.device(DEV))
    lib, h = eng.lib, eng.ctx.handle
    buf = torch.zeros(1 << 16, device=DEV)
    p = C.c_void_p(buf.data_ptr())
    z = C.c_void_p(0)
    assert lib.mcmc_train_forward(h, p, 32, 11, 300, 300, p, 0, z, z, z, 0, 0.1, p, z) == -1      # too many pairs
    assert lib.mcmc_train_forward(h, p, 32, 11, 1, 1, p, 1, z, z, z, 0, 0.1, p, z) == -1          # host masks missing
    assert lib.mcmc_train_forward(h, p, 32, 11, 1, 1, p, 2, z, z, z, 0, 1.0, p, z) == -1          # p_drop == 1
    assert lib.mcmc_train_forward(h, p, 0, 11, 1, 1, p, 0, z, z, z, 0, 0.1, p, z) == -1           # rows == 0
    assert lib.mcmc_train_sample(h, p, p, 1, 1, 11, 1, z, 0, p, p, p, z) == -1                    # injected without eps
    assert lib.mcmc_train_update(h, p, p, p, 1, 1, 11, p, p, p, p, p, p, p, z, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 0, z, z) == -1
    assert b"step" in lib.mcmc_last_error()
    # mcmc_train_step: NULL structs, a batch of more than one row block, member strides shorter than a batch
    bufs = _lib.TrainBuffers(*([buf.data_ptr()] * 16))
    hyp = _lib.TrainHyper(0.1, 2.0, 0.5, 0.999, 1e-8, 0.01, 1.0)
    args = lambda b, hy, rows, xs, ys: (h, b, hy, p, xs, p, ys, rows, 11, 1, 1, 2, z, 0, 0, z, z, z, 0, z, 1e-3, 0.9, 1, z, z)
    assert lib.mcmc_train_step(*args(None, C.byref(hyp), 32, 352, 32)) == -1
    assert lib.mcmc_train_step(*args(C.byref(bufs), C.byref(hyp), 4096, 4096 * 11, 4096)) == -1
    assert b"row block" in lib.mcmc_last_error()
    assert lib.mcmc_train_step(*args(C.byref(bufs), C.byref(hyp), 32, 100, 32)) == -1
    assert b"stride" in lib.mcmc_last_error()
    bufs.Y = None
    assert lib.mcmc_train_step(*args(C.byref(bufs), C.byref(hyp), 32, 352, 32)) == -1


def test_ensemble_train_then_predict_end_to_end():
    """GPUBayesianEnsemble.train (fused HIP steps) -> predict, through the reference's API."""
    from mcmc_amd import GPUBayesianEnsemble
    g = torch.Generator().manual_seed(3)
    F = 11
    X = torch.randn(200, F, generator=g)
    y = torch.sin(X[:, 0]) + 0.5 * X[:, 1] + 0.05 * torch.randn(200, generator=g)
    torch.manual_seed(11)
    ens = GPUBayesianEnsemble(input_dim=F, n_models=3, device=torch.device(DEV), seed=17)
    ens.train(X, y, epochs=15, num_samples=4)
    assert len(ens.models) == 3 and ens.epochs_trained == [15, 15, 15]
    mean, std = ens.predict(X, num_samples=50)
    assert mean.names == ("batch",) and std.names == ("batch",)
    mse = float(((mean.rename(None).cpu() - y) ** 2).mean())
    assert mse < 0.25 * float(y.var()), mse          # an untrained ensemble sits near var(y) or above
    assert float(std.rename(None).min()) > 0
    with pytest.raises(ValueError):
        ens.train(X[:, :5], y, epochs=1)


def test_pipeline_bayesian_task_end_to_end(golden, tmp_path):
    """main.py's Bayesian leg on the config-1 series: select -> scale -> train -> predict -> de-scale -> JSON."""
    import json
    from mcmc_amd import pipeline
    z = golden("glue_cfg1.npz")
    feats, temp, target = (torch.from_numpy(z[k]) for k in ("features", "temporal", "target"))
    torch.manual_seed(5)
    res = pipeline.train_and_evaluate_all(feats.refine_names("time", "features"), temp.refine_names("time", "temporal_features"),
                                          target.refine_names("time"), combos=[(), (0, 1)], devices=[torch.device(DEV)],
                                          epochs=40, predict_samples=50, interval=0.95, seed=3, log=lambda *_: None)
    assert set(res["predictions"]) == {"baseline", "0_1"} and len(res["actual"]) == 365
    for name, rec in res["predictions"].items():
        b = rec["bayesian"]
        assert len(b["predictions"]) == len(b["uncertainty"]) == len(b["lower"]) == 365
        assert b["metrics"]["mape"] < 12.0, (name, b["metrics"])      # target ~1000 +- 250: an untrained net is > 20 %
        assert min(b["uncertainty"]) > 0
        assert sum(lo <= hi for lo, hi in zip(b["lower"], b["upper"])) == 365
    pipeline.save_results(res, str(tmp_path / "model_results.json"))
    assert json.loads((tmp_path / "model_results.json").read_text())["predictions"]["0_1"]["bayesian"]["metrics"]["rmse"] > 0


@pytest.mark.gpu
def test_trainer_on_a_device_that_is_not_current():
    """main.py:236-238 puts task i on device i % n while another device may be current: the trainer and the TFT engine
    make their own device current around every C-ABI call (a context refuses a foreign current device)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from mcmc_amd.engine import param_count
    from mcmc_amd.train import EnsembleTrainer
    torch.cuda.set_device(0)
    d1 = torch.device("cuda", 1)
    P = param_count(11)
    tr = EnsembleTrainer(torch.randn(2, P, device=d1) * 0.05, torch.full((2, P), -3.0, device=d1), 11, num_samples=3)
    loss = tr.step(torch.randn(2, 32, 11, device=d1), torch.randn(2, 32, device=d1), 1e-3, seed=1)
    assert torch.isfinite(loss).all() and torch.cuda.current_device() == 0
