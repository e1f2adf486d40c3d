"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Run on the MI355X box: pytest -m gpu.

Tolerances (BASELINE.json north_star): fp32 mean/std within 1e-5 relative - measured against
max(|mean|, std) because the standardised mean crosses zero (hard part 4);
bf16 within 2e-2; interval bounds within one histogram bin of torch.quantile.
"""
import numpy as np
import pytest
import torch

from oracle import bnn_oracle as orc

pytestmark = pytest.mark.gpu

Q = (0.025, 0.5, 0.975)


def _dev():
    return torch.device("cuda", 0)


def _rel(a, b, scale):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / scale))


def _check_stats(res, mean_ref, std_ref, tol):
    mean, std = res.mean.cpu().numpy(), res.std.cpu().numpy()
    scale = np.maximum(np.abs(mean_ref), std_ref).astype(np.float64)
    assert _rel(mean, mean_ref, scale) <= tol
    assert _rel(std, std_ref, np.abs(std_ref).astype(np.float64)) <= tol


def _run(g, dtype, S, eps, q_levels=Q, **kw):
    from mcmc_amd import mc_predict
    d = _dev()
    X = torch.from_numpy(g["X"]).to(d)
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    return mc_predict(X, mu, rho, S, dtype=dtype, noise="injected", eps=eps.reshape(-1, eps.shape[-1]).to(d),
                      q_levels=q_levels, return_samples=True, **kw)


@pytest.mark.parametrize("name", ["predict_small.npz", "predict_small_F7.npz", "predict_small_F9.npz"])
def test_fp32_forward_matches_reference_vectors(golden, name):
    g = golden(name)
    eps = torch.from_numpy(g["eps"])
    S = eps.shape[1]
    res = _run(g, "fp32", S, eps, q_levels=())
    ref = g["all_predictions"]
    got = res.samples.cpu().numpy()
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
    _check_stats(res, g["mean"], g["std"], 1e-5)


def test_fp32_config2_mean_std_quantiles(golden):
    g = golden("predict_cfg1.npz")
    M, S = int(g["M"]), int(g["S"])
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, 11)
    res = _run(g, "fp32", S, eps, nbins=128)
    got = res.samples.cpu().numpy()
    assert np.max(np.abs(got[:8] - g["all_predictions_head"])) <= 2e-5
    assert np.max(np.abs(got[-4:] - g["all_predictions_tail"])) <= 2e-5
    _check_stats(res, g["mean"], g["std"], 1e-5)
    width = res.bin_width.cpu().numpy()
    q = res.quantiles.cpu().numpy()
    assert np.all(np.abs(q - g["quantiles"]) <= width[None, :])
    # chunking must not change anything: sums are fp64, histogram integer
    res2 = _run(g, "fp32", S, eps, nbins=128, chunk_pairs=35)
    assert torch.equal(res2.mean, res.mean) or _rel(res2.mean.cpu().numpy(), res.mean.cpu().numpy(), 1.0) < 1e-7
    assert np.all(np.abs(res2.quantiles.cpu().numpy() - g["quantiles"]) <= res2.bin_width.cpu().numpy()[None, :])


@pytest.mark.parametrize("dtype", ["bf16", "bf16_fast"])
def test_bf16_config2_within_2e2(golden, dtype):
    """Both bf16 precisions against the north star's 2e-2 (measured: exact ELU 2.2e-3 / 2.5e-3, exponent-field ELU
    4.3e-3 / 3.3e-3 on mean / std; tools/elu_approx_study.py restates both on the CPU)."""
    g = golden("predict_cfg1.npz")
    M, S = int(g["M"]), int(g["S"])
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, 11)
    res = _run(g, dtype, S, eps, nbins=128)
    _check_stats(res, g["mean"], g["std"], 2e-2)
    if dtype == "bf16_fast":  # the documented accuracy of the exponent-field ELU at this shape
        _check_stats(res, g["mean"], g["std"], 6e-3)
    width = res.bin_width.cpu().numpy()
    allp = orc.predict_all(torch.from_numpy(g["X"]), torch.from_numpy(g["mu"]), torch.from_numpy(g["rho"]), eps)
    got = res.samples.cpu().numpy()
    scale = float(allp.abs().max())
    assert np.max(np.abs(got - allp.numpy())) <= 2e-2 * scale
    # the histogram itself: within ONE bin of torch.quantile over the kernel's own (bf16-perturbed) samples ...
    q = res.quantiles.cpu().numpy()
    own = torch.quantile(res.samples.cpu(), torch.tensor(Q), dim=0).numpy()
    assert _bins_off(q, own, width) <= 1.0
    # ... and against the reference's fp32 quantiles: one bin + what the bf16 samples themselves moved the bounds by
    shift = np.max(np.abs(own - g["quantiles"]), axis=0)
    assert np.all(np.abs(q - g["quantiles"]) <= width[None, :] + shift[None, :])
    assert float(shift.max()) <= 2e-2 * scale


def _bins_off(q, q_ref, width):
    """Largest distance of interval bounds from their reference, in histogram bins of the row."""
    return float(np.max(np.abs(q.astype(np.float64) - q_ref.astype(np.float64)) / width[None, :].astype(np.float64)))


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 2e-2), ("bf16_fast", 2e-2)])
def test_trained_members_predict(golden, dtype, tol):
    """The reference's use is train() then predict() (main.py:169-170): members trained by the REFERENCE (40 epochs on the
    config-1 series; pre-activations up to 27, rho -3.6 .. -2.3) through every kernel on the reference's own noise stream.
    fp32: 1e-5; both bf16 kernels: per-sample outputs, mean and std within the north star's 2e-2.  Interval bounds: within
    one bin of torch.quantile over the kernel's OWN samples (isolates the histogram from the bf16 perturbation), and within
    one bin + the measured shift of the samples against the reference's fp32 quantiles."""
    g = golden("predict_trained.npz")
    M, S = int(g["M"]), int(g["S"])
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, 11)
    res = _run(g, dtype, S, eps, nbins=128)
    got = res.samples.cpu().numpy()
    scale = max(1.0, float(np.abs(g["all_predictions_head"]).max()))
    assert np.max(np.abs(got[:8] - g["all_predictions_head"])) <= max(tol, 2e-5) * scale
    assert np.max(np.abs(got[-4:] - g["all_predictions_tail"])) <= max(tol, 2e-5) * scale
    _check_stats(res, g["mean"], g["std"], tol)
    width = res.bin_width.cpu().numpy()
    q = res.quantiles.cpu().numpy()
    own = torch.quantile(res.samples.cpu(), torch.tensor(Q), dim=0).numpy()
    assert _bins_off(q, own, width) <= 1.0, (dtype, _bins_off(q, own, width))
    if dtype == "fp32":
        assert _bins_off(q, g["quantiles"], width) <= 1.0
    else:
        allp = orc.predict_all(torch.from_numpy(g["X"]), torch.from_numpy(g["mu"]), torch.from_numpy(g["rho"]), eps).numpy()
        assert np.max(np.abs(got - allp)) <= tol * max(1.0, float(np.abs(allp).max()))
        shift = np.max(np.abs(own - g["quantiles"]), axis=0)  # what the bf16 samples themselves moved the bounds by, per row
        assert np.all(np.abs(q - g["quantiles"]) <= width[None, :] + shift[None, :])
        print(f"[{dtype}] trained members: bounds {_bins_off(q, own, width):.2f} bins from torch.quantile of the kernel's samples, "
              f"{_bins_off(q, g['quantiles'], width):.2f} bins from the reference's fp32 quantiles")


def test_train_mode_masks_fp32(golden):
    """identical weights + identical injected dropout masks (train-mode forward)."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine
    g = golden("train_fwd.npz")
    d = _dev()
    eng = Engine(d)
    X = torch.from_numpy(g["X"]).to(d)
    B, F = X.shape
    npass = g["eps"].shape[0]
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    eps = torch.from_numpy(g["eps"]).to(d)  # [npass, P]: M = 1, S = npass -> g = s
    xp = eng.pack_x(X, _lib.DT_F32)
    wp = torch.empty(npass * eng.wpack_bytes(F, _lib.DT_F32), dtype=torch.uint8, device=d)
    eng.sample_weights(mu, rho, npass, F, 0, npass, _lib.NOISE_INJECTED, eps, 0, _lib.DT_F32, wp)
    ldY = 64
    Y = torch.zeros(npass, ldY, device=d)
    masks = [torch.from_numpy(g[f"mask{l}"]).to(d).contiguous() for l in range(3)]
    keep_scale = float(torch.tensor(1.0 / 0.9, dtype=torch.float32))
    eng.forward(_lib.DT_F32, xp, B, F, wp, npass, Y, ldY, masks=masks, keep_scale=keep_scale)
    got = Y[:, :B].cpu().numpy()
    assert np.max(np.abs(got - g["out"])) <= 1e-5 * max(1.0, np.max(np.abs(g["out"])))


def test_philox_noise_matches_oracle():
    from mcmc_amd.engine import Engine
    eng = Engine(_dev())
    seed = 0x1234_5678_9ABC_DEF0
    got = eng.philox_eps(seed, 5, 3, 11).cpu().numpy()
    ref = orc.philox_eps(seed, [5, 6, 7], 44289)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-5  # same integer stream; libm vs ocml log/sin/cos differ by ulps
    assert abs(got.mean()) < 2e-2 and abs(got.std() - 1) < 2e-2


def test_philox_predict_matches_oracle_on_same_noise(golden):
    """MCMC_NOISE_PHILOX end to end: oracle fed with the CPU restatement of the same counters."""
    from mcmc_amd import mc_predict
    g = golden("predict_small.npz")
    d = _dev()
    X = torch.from_numpy(g["X"])
    mu, rho = torch.from_numpy(g["mu"]), torch.from_numpy(g["rho"])
    M, S, seed = mu.shape[0], 5, 99
    res = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype="fp32", noise="philox", seed=seed, return_samples=True)
    eps = torch.from_numpy(orc.philox_eps(seed, range(M * S), mu.shape[1])).reshape(M, S, -1)
    allp = orc.predict_all(X, mu, rho, eps)
    assert np.max(np.abs(res.samples.cpu().numpy() - allp.numpy())) <= 1e-4
    m, s = orc.reduce_mean_std(allp)
    _check_stats(res, m.numpy(), s.numpy(), 1e-4)
    # bf16 with the same key
    resb = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype="bf16", noise="philox", seed=seed)
    _check_stats(resb, m.numpy(), s.numpy(), 2e-2)


def test_noise_none_is_deterministic_mean(golden):
    from mcmc_amd import mc_predict
    g = golden("predict_small.npz")
    d = _dev()
    X, mu, rho = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho"))
    res = mc_predict(X.to(d), mu.to(d), rho.to(d), 1, dtype="fp32", noise="none", return_samples=True)
    ref = torch.stack([orc.forward_pair(X, mu[m], rho[m], None) for m in range(mu.shape[0])])
    assert np.max(np.abs(res.samples.cpu().numpy() - ref.detach().numpy())) <= 1e-5


@pytest.mark.parametrize("B", [1, 31, 33, 129, 300])
def test_ragged_row_counts(golden, B):
    from mcmc_amd import mc_predict
    g = golden("predict_small.npz")
    d = _dev()
    torch.manual_seed(B)
    X = torch.randn(B, 11)
    mu, rho = torch.from_numpy(g["mu"]), torch.from_numpy(g["rho"])
    eps = torch.from_numpy(g["eps"])
    M, S = eps.shape[:2]
    allp = orc.predict_all(X, mu, rho, eps)
    for dtype, tol in (("fp32", 1e-5), ("bf16", 3e-2), ("bf16_fast", 3e-2)):
        res = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype=dtype, noise="injected", eps=eps.reshape(M * S, -1).to(d),
                         return_samples=True)
        assert np.max(np.abs(res.samples.cpu().numpy() - allp.numpy())) <= tol * max(1.0, float(allp.abs().max()))


def test_empty_input_gives_empty_outputs(golden):
    """Zero rows: the reference's predict() runs its loops over an empty batch and returns empty (time,) tensors
    (bayesian_ensemble.py:225-252); so does the HIP path, for every dtype, with and without the interval."""
    from mcmc_amd import mc_predict
    from mcmc_amd.bayesian_ensemble import GPUBayesianEnsemble
    g = golden("predict_small.npz")
    d = _dev()
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    X = torch.empty(0, 11, device=d)
    for dtype in ("fp32", "bf16"):
        for q in ((), (0.025, 0.975)):
            res = mc_predict(X, mu, rho, 4, dtype=dtype, seed=3, q_levels=q)
            assert res.mean.shape == (0,) and res.std.shape == (0,) and res.n_samples == mu.shape[0] * 4
            assert res.quantiles is None or res.quantiles.shape == (len(q), 0)
    from mcmc_amd import BayesianNetwork
    ens = GPUBayesianEnsemble(input_dim=11, n_models=2, device=d)
    ens.models = [BayesianNetwork(11).to(d) for _ in range(2)]
    mean, std = ens.predict(X, num_samples=3)
    assert mean.shape == (0,) and std.shape == (0,) and mean.names == ("batch",)


def test_single_sample_std_is_nan(golden):
    from mcmc_amd import mc_predict
    g = golden("predict_small.npz")
    d = _dev()
    X, mu, rho = (torch.from_numpy(g[k]).to(d) for k in ("X", "mu", "rho"))
    res = mc_predict(X, mu[:1].contiguous(), rho[:1].contiguous(), 1, dtype="fp32", seed=1)
    assert torch.isnan(res.std).all()  # torch.std of one sample (bayesian_ensemble.py:246)
    assert torch.isfinite(res.mean).all()


def test_reduce_kernel_bit_exact_vs_oracle():
    from mcmc_amd.engine import Engine
    d = _dev()
    eng = Engine(d)
    rng = np.random.default_rng(3)
    for (C, B, nbins) in [(37, 70, 64), (256, 200, 128), (100, 65, 256), (64, 1, 512)]:
        Y = (rng.normal(size=(C, B)) * rng.uniform(0.05, 3.0, size=B) + rng.normal(size=B) * 2).astype(np.float32)
        Y[0, 0] = np.float32(1e6)   # overflow bin
        Y[1, 0] = np.float32(-1e6)  # underflow bin
        ldY = (B + 63) // 64 * 64
        Yd = torch.zeros(C, ldY, device=d)
        Yd[:, :B] = torch.from_numpy(Y).to(d)
        pilot = eng.new_state(B)
        npil = min(C, 16)
        eng.reduce_update(Yd, ldY, npil, B, pilot)
        st = eng.new_state(B, nbins)
        eng.set_edges(B, npil, pilot.sum, pilot.sumsq, st, 6.0)
        # two chunks: state carries over
        eng.reduce_update(Yd, ldY, C // 2, B, st)
        eng.reduce_update(Yd[C // 2:], ldY, C - C // 2, B, st)
        s1 = Y[:npil].astype(np.float64).sum(0)
        s2 = (Y[:npil].astype(np.float64) ** 2).sum(0)
        lo, inv_w = orc.hist_edges_from_pilot(npil, s1, s2, nbins, 6.0)
        assert np.array_equal(st.lo.cpu().numpy(), lo) and np.array_equal(st.inv_w.cpu().numpy(), inv_w)
        hist = orc.hist_accumulate(Y, lo, inv_w, nbins)
        got = st.hist.cpu().numpy().view(np.uint16).astype(np.int64)
        assert np.array_equal(got, hist)
        assert np.allclose(st.sum.cpu().numpy(), Y.astype(np.float64).sum(0), rtol=1e-13, atol=1e-9)
        mean, std, quant = eng.finalize(B, st, Q)
        m_ref, s_ref, q_ref = orc.finalize_from_state(C, Y.astype(np.float64).sum(0), (Y.astype(np.float64) ** 2).sum(0), hist, lo, inv_w, Q)
        assert np.allclose(mean.cpu().numpy(), m_ref, rtol=1e-6, atol=1e-6)
        assert np.allclose(std.cpu().numpy(), s_ref, rtol=1e-6)
        assert np.allclose(quant.cpu().numpy(), q_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N", [1, 2, 13, 32, 4096])
def test_loss_forward_backward(golden, N):
    from mcmc_amd import peak_weighted_loss
    g = golden("loss.npz")
    d = _dev()
    p = torch.from_numpy(g[f"p_{N}"]).to(d).requires_grad_(True)
    t = torch.from_numpy(g[f"t_{N}"]).to(d)
    L = peak_weighted_loss(p, t)
    L.backward()
    L64, g64 = float(g[f"loss64_{N}"]), g[f"grad64_{N}"]
    assert abs(float(L) - L64) <= 2e-6 * max(1.0, abs(L64))
    assert np.max(np.abs(p.grad.cpu().numpy() - g64)) <= 2e-6 * max(1e-3, np.max(np.abs(g64)))
    # and against the reference's own fp32 result
    assert abs(float(L) - float(g[f"loss_{N}"])) <= 1e-5 * max(1.0, abs(L64))


def test_loss_alpha_beta_and_large(golden):
    from mcmc_amd import peak_weighted_loss
    g = golden("loss.npz")
    d = _dev()
    p = torch.from_numpy(g["p_ab"]).to(d).requires_grad_(True)
    t = torch.from_numpy(g["t_ab"]).to(d)
    L = peak_weighted_loss(p, t, alpha=3.5, beta=1.0)
    (2.0 * L).backward()
    assert abs(float(L) - float(g["loss_ab"])) <= 1e-5
    assert np.max(np.abs(p.grad.cpu().numpy() - 2.0 * g["grad_ab"])) <= 1e-6
    # large N (config 5 shape per GPU: 4.56e6): closed form in fp64 on the host
    rng = np.random.default_rng(0)
    n = 4_560_001
    tt = rng.normal(size=n).astype(np.float32)
    pp = (tt + 0.3 * rng.normal(size=n)).astype(np.float32)
    pt = torch.from_numpy(pp).to(d).requires_grad_(True)
    Lb = peak_weighted_loss(pt, torch.from_numpy(tt).to(d))
    Lb.backward()
    L64, g64 = orc.peak_weighted_loss_closed_form(pp, tt)
    assert abs(float(Lb) - L64) <= 1e-5 * L64
    assert np.max(np.abs(pt.grad.cpu().numpy() - g64)) <= 1e-5 * np.max(np.abs(g64))


def test_loss_when_only_the_target_requires_grad():
    """y_true gets no gradient (the reference's callers never ask for one: bayesian_ensemble.py:176, temporal_fusion.py:251);
    a target that requires grad must not break backward on the one-call path, with or without a gradient for y_pred."""
    from mcmc_amd import peak_weighted_loss
    d = _dev()
    p = torch.randn(64, device=d)
    t = torch.randn(64, device=d, requires_grad=True)
    L = peak_weighted_loss(p, t)
    L.backward()
    assert t.grad is None and p.grad is None
    p2 = p.clone().requires_grad_(True)
    L2 = peak_weighted_loss(p2, t)
    L2.backward()
    assert float(L2) == float(L) and p2.grad is not None and t.grad is None
    _, g64 = orc.peak_weighted_loss_closed_form(p.cpu().numpy(), t.detach().cpu().numpy())
    assert np.max(np.abs(p2.grad.cpu().numpy() - g64)) <= 2e-6 * max(1e-3, np.max(np.abs(g64)))


def test_ensemble_class_contract(golden):
    """Drop-in boundary: named ('batch',) fp32 tensors on the ensemble's device."""
    from mcmc_amd import BayesianNetwork, GPUBayesianEnsemble
    g = golden("predict_cfg1.npz")
    d = _dev()
    ens = GPUBayesianEnsemble(input_dim=11, device=d, batch_size=32, seed=7)
    torch.manual_seed(0)
    ens.models = [BayesianNetwork(11).to(d) for _ in range(5)]
    X = torch.from_numpy(g["X"]).to(d).refine_names("time", "features")
    mean, std = ens.predict(X)
    assert mean.names == ("batch",) and std.names == ("batch",)
    assert mean.dtype == torch.float32 and mean.shape == (365,) and mean.device.type == "cuda"
    m2, s2 = ens(X)
    assert m2.shape == (365,)
    with pytest.raises(ValueError):
        ens.predict(X.rename(None)[:, :7])
    # statistical agreement with the oracle's own MC estimate (different noise: 500 samples)
    mu, rho = orc.pack_members(ens.models)
    m_ref, s_ref = orc.predict_reference_semantics(torch.from_numpy(g["X"]), mu, rho, 100)
    se = s_ref.numpy() / np.sqrt(500)
    assert np.mean(np.abs(mean.cpu().rename(None).numpy() - m_ref.numpy()) < 6 * se + 1e-4) > 0.99
    mean, std, lo, hi = ens.predict_interval(X, 100)
    assert torch.all(lo.rename(None) <= hi.rename(None))
    # main.py:174-177 consumer idiom
    _ = mean.cpu().rename(None).reshape(-1, 1)
    _ = std.cpu().rename(None) * 2.0
    # opt-in MC dropout (extension): same seed, masks on -> a different, wider predictive distribution
    plain = ens.predict_full(X, 100, seed=11, quantiles=())
    mcd = ens.predict_full(X, 100, seed=11, quantiles=(), mc_dropout=True)
    again = ens.predict_full(X, 100, seed=11, quantiles=(), mc_dropout=True)
    assert torch.equal(mcd.mean, again.mean) and not torch.equal(mcd.mean, plain.mean)
    assert float(mcd.std.mean()) > float(plain.std.mean())


@pytest.mark.parametrize("dtype", ["bf16", "bf16_fast"])
def test_config3_full_size_properties(dtype):
    """BASELINE configs[2] at full row count (10k series x 365 = 3.65 M rows, 8 members, bf16, Philox) with 128 samples per
    member (1,024 samples per row): size-independent properties + a 1,024-row subsample against the fp32 CPU restatement
    fed with the CPU restatement of the same Philox counters: mean / std within 2e-2 for BOTH bf16 kernels; interval bounds
    within one bin of torch.quantile over the same-dtype samples (re-computed by the kernel on the subsample's rows: row
    results do not depend on the launch shape), and within one bin + the measured bf16 shift of the fp32 quantiles."""
    from mcmc_amd import BayesianNetwork, mc_predict, pack_members
    from mcmc_amd.data import make_series_batch
    d = _dev()
    M, S, seed = 8, 128, 424242
    X, _ = make_series_batch(10_000, seed=2025, device=d)
    assert X.shape == (3_650_000, 11)
    torch.manual_seed(1)
    nets = [BayesianNetwork(11) for _ in range(M)]
    with torch.no_grad():
        for n in nets:
            for l in list(n.hidden_layers) + [n.output_layer]:
                l.bias_mu.add_(0.1 * torch.randn_like(l.bias_mu))
    mu, rho = pack_members(nets, d)
    q = (0.025, 0.5, 0.975)
    res = mc_predict(X, mu, rho, S, dtype=dtype, noise="philox", seed=seed, q_levels=q, nbins=128)
    assert res.n_samples == M * S
    assert torch.isfinite(res.mean).all() and torch.isfinite(res.std).all() and (res.std > 0).all()
    # quantiles are sorted and bracket the mean's neighbourhood
    assert (res.quantiles[0] <= res.quantiles[1]).all() and (res.quantiles[1] <= res.quantiles[2]).all()
    # identical rows (same calendar day, same indicators) give identical statistics
    Xc = X.cpu()
    same = (Xc[:365 * 50:365] == Xc[0]).all(dim=1).nonzero().flatten()
    if len(same) > 1:
        i, j = int(same[0]) * 365, int(same[1]) * 365
        assert float(res.mean[i]) == float(res.mean[j]) and float(res.std[i]) == float(res.std[j])
    # row subsample (1,024 rows x 1,024 samples) vs the oracle on the same noise
    idx = torch.randperm(X.shape[0], generator=torch.Generator().manual_seed(0))[:1024]
    eps = torch.from_numpy(orc.philox_eps(seed, range(M * S), mu.shape[1])).reshape(M, S, -1)
    allp = orc.predict_all(Xc[idx], mu.cpu(), rho.cpu(), eps)
    m_ref, s_ref = orc.reduce_mean_std(allp)
    scale = torch.maximum(m_ref.abs(), s_ref)
    e_mean = float(((res.mean.cpu()[idx] - m_ref).abs() / scale).max())
    e_std = float(((res.std.cpu()[idx] - s_ref).abs() / s_ref).max())
    assert e_mean <= 2e-2 and e_std <= 2e-2, (dtype, e_mean, e_std)
    # the same rows alone, samples returned: identical statistics (rows are independent of the launch shape), and the
    # kernel's own samples for the same-dtype quantile check
    sub = mc_predict(X[idx.to(d)].contiguous(), mu, rho, S, dtype=dtype, noise="philox", seed=seed, q_levels=q, nbins=128, return_samples=True)
    assert torch.allclose(sub.mean, res.mean[idx.to(d)], rtol=0, atol=1e-6) and torch.allclose(sub.quantiles, res.quantiles[:, idx.to(d)], rtol=0, atol=1e-5)
    samples = sub.samples.cpu()
    assert float((samples - allp).abs().max()) <= 2e-2 * max(1.0, float(allp.abs().max()))
    width = res.bin_width.cpu()[idx].numpy()
    got_q = res.quantiles.cpu()[:, idx].numpy()
    own = torch.quantile(samples, torch.tensor(q), dim=0).numpy()
    tq = torch.quantile(allp, torch.tensor(q), dim=0).numpy()
    bins_own, bins_ref = _bins_off(got_q, own, width), _bins_off(got_q, tq, width)
    print(f"[{dtype}] config 3, 1,024 rows x 1,024 samples: mean {e_mean:.2e}, std {e_std:.2e}; bounds {bins_own:.2f} bins from "
          f"torch.quantile of the same-dtype samples, {bins_ref:.2f} bins from the fp32 oracle's")
    assert bins_own <= 1.0
    shift = np.max(np.abs(own - tq), axis=0)
    assert np.all(np.abs(got_q - tq) <= width[None, :] + shift[None, :])


def test_config3_chunking_does_not_change_moments():
    """Full row count, few samples: pair chunks (state read-modify-write per chunk) against the default row blocks."""
    from mcmc_amd import BayesianNetwork, mc_predict, pack_members
    from mcmc_amd.data import make_series_batch
    d = _dev()
    M, S, seed = 8, 12, 424242
    X, _ = make_series_batch(10_000, seed=2025, device=d)
    torch.manual_seed(1)
    mu, rho = pack_members([BayesianNetwork(11) for _ in range(M)], d)
    q = (0.025, 0.5, 0.975)
    res = mc_predict(X, mu, rho, S, dtype="bf16_fast", noise="philox", seed=seed, q_levels=q, nbins=128)
    res2 = mc_predict(X, mu, rho, S, dtype="bf16_fast", noise="philox", seed=seed, q_levels=q, nbins=128, chunk_pairs=40)
    assert torch.allclose(res2.mean, res.mean, rtol=0, atol=1e-6)
    assert torch.allclose(res2.std, res.std, rtol=1e-6, atol=1e-7)


def test_histogram_counts_sum_and_determinism(golden):
    """Size-independent integer properties: every row's histogram sums to M*S; two runs with the
    same key are bitwise identical (no atomics on global state)."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine
    g = golden("predict_cfg1.npz")
    d = _dev()
    eng = Engine(d)
    X = torch.from_numpy(g["X"]).to(d)
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    B, F = X.shape
    M, S = mu.shape[0], 40
    outs = []
    for _ in range(2):
        xp = eng.pack_x(X, _lib.DT_BF16)
        wp = torch.empty(M * S * eng.wpack_bytes(F, _lib.DT_BF16), dtype=torch.uint8, device=d)
        eng.sample_weights(mu, rho, S, F, 0, M * S, _lib.NOISE_PHILOX, None, 5, _lib.DT_BF16, wp)
        ldY = 384
        Y = torch.zeros(M * S, ldY, device=d)
        eng.forward(_lib.DT_BF16, xp, B, F, wp, M * S, Y, ldY)
        pilot = eng.new_state(B)
        eng.reduce_update(Y, ldY, 20, B, pilot)
        st = eng.new_state(B, 256)
        eng.set_edges(B, 20, pilot.sum, pilot.sumsq, st, 6.0)
        eng.reduce_update(Y, ldY, M * S, B, st)
        counts = st.hist.cpu().numpy().view(np.uint16).astype(np.int64)
        assert (counts.sum(1) == M * S).all()
        outs.append((Y.cpu(), st.sum.cpu(), st.sumsq.cpu(), counts))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


def test_c_abi_rejects_bad_arguments():
    """Error behaviour at the boundary: negative code + message -> ValueError, nothing launched."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine, _p, _stream
    d = _dev()
    eng = Engine(d)
    Y = torch.zeros(4, 64, device=d)
    s = torch.zeros(64, dtype=torch.float64, device=d)
    with pytest.raises(ValueError, match="nbins"):
        _lib.check(eng.lib.mcmc_reduce_update(eng.ctx.handle, _p(Y), 64, 4, 64, _p(s), _p(s), _p(Y), _p(Y), _p(Y), 100, _stream()), "x")
    with pytest.raises(ValueError, match="ldY"):
        _lib.check(eng.lib.mcmc_forward(eng.ctx.handle, 0, _p(Y), 100, 11, _p(Y), 1, None, None, None, 1.0, _p(Y), 50, _stream()), "x")
    with pytest.raises(ValueError, match="input_dim"):
        _lib.check(eng.lib.mcmc_pack_x(eng.ctx.handle, _p(Y), 4, 99, 0, _p(Y), _stream()), "x")
    with pytest.raises(ValueError, match="pair range"):
        _lib.check(eng.lib.mcmc_sample_weights(eng.ctx.handle, _p(Y), _p(Y), 2, 3, 11, 4, 5, 2, None, 0, 0, _p(Y), _stream()), "x")
    with pytest.raises(ValueError, match="65,535"):
        _lib.check(eng.lib.mcmc_reduce_finalize(eng.ctx.handle, 64, 70000, _p(s), _p(s), _p(Y), _p(Y), _p(Y), 128,
                                                (__import__("ctypes").c_double * 1)(0.5), 1, _p(Y), _p(Y), _p(Y), _stream()), "x")
    # the fast bf16 image has no dropout-mask stage
    with pytest.raises(ValueError, match="MCMC_DT_BF16_FAST"):
        _lib.check(eng.lib.mcmc_forward_dropout(eng.ctx.handle, 2, _p(Y), 64, 11, _p(Y), 2, 3, 0, 1, 0, 0.1, 0, _p(Y), 64, _stream()), "x")
    # empty inputs are accepted and do nothing
    assert eng.lib.mcmc_forward(eng.ctx.handle, 0, None, 0, 11, None, 0, None, None, None, 1.0, None, 0, _stream()) == 0


@pytest.mark.parametrize("F,B", [(1, 33), (12, 100), (15, 64), (16, 97), (17, 200), (24, 513)])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 2e-2), ("bf16_fast", 2e-2)])
def test_input_widths_across_the_k_granule(F, B, dtype, tol):
    """Every input width class: F+1 (bias column) below / at / above the 16-wide bf16 K granule and the
    kernel's maximum (24), against the oracle on the same injected noise: every per-pair output within the north star's
    tolerance (2e-2 of max|y| for both bf16 kernels), and the mean / std over 2 x 64 = 128 samples within the same 2e-2
    (the statistic of a handful of samples has the relative error of single outputs, so it is taken over 128)."""
    from mcmc_amd import mc_predict
    g = torch.Generator().manual_seed(1000 + F)
    M, S = 2, 64
    P = orc.n_params(F)
    X = torch.randn(B, F, generator=g)
    mu = torch.zeros(M, P)
    rho = torch.full((M, P), -3.0) + 0.3 * torch.randn(M, P, generator=g)
    for m in range(M):
        for (wo, bo, k_in, k_out) in orc.param_offsets(F):
            mu[m, wo:wo + k_in * k_out] = torch.randn(k_in * k_out, generator=g) * (2.0 / k_in) ** 0.5
            mu[m, bo:bo + k_out] = 0.1 * torch.randn(k_out, generator=g)
    eps = torch.randn(M, S, P, generator=g)
    ref = orc.predict_all(X, mu, rho, eps).numpy()
    d = _dev()
    res = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype=dtype, noise="injected", eps=eps.reshape(-1, P).to(d), return_samples=True)
    got = res.samples.cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.max(np.abs(got - ref)) <= tol * scale, (F, B, dtype, float(np.max(np.abs(got - ref))), scale)
    m_ref, s_ref = orc.reduce_mean_std(torch.from_numpy(ref))
    _check_stats(res, m_ref.numpy(), s_ref.numpy(), tol)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_fast"])
def test_predict_plan_graph_replay_equals_eager(golden, dtype):
    """The captured hipGraph of a whole predict gives what mc_predict gives for the same seed, replay after replay."""
    from mcmc_amd import mc_predict
    from mcmc_amd.engine import PredictPlan
    g = golden("predict_cfg1.npz")
    d = _dev()
    X = torch.from_numpy(g["X"]).to(d)
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    M, S = mu.shape[0], 40
    plan = PredictPlan(X.shape[0], X.shape[1], M, S, device=d, dtype=dtype, q_levels=Q)
    for seed in (7, 0xFEDCBA9876543210, 7):
        got = plan(X, mu, rho, seed)
        want = mc_predict(X, mu, rho, S, dtype=dtype, seed=seed, q_levels=Q)
        assert got.n_samples == want.n_samples == M * S
        assert torch.equal(got.mean, want.mean) and torch.equal(got.std, want.std) and torch.equal(got.quantiles, want.quantiles)
    other = plan(X, mu, rho, 8).mean.clone()
    assert not torch.equal(other, plan(X, mu, rho, 7).mean)
    # new inputs through the same graph
    X2 = X.flip(0).contiguous()
    assert torch.equal(plan(X2, mu, rho, 3).mean, mc_predict(X2, mu, rho, S, dtype=dtype, seed=3, q_levels=Q).mean)
    with pytest.raises(ValueError):
        plan(X[:100], mu, rho, 1)


def _philox_masks(seed, pair_ids, rows):
    """Keep-masks [pairs, rows, 256/128/64] from the oracle's restatement of the 'MASK' Philox stream."""
    per = [orc.philox_keep_masks(seed, int(g), rows) for g in pair_ids]
    return [np.stack([p[l] for p in per]) for l in range(3)]


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 2e-2)])
def test_mc_dropout_philox_masks_match_oracle(golden, dtype, tol):
    """Counter-based dropout masks inside the MFMA forward (north-star piece 1): the samples equal the oracle's
    train-mode forward fed with the oracle's restatement of the same Philox mask stream; chunking does not matter."""
    from mcmc_amd import mc_predict
    g = golden("predict_small.npz")
    d = _dev()
    X, mu, rho, eps = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho", "eps"))
    M, S, P = eps.shape
    B = X.shape[0]
    seed = 0xC0FFEE1234
    kw = dict(dtype=dtype, noise="injected", eps=eps.reshape(M * S, P).to(d), return_samples=True, mc_dropout=True, mask_seed=seed)
    res = mc_predict(X.to(d), mu.to(d), rho.to(d), S, **kw)
    masks = _philox_masks(seed, range(M * S), B)
    ref = torch.stack([orc.forward_pair(X, mu[p // S], rho[p // S], eps[p // S, p % S],
                                        masks=[torch.from_numpy(m[p]) for m in masks]) for p in range(M * S)])
    err = float((res.samples.cpu() - ref).abs().max())
    assert err <= tol * max(1.0, float(ref.abs().max())), (dtype, err)
    plain = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype=dtype, noise="injected", eps=eps.reshape(M * S, P).to(d), return_samples=True)
    assert float((plain.samples - res.samples).abs().max()) > 1e-3  # the masks did something
    chunked = mc_predict(X.to(d), mu.to(d), rho.to(d), S, chunk_pairs=5, **kw)
    assert torch.equal(chunked.samples, res.samples)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mc_dropout_independent_of_row_blocks(dtype):
    """Philox dropout masks are keyed by the GLOBAL row (mcmc_forward_dropout's row_begin): an MC-dropout predict cut into
    row blocks - and into row blocks x pair chunks - gives bit for bit the samples of one launch over all rows."""
    from mcmc_amd import mc_predict
    d = _dev()
    g = torch.Generator().manual_seed(11)
    B, F, M, S = 40_001, 11, 2, 3
    P = orc.n_params(F)
    X = torch.randn(B, F, generator=g).to(d)
    mu = (torch.randn(M, P, generator=g) * 0.08).to(d)
    rho = (-3.0 + 0.3 * torch.randn(M, P, generator=g)).to(d)
    kw = dict(dtype=dtype, seed=5, return_samples=True, mc_dropout=True, mask_seed=99)
    whole = mc_predict(X, mu, rho, S, **kw)
    blocks = mc_predict(X, mu, rho, S, chunk_rows=16384, **kw)
    both = mc_predict(X, mu, rho, S, chunk_rows=16384, chunk_pairs=4, **kw)
    assert torch.equal(blocks.samples, whole.samples) and torch.equal(both.samples, whole.samples)
    plain = mc_predict(X, mu, rho, S, dtype=dtype, seed=5, return_samples=True)
    assert float((plain.samples - whole.samples).abs().max()) > 1e-3  # the masks did something
    # a row deep inside the third block against the oracle fed the oracle's restatement of the same mask stream
    row = 16384 * 2 + 4321
    eps = torch.from_numpy(orc.philox_eps(5, range(M * S), P)).reshape(M, S, P)
    for p in (0, M * S - 1):
        masks = [torch.from_numpy(m) for m in orc.philox_keep_masks(99, p, 1, row_begin=row)]
        ref = orc.forward_pair(X[row:row + 1].cpu(), mu.cpu()[p // S], rho.cpu()[p // S], eps[p // S, p % S], masks=masks)
        tol = 1e-4 if dtype == "fp32" else 2e-2
        assert abs(float(whole.samples[p, row]) - float(ref[0])) <= tol * max(1.0, abs(float(ref[0])))


def test_bf16_host_masks_equal_philox_masks(golden):
    """bf16 MFMA forward: masks supplied from the host give bit-for-bit what the in-kernel Philox masks give
    when the host masks are that same stream (and match the fp32 oracle within the bf16 bar)."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine
    g = golden("train_fwd.npz")
    d = _dev()
    eng = Engine(d)
    X = torch.from_numpy(g["X"]).to(d)
    B, F = X.shape
    npass = g["eps"].shape[0]
    mu, rho = torch.from_numpy(g["mu"]).to(d), torch.from_numpy(g["rho"]).to(d)
    eps = torch.from_numpy(g["eps"]).to(d)  # M = 1, S = npass
    seed, ldY = 77, 64
    for dt in (_lib.DT_BF16, _lib.DT_F32):
        xp = eng.pack_x(X, dt)
        wp = torch.empty(npass * eng.wpack_bytes(F, dt), dtype=torch.uint8, device=d)
        eng.sample_weights(mu, rho, npass, F, 0, npass, _lib.NOISE_INJECTED, eps, 0, dt, wp)
        Yp, Yh = torch.zeros(npass, ldY, device=d), torch.zeros(npass, ldY, device=d)
        eng.forward(dt, xp, B, F, wp, npass, Yp, ldY, philox_dropout=(seed, 0.1, 1, npass, 0))
        hm = [torch.from_numpy(m).to(d).contiguous() for m in _philox_masks(seed, range(npass), B)]
        eng.forward(dt, xp, B, F, wp, npass, Yh, ldY, masks=hm, keep_scale=float(torch.tensor(1.0 / 0.9, dtype=torch.float32)))
        assert torch.equal(Yp, Yh)
    # the reference's own recorded masks through the bf16 kernel
    masks = [torch.from_numpy(g[f"mask{l}"]).to(d).contiguous() for l in range(3)]
    xp = eng.pack_x(X, _lib.DT_BF16)
    wp = torch.empty(npass * eng.wpack_bytes(F, _lib.DT_BF16), dtype=torch.uint8, device=d)
    eng.sample_weights(mu, rho, npass, F, 0, npass, _lib.NOISE_INJECTED, eps, 0, _lib.DT_BF16, wp)
    Y = torch.zeros(npass, ldY, device=d)
    eng.forward(_lib.DT_BF16, xp, B, F, wp, npass, Y, ldY, masks=masks, keep_scale=float(torch.tensor(1.0 / 0.9, dtype=torch.float32)))
    got = Y[:, :B].cpu().numpy()
    assert np.max(np.abs(got - g["out"])) <= 2e-2 * max(1.0, np.max(np.abs(g["out"])))


@pytest.mark.parametrize("W,rows,nbins", [(8, 457, 128), (2, 64, 64), (3, 1, 512)])
def test_reduce_merge_bit_exact(W, rows, nbins):
    """mcmc_reduce_merge (the local half of the multi-GPU reduce-scatter) against numpy: fp64 sums in rank order bit for
    bit, uint16 counts exact; a count that passes 65,535 saturates and raises instead of carrying into its neighbour."""
    from mcmc_amd.engine import Engine
    rng = np.random.default_rng(W * 1000 + rows)
    eng = Engine(_dev())
    s1 = rng.normal(size=(W, rows)) * 1e3
    s2 = rng.random(size=(W, rows)) * 1e6
    hist = rng.integers(0, 65535 // W + 1, size=(W, rows, nbins)).astype(np.uint16)
    hist[:, 0, 0] = 65535 // W  # the largest legal total
    d = _dev()
    a, b, h = eng.reduce_merge(torch.from_numpy(s1).to(d), torch.from_numpy(s2).to(d), torch.from_numpy(hist.view(np.int16)).to(d), nbins)
    ra, rb = s1[0].copy(), s2[0].copy()
    for p in range(1, W):
        ra += s1[p]
        rb += s2[p]
    assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(b.cpu().numpy(), rb)
    assert np.array_equal(h.cpu().numpy().view(np.uint16).astype(np.int64), hist.astype(np.int64).sum(0))
    a2, b2, none = eng.reduce_merge(torch.from_numpy(s1).to(d), torch.from_numpy(s2).to(d), None, 0)
    assert none is None and np.array_equal(a2.cpu().numpy(), ra)
    if W > 1:
        hist[:, rows - 1, nbins - 1] = 40000
        with pytest.raises(ValueError, match="65,535"):
            eng.reduce_merge(torch.from_numpy(s1).to(d), torch.from_numpy(s2).to(d), torch.from_numpy(hist.view(np.int16)).to(d), nbins)


def test_device_generator_equals_cpu():
    """Row f3: the batched generator draws from Philox in integer torch ops, so a HIP device and the CPU give the same
    series for a seed: identical promotion / weather indicator columns, X and y equal to float rounding."""
    from mcmc_amd.data import make_series_batch
    Xc, yc = make_series_batch(300, seed=77, standardize=False)
    Xg, yg = make_series_batch(300, seed=77, standardize=False, device=_dev())
    assert torch.equal(Xg.cpu()[:, :5], Xc[:, :5])          # indicator columns: integer draws only
    assert torch.allclose(Xg.cpu()[:, 5:], Xc[:, 5:], rtol=0, atol=1e-6)   # calendar columns
    assert torch.allclose(yg.cpu(), yc, rtol=1e-5, atol=1e-3)
    Xc, yc = make_series_batch(300, seed=77)
    Xg, yg = make_series_batch(300, seed=77, device=_dev())
    assert torch.allclose(Xg.cpu(), Xc, rtol=0, atol=2e-6)
    assert torch.allclose(yg.cpu(), yc, rtol=0, atol=1e-4), float((yg.cpu() - yc).abs().max())  # fp32 sums on two devices


@pytest.mark.parametrize("dtype", ["fp32", "bf16_fast"])
def test_row_blocks_equal_whole_rows(dtype):
    """The default chunking of big jobs - row blocks of whole 16,384-row superblocks that hold every pair, so each row's state
    is written once - against one block of all rows and against pair chunks: the same samples reach the same rows, so the
    statistics are identical (moments: fp64 sums of the same values in another association; histogram: exact)."""
    from mcmc_amd import mc_predict
    from mcmc_amd.engine import plan_chunks
    assert plan_chunks(8192, 3_650_000, 8 << 30, None, None, 8) == (8192, 262144)   # config 3, one GPU: 14 blocks of 8 GiB
    assert plan_chunks(1024, 3_650_000, 8 << 30, None, None, 8) == (1024, 1835008)  # config 4, one of 8 ranks: 2 blocks
    d = _dev()
    g = torch.Generator().manual_seed(3)
    B, F, M, S = 40_001, 11, 3, 14
    P = orc.n_params(F)
    X = torch.randn(B, F, generator=g).to(d)
    mu = (torch.randn(M, P, generator=g) * 0.08).to(d)
    rho = (-3.0 + 0.3 * torch.randn(M, P, generator=g)).to(d)
    kw = dict(dtype=dtype, seed=5, q_levels=Q, nbins=128, return_samples=True)
    whole = mc_predict(X, mu, rho, S, **kw)
    blocks = mc_predict(X, mu, rho, S, chunk_rows=16384, **kw)                    # 3 row blocks, the last one ragged
    both = mc_predict(X, mu, rho, S, chunk_rows=16384, chunk_pairs=12, **kw)      # row blocks x pair chunks (state read-modify-write)
    for other in (blocks, both):
        assert torch.equal(other.samples, whole.samples)
        assert torch.allclose(other.mean, whole.mean, rtol=0, atol=1e-6) and torch.allclose(other.std, whole.std, rtol=1e-6, atol=1e-7)
    assert torch.equal(blocks.quantiles, whole.quantiles) and torch.equal(blocks.bin_width, whole.bin_width)


@pytest.mark.parametrize("scale_x", [30.0, 1000.0])
def test_bf16_fast_extreme_preactivations(scale_x):
    """The exponent-field ELU far outside the trained range: inputs scaled so that pre-activations reach +-10^2 .. 10^4
    (deeply saturated negative branch: the constructed bits clamp at 0; large positive: the linear branch wins the integer
    max).  Outputs stay finite and follow the fp32 oracle like the exact bf16 kernel does (bf16 relative error of large
    activations dominates both)."""
    from mcmc_amd import mc_predict
    g = torch.Generator().manual_seed(9)
    F, B, M, S = 11, 257, 2, 4
    P = orc.n_params(F)
    X = torch.randn(B, F, generator=g) * scale_x
    mu = torch.zeros(M, P)
    for m in range(M):
        for (wo, bo, k_in, k_out) in orc.param_offsets(F):
            mu[m, wo:wo + k_in * k_out] = torch.randn(k_in * k_out, generator=g) * (2.0 / k_in) ** 0.5
            mu[m, bo:bo + k_out] = 0.1 * torch.randn(k_out, generator=g)
    rho = torch.full((M, P), -3.0)
    eps = torch.randn(M, S, P, generator=g)
    ref = orc.predict_all(X, mu, rho, eps).numpy()
    d = _dev()
    errs = {}
    for dtype in ("bf16", "bf16_fast"):
        res = mc_predict(X.to(d), mu.to(d), rho.to(d), S, dtype=dtype, noise="injected", eps=eps.reshape(-1, P).to(d), return_samples=True)
        got = res.samples.cpu().numpy()
        assert np.isfinite(got).all(), dtype
        errs[dtype] = float(np.max(np.abs(got - ref)) / np.abs(ref).max())
    assert errs["bf16"] <= 2e-2 and errs["bf16_fast"] <= 2e-2, errs
