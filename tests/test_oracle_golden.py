"""Pin the CPU oracle against vectors captured from the reference itself (oracle/gen_golden.py).

CPU-only.  Bit-exact where the oracle replays the reference's torch ops; fp64 closed form of
the loss gradient within 1e-12 of the reference's fp64 autograd."""
import numpy as np
import pytest
import torch

from oracle import bnn_oracle as orc


def test_layout_counts():
    # 88,578 params = 2 x 44,289 effective weights at F=11
    assert orc.n_params(11) == 44289
    offs = orc.param_offsets(11)
    assert offs[0] == (0, 2816, 11, 256)
    assert offs[-1][2:] == (64, 1)


@pytest.mark.parametrize("name", ["predict_small.npz", "predict_small_F7.npz", "predict_small_F9.npz"])
def test_predict_small_bit_exact(golden, name):
    g = golden(name)
    X, mu, rho, eps = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho", "eps"))
    allp = orc.predict_all(X, mu, rho, eps)
    assert np.array_equal(allp.numpy(), g["all_predictions"])
    mean, std = orc.reduce_mean_std(allp)
    assert np.array_equal(mean.numpy(), g["mean"])
    assert np.array_equal(std.numpy(), g["std"])


def test_predict_cfg1_regenerated_eps_bit_exact(golden):
    g = golden("predict_cfg1.npz")
    M, S = int(g["M"]), int(g["S"])
    X, mu, rho = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho"))
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, X.shape[1])
    allp = orc.predict_all(X, mu, rho, eps)
    assert np.array_equal(allp[:8].numpy(), g["all_predictions_head"])
    assert np.array_equal(allp[-4:].numpy(), g["all_predictions_tail"])
    mean, std = orc.reduce_mean_std(allp)
    assert np.array_equal(mean.numpy(), g["mean"])
    assert np.array_equal(std.numpy(), g["std"])
    q = torch.quantile(allp, torch.from_numpy(g["q_levels"]).float(), dim=0)
    assert np.array_equal(q.numpy(), g["quantiles"])
    # main.py:174-177 de-scaling
    assert np.allclose(mean.numpy().astype(np.float64) * float(g["y_scale"]) + float(g["y_mean"]), g["descaled_mean"])


def test_predict_trained_members_bit_exact(golden):
    """main.py:169-170 on the reference itself: members TRAINED by the reference's train() (40 epochs, config-1 series),
    then its predict() - the oracle must land on it bit for bit, like on the init-time members."""
    g = golden("predict_trained.npz")
    M, S = int(g["M"]), int(g["S"])
    X, mu, rho = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho"))
    assert float(np.abs(g["rho"] + 3.0).max()) > 0.3 and float(g["preact_absmax"].max()) > 15.0  # not init-time weights
    eps = orc.reference_rng_eps(int(g["eps_seed"]), M, S, X.shape[1])
    allp = orc.predict_all(X, mu, rho, eps)
    assert np.array_equal(allp[:8].numpy(), g["all_predictions_head"])
    assert np.array_equal(allp[-4:].numpy(), g["all_predictions_tail"])
    mean, std = orc.reduce_mean_std(allp)
    assert np.array_equal(mean.numpy(), g["mean"])
    assert np.array_equal(std.numpy(), g["std"])
    q = torch.quantile(allp, torch.from_numpy(g["q_levels"]).float(), dim=0)
    assert np.array_equal(q.numpy(), g["quantiles"])


def test_train_mode_masks_bit_exact(golden):
    g = golden("train_fwd.npz")
    X, mu, rho, eps = (torch.from_numpy(g[k]) for k in ("X", "mu", "rho", "eps"))
    for i in range(eps.shape[0]):
        masks = [torch.from_numpy(g[f"mask{l}"][i]) for l in range(3)]
        out = orc.forward_pair(X, mu[0], rho[0], eps[i], masks=masks)
        assert np.array_equal(out.numpy(), g["out"][i])


@pytest.mark.parametrize("N", [1, 2, 13, 32, 4096])
def test_loss_matches_reference(golden, N):
    g = golden("loss.npz")
    p = torch.from_numpy(g[f"p_{N}"]).requires_grad_(True)
    t = torch.from_numpy(g[f"t_{N}"])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = orc.peak_weighted_loss(p, t)
    L.backward()
    assert np.array_equal(L.detach().numpy(), g[f"loss_{N}"])
    assert np.array_equal(p.grad.numpy(), g[f"grad_{N}"])
    L64, g64 = orc.peak_weighted_loss_closed_form(g[f"p_{N}"], g[f"t_{N}"])
    assert abs(L64 - float(g[f"loss64_{N}"])) <= 1e-12 * max(1.0, abs(L64))
    assert np.max(np.abs(g64 - g[f"grad64_{N}"])) <= 1e-12


def test_loss_alpha_beta(golden):
    g = golden("loss.npz")
    L64, g64 = orc.peak_weighted_loss_closed_form(g["p_ab"], g["t_ab"], alpha=3.5, beta=1.0)
    assert abs(L64 - float(g["loss_ab"])) < 1e-5
    assert np.max(np.abs(g64 - g["grad_ab"])) < 1e-6


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    z = np.zeros((1, 4), np.uint32)
    out = orc.philox4x32_10(z, np.zeros((1, 2), np.uint32))[0]
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    out = orc.philox4x32_10(f, np.full((1, 2), 0xFFFFFFFF, np.uint32))[0]
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    c = np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32)
    k = np.array([[0xA4093822, 0x299F31D0]], np.uint32)
    out = orc.philox4x32_10(c, k)[0]
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_normals_moments():
    eps = orc.philox_eps(1234, range(8), 44289)
    assert eps.dtype == np.float32 and eps.shape == (8, 44289)
    assert abs(eps.mean()) < 5e-3 and abs(eps.std() - 1.0) < 5e-3
    # distinct pairs / seeds give distinct streams; same key reproduces
    assert not np.array_equal(eps[0], eps[1])
    assert np.array_equal(eps[3], orc.philox_eps(1234, [3], 44289)[0])
    assert not np.array_equal(eps[3], orc.philox_eps(1235, [3], 44289)[0])


def test_histogram_quantiles_within_one_bin():
    rng = np.random.default_rng(0)
    N, B, nbins = 600, 50, 128
    Y = (rng.normal(size=(N, B)) * rng.uniform(0.1, 2.0, size=B) + rng.normal(size=B) * 3).astype(np.float32)
    Y[::2] += 0.7  # two "members" -> mixture
    s1 = Y[:64].astype(np.float64).sum(0)
    s2 = (Y[:64].astype(np.float64) ** 2).sum(0)
    lo, inv_w = orc.hist_edges_from_pilot(64, s1, s2, nbins, 6.0)
    hist = orc.hist_accumulate(Y, lo, inv_w, nbins)
    assert hist.sum(1).tolist() == [N] * B
    S1, S2 = Y.astype(np.float64).sum(0), (Y.astype(np.float64) ** 2).sum(0)
    q_levels = (0.025, 0.5, 0.975)
    mean, std, qs = orc.finalize_from_state(N, S1, S2, hist, lo, inv_w, q_levels)
    tq = torch.quantile(torch.from_numpy(Y), torch.tensor(q_levels), dim=0).numpy()
    assert np.all(np.abs(qs - tq) <= 1.0 / inv_w[None, :])
    tm, ts = orc.reduce_mean_std(torch.from_numpy(Y))
    assert np.allclose(mean, tm.numpy(), rtol=1e-6, atol=1e-6)
    assert np.allclose(std, ts.numpy(), rtol=1e-5)


# ----------------------------------------------------------------------------- training step (row f1)
def test_train_oracle_reproduces_reference_trajectory(golden):
    """TrainOracle stepped with the eps / masks / lr / beta1 the reference's own train() used must
    land on the reference's losses and final (mu, rho)."""
    z = golden("train_steps.npz")
    M, S, spm = int(z["M"]), int(z["S"]), int(z["steps_per_member"])
    for m in range(M):
        orc_m = orc.TrainOracle(torch.from_numpy(z[f"mu0_{m}"]), torch.from_numpy(z[f"rho0_{m}"]))
        for j in range(spm):
            i = m * spm + j
            masks = [torch.from_numpy(z[f"mask{l}_{i}"]) for l in range(3)]
            loss, _, _, _ = orc_m.step(torch.from_numpy(z[f"xb_{i}"]), torch.from_numpy(z[f"yb_{i}"]),
                                       torch.from_numpy(z[f"eps_{i}"]), masks, lr=float(z["lr"][i]), beta1=float(z["beta1"][i]))
            np.testing.assert_allclose(loss.numpy(), z["loss"][i], rtol=2e-6)
        np.testing.assert_allclose(orc_m.mu.detach().numpy(), z[f"mu1_{m}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc_m.rho.detach().numpy(), z[f"rho1_{m}"], rtol=0, atol=2e-6)


def test_train_oracle_schedule_matches_reference(golden):
    """The OneCycleLR the oracle builds yields the lr / beta1 sequence the reference's optimiser saw."""
    z = golden("train_steps.npz")
    spm, epochs = int(z["steps_per_member"]), int(z["epochs"])
    o = orc.TrainOracle(torch.from_numpy(z["mu0_0"]), torch.from_numpy(z["rho0_0"]))
    sched = o.one_cycle(epochs, spm // epochs)
    for j in range(spm):
        g = o.opt.param_groups[0]
        assert abs(g["lr"] - z["lr"][j]) <= 1e-12 and abs(g["betas"][0] - z["beta1"][j]) <= 1e-12
        o.opt.step()
        if j + 1 < spm:
            sched.step()


def test_philox_keep_masks_rate_and_determinism():
    a = orc.philox_keep_masks(99, 3, 64)
    b = orc.philox_keep_masks(99, 3, 64)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert [x.shape for x in a] == [(64, 256), (64, 128), (64, 64)]
    rate = np.concatenate([x.reshape(-1) for x in a]).mean()
    assert abs(rate - 0.9) < 0.01
    c = orc.philox_keep_masks(99, 4, 64)
    assert not np.array_equal(a[0], c[0])
