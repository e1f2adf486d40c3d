"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE.  Imports /root/reference (read-only) to produce small input/output
vectors; the reference never travels to the GPU box, only these fixtures do.  Run:

    python oracle/gen_golden.py

What is captured:
  predict_small.npz   weights (mu,rho) of M seeded BayesianNetworks, X, the eps the
                      reference drew (torch.randn_like wrapped), all_predictions, mean, std,
                      torch.quantile levels.
  predict_cfg1.npz    config 1/2: data.py series + main.py:94-130 scaling -> X [365,11];
                      M=5, S=100; eps is NOT stored (88 MB) - it is regenerated from
                      ``eps_seed`` with torch's CPU generator in the reference's draw order
                      (oracle.reference_rng_eps); stores mean, std, quantiles and a slice
                      of all_predictions.
  predict_trained.npz the reference's own train() (main.py:169: 5 members, batch 32, 10 samples per step, 40 epochs,
                      seeded, CPU) on the config-1 series, then its predict() (main.py:170) with recorded noise:
                      the TRAINED (mu, rho), X, y, mean, std, quantiles, head / tail of all_predictions; eps is
                      regenerated from ``eps_seed`` like predict_cfg1.  Pins the predict kernels on the weight and
                      pre-activation distribution a trained ensemble has (every other predict fixture uses
                      init-time members).
  train_fwd.npz       train-mode forward (noise + live dropout masks, captured by a forward
                      hook on model.dropout): eps, masks, outputs.
  loss.npz            peak_weighted_loss value + autograd grad at N in {1, 13, 32, 4096}.
  glue_cfg1.npz       main.py:71-130,174-177 run with the real sklearn StandardScaler and the reference's
                      utils.evaluate_predictions on the config-1 series for several feature subsets
                      (main.py itself needs prophet, absent here: its few lines are re-run verbatim in
                      spirit by gen_glue below, on the reference's data.py output).
  tft_fwd.npz         reference TemporalFusionTransformer (seeded init) on the config-1 scaled series:
                      flat parameters, TFTModel.predict output [365] and the raw model output of a batch.
  tft_train.npz       the reference's TFTModel.train run on CPU (2 epochs of a 32- and a 9-sequence
                      batch): initial parameters, per step the batch, the 4 dropout keep-masks
                      (torch.nn.functional.dropout wrapped), lr / beta1, loss; final parameters.
  train_steps.npz     the reference's own GPUBayesianEnsemble.train run on CPU (2 members, 2 epochs
                      of a 32-row and a 13-row batch, 2 samples): initial (mu, rho), and per step the
                      batch, eps, dropout masks, lr / beta1 the optimiser used, loss; final (mu, rho).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))

from models.bayesian_ensemble import BayesianNetwork, GPUBayesianEnsemble  # noqa: E402  (reference)
from utils import peak_weighted_loss as ref_loss  # noqa: E402  (reference)
from data import SyntheticDataGenerator  # noqa: E402  (reference)

from oracle import bnn_oracle as orc  # noqa: E402

Q_LEVELS = (0.025, 0.5, 0.975)


class EpsRecorder:
    """Wrap torch.randn_like so every draw of the reference is recorded in order."""

    def __enter__(self):
        self.draws = []
        self._orig = torch.randn_like

        def wrapped(t, *a, **k):
            out = self._orig(t, *a, **k)
            self.draws.append(out.detach().clone().reshape(-1))
            return out

        torch.randn_like = wrapped
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


def make_members(input_dim: int, M: int, seed: int, perturb: bool = True):
    torch.manual_seed(seed)
    members = []
    for _ in range(M):
        net = BayesianNetwork(input_dim)
        if perturb:  # untrained nets have bias_mu == 0 and rho == -3 everywhere: make them generic
            with torch.no_grad():
                for l in list(net.hidden_layers) + [net.output_layer]:
                    l.bias_mu.add_(0.1 * torch.randn_like(l.bias_mu))
                    l.weight_rho.add_(0.5 * torch.randn_like(l.weight_rho))
                    l.bias_rho.add_(0.5 * torch.randn_like(l.bias_rho))
        members.append(net)
    return members


def run_reference_predict(members, X, S, eps_seed):
    ens = GPUBayesianEnsemble(input_dim=X.shape[1], n_models=len(members), device=torch.device("cpu"))
    ens.models = members
    captured = {}
    orig_reshape = torch.Tensor.reshape

    torch.manual_seed(eps_seed)
    with EpsRecorder() as rec:
        # capture all_predictions (bayesian_ensemble.py:242) through a mean() hook-free route:
        # re-run the same loops is not needed - std/mean are returned; all_predictions is
        # rebuilt below from the recorded eps through the reference modules' own forward.
        mean, std = ens.predict(X, num_samples=S)
    P = orc.n_params(X.shape[1])
    eps = torch.cat(rec.draws).reshape(len(members), S, P)
    return mean.rename(None), std.rename(None), eps


def reference_all_predictions(members, X, eps):
    """Replay the recorded eps through the reference modules (randn_like patched to feed them)."""
    M, S, P = eps.shape
    offs = orc.param_offsets(X.shape[1])
    out = torch.empty(M * S, X.shape[0])
    orig = torch.randn_like
    try:
        for m, net in enumerate(members):
            net.eval()
            for s in range(S):
                queue = []
                for (wo, bo, k_in, k_out) in offs:
                    queue.append(eps[m, s, wo:wo + k_in * k_out].view(k_out, k_in))
                    queue.append(eps[m, s, bo:bo + k_out])
                it = iter(queue)
                torch.randn_like = lambda t, *a, **k: next(it).clone()
                with torch.no_grad():
                    out[m * S + s] = net(X, sample=True).rename(None)
    finally:
        torch.randn_like = orig
    return out


def gen_predict_small():
    F_in, B, M, S = 11, 37, 3, 6
    members = make_members(F_in, M, seed=101)
    torch.manual_seed(7)
    X = torch.randn(B, F_in)
    mean, std, eps = run_reference_predict(members, X, S, eps_seed=202)
    allp = reference_all_predictions(members, X, eps)
    assert torch.equal(allp.mean(0), mean) and torch.equal(allp.std(0), std)
    mu, rho = orc.pack_members(members)
    q = torch.quantile(allp, torch.tensor(Q_LEVELS), dim=0)
    np.savez_compressed(os.path.join(OUT, "predict_small.npz"), X=X.numpy(), mu=mu.numpy(), rho=rho.numpy(),
                        eps=eps.numpy(), all_predictions=allp.numpy(), mean=mean.numpy(), std=std.numpy(),
                        q_levels=np.array(Q_LEVELS), quantiles=q.numpy())
    # other input widths the 32 feature subsets produce (main.py:38-57 -> F = 7..11)
    for F_alt in (7, 9):
        members = make_members(F_alt, 2, seed=300 + F_alt)
        torch.manual_seed(8)
        Xa = torch.randn(19, F_alt)
        mean, std, eps = run_reference_predict(members, Xa, 4, eps_seed=400 + F_alt)
        allp = reference_all_predictions(members, Xa, eps)
        mu, rho = orc.pack_members(members)
        np.savez_compressed(os.path.join(OUT, f"predict_small_F{F_alt}.npz"), X=Xa.numpy(), mu=mu.numpy(),
                            rho=rho.numpy(), eps=eps.numpy(), all_predictions=allp.numpy(), mean=mean.numpy(),
                            std=std.numpy())


def make_cfg1_inputs(seed: int):
    """data.py defaults + main.py:94-130 (concat, two StandardScalers) -> X_scaled, y_scaled."""
    from sklearn.preprocessing import StandardScaler
    torch.manual_seed(seed)
    data = SyntheticDataGenerator().generate_data()
    X = torch.cat([data.features.rename(None), data.temporal.rename(None)], dim=1)
    y = data.target.rename(None)
    sx, sy = StandardScaler(), StandardScaler()
    Xs = torch.tensor(sx.fit_transform(X.numpy()))
    ys = torch.tensor(sy.fit_transform(y.numpy().reshape(-1, 1))).squeeze(-1)
    return X, y, Xs, ys, float(sy.scale_[0]), float(sy.mean_[0])


def gen_predict_cfg1():
    M, S = 5, 100
    X_raw, y_raw, Xs, ys, y_scale, y_mean = make_cfg1_inputs(seed=2025)
    assert Xs.dtype == torch.float32 and Xs.shape == (365, 11)
    members = make_members(11, M, seed=11)
    eps_seed = 12345
    mean, std, eps = run_reference_predict(members, Xs, S, eps_seed=eps_seed)
    # the reference's draws == torch CPU generator stream in draw order (no storage needed)
    assert torch.equal(eps, orc.reference_rng_eps(eps_seed, M, S, 11))
    allp = reference_all_predictions(members, Xs, eps)
    assert torch.equal(allp.mean(0), mean) and torch.equal(allp.std(0), std)
    mu, rho = orc.pack_members(members)
    q = torch.quantile(allp, torch.tensor(Q_LEVELS), dim=0)
    # main.py:174-177 de-scaling
    descaled_mean = mean.numpy().astype(np.float64) * y_scale + y_mean
    descaled_std = std.numpy() * y_scale
    np.savez_compressed(os.path.join(OUT, "predict_cfg1.npz"), X_raw=X_raw.numpy(), y_raw=y_raw.numpy(),
                        X=Xs.numpy(), y=ys.numpy(), y_scale=y_scale, y_mean=y_mean, mu=mu.numpy(), rho=rho.numpy(),
                        eps_seed=eps_seed, M=M, S=S, mean=mean.numpy(), std=std.numpy(),
                        q_levels=np.array(Q_LEVELS), quantiles=q.numpy(),
                        all_predictions_head=allp[:8].numpy(), all_predictions_tail=allp[-4:].numpy(),
                        descaled_mean=descaled_mean, descaled_std=descaled_std)


def gen_predict_trained(epochs: int = 40):
    """main.py:169-170 on the reference itself: train() then predict() on the config-1 series."""
    import contextlib
    import io
    M, S = 5, 100
    X_raw, y_raw, Xs, ys, y_scale, y_mean = make_cfg1_inputs(seed=2025)
    ens = GPUBayesianEnsemble(input_dim=Xs.shape[1], n_models=M, device=torch.device("cpu"), batch_size=32)
    torch.manual_seed(777)
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        ens.train(Xs.refine_names("time", "features"), ys.refine_names("time"), epochs=epochs, num_samples=10)
    assert len(ens.models) == M
    members = list(ens.models)
    eps_seed = 54321
    mean, std, eps = run_reference_predict(members, Xs, S, eps_seed=eps_seed)
    assert torch.equal(eps, orc.reference_rng_eps(eps_seed, M, S, 11))
    allp = reference_all_predictions(members, Xs, eps)
    assert torch.equal(allp.mean(0), mean) and torch.equal(allp.std(0), std)
    mu, rho = orc.pack_members(members)
    q = torch.quantile(allp, torch.tensor(Q_LEVELS), dim=0)
    # the largest pre-activations of the trained members at sample=False: what the bf16 kernels' ELU range has to cover
    with torch.no_grad():
        zmax = []
        for net in members:
            h = Xs
            for layer in net.hidden_layers:
                z = layer(h, sample=False)
                zmax.append(float(z.abs().max()))
                h = torch.nn.functional.elu(z)
    np.savez_compressed(os.path.join(OUT, "predict_trained.npz"), X=Xs.numpy(), y=ys.numpy(), y_scale=y_scale, y_mean=y_mean,
                        mu=mu.numpy(), rho=rho.numpy(), eps_seed=eps_seed, M=M, S=S, epochs=epochs, mean=mean.numpy(), std=std.numpy(),
                        q_levels=np.array(Q_LEVELS), quantiles=q.numpy(), all_predictions_head=allp[:8].numpy(),
                        all_predictions_tail=allp[-4:].numpy(), preact_absmax=np.array(zmax).reshape(M, 3),
                        train_log=np.array(log.getvalue().splitlines()[-6:]))
    resid = (mean - ys).pow(2).mean().sqrt()
    print(f"predict_trained: {epochs} epochs; scaled-space RMSE of the mean {float(resid):.3f}; mean std {float(std.mean()):.3f}; "
          f"|mu| max {float(mu.abs().max()):.2f}, rho range [{float(rho.min()):.2f}, {float(rho.max()):.2f}]; pre-activation max {max(zmax):.1f}")


def gen_train_fwd():
    F_in, B, npass = 11, 13, 3   # 13 = the reference's last mini-batch (365 % 32)
    members = make_members(F_in, 1, seed=55)
    net = members[0]
    net.train()
    masks = []

    def hook(mod, inp, out):
        x = inp[0]
        # mask = (out != 0) is ambiguous where x == 0; recover from out/x where x != 0
        keep = torch.where(x != 0, (out != 0), torch.ones_like(out, dtype=torch.bool))
        masks.append(keep.to(torch.uint8).clone())

    h = net.dropout.register_forward_hook(hook)
    torch.manual_seed(9)
    X = torch.randn(B, F_in)
    outs = []
    torch.manual_seed(77)
    with EpsRecorder() as rec:
        for _ in range(npass):
            with torch.no_grad():
                outs.append(net(X, sample=True).rename(None).clone())
    h.remove()
    P = orc.n_params(F_in)
    eps = torch.cat(rec.draws).reshape(npass, P)
    mu, rho = orc.pack_members(members)
    np.savez_compressed(os.path.join(OUT, "train_fwd.npz"), X=X.numpy(), mu=mu.numpy(), rho=rho.numpy(),
                        eps=eps.numpy(), out=torch.stack(outs).numpy(),
                        mask0=torch.stack(masks[0::3]).numpy(), mask1=torch.stack(masks[1::3]).numpy(),
                        mask2=torch.stack(masks[2::3]).numpy())


def gen_loss():
    rec = {}
    for N in (1, 2, 13, 32, 4096):
        torch.manual_seed(1000 + N)
        t = torch.randn(N)
        t = t + (torch.rand(N) > 0.9).float() * 2.0  # a few peaks
        p = (t + 0.3 * torch.randn(N)).requires_grad_(True)
        L = ref_loss(p, t)
        L.backward()
        p64 = p.detach().double().requires_grad_(True)
        L64 = ref_loss(p64, t.double())
        L64.backward()
        rec[f"p_{N}"] = p.detach().numpy()
        rec[f"t_{N}"] = t.numpy()
        rec[f"loss_{N}"] = L.detach().numpy()
        rec[f"grad_{N}"] = p.grad.numpy()
        rec[f"loss64_{N}"] = L64.detach().numpy()
        rec[f"grad64_{N}"] = p64.grad.numpy()
    # non-default alpha/beta
    torch.manual_seed(5)
    t = torch.randn(257)
    p = (t + 0.5 * torch.randn(257)).requires_grad_(True)
    L = ref_loss(p, t, alpha=3.5, beta=1.0)
    L.backward()
    rec.update(p_ab=p.detach().numpy(), t_ab=t.numpy(), loss_ab=L.detach().numpy(), grad_ab=p.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **rec)


def gen_train_steps():
    """Run the reference's train() itself on CPU and record everything its RNG decided."""
    import models.bayesian_ensemble as ref_mod  # reference
    F_in, N, M, S, epochs = 7, 45, 2, 2, 2
    torch.manual_seed(31)
    X = torch.randn(N, F_in)
    y = torch.randn(N) + (torch.rand(N) > 0.85).float() * 2.0
    inits, xbs, ybs, losses, lrs, beta1s, masks = [], [], [], [], [], [], []

    orig_init = ref_mod.BayesianNetwork.__init__
    orig_fwd = ref_mod.BayesianNetwork.forward
    orig_loss = ref_mod.peak_weighted_loss
    orig_drop = torch.nn.Dropout.forward
    orig_step = torch.optim.AdamW.step

    def init_wrap(self, *a, **k):
        orig_init(self, *a, **k)
        inits.append(orc.pack_members([self]))

    def fwd_wrap(self, x, sample=False):
        xbs.append(x.rename(None).detach().clone())
        return orig_fwd(self, x, sample)

    def loss_wrap(p, t, *a, **k):
        L = orig_loss(p, t, *a, **k)
        ybs.append(t.detach().clone())
        losses.append(L.detach().clone())
        return L

    def drop_wrap(self, x):
        out = orig_drop(self, x)
        keep = torch.where(x != 0, (out != 0), torch.ones_like(out, dtype=torch.bool))
        masks.append(keep.to(torch.uint8).clone())
        return out

    def step_wrap(self, *a, **k):
        lrs.append(float(self.param_groups[0]["lr"]))
        beta1s.append(float(self.param_groups[0]["betas"][0]))
        assert self.param_groups[0]["betas"][1] == 0.999 and self.param_groups[0]["eps"] == 1e-8
        assert self.param_groups[0]["weight_decay"] == 0.01
        return orig_step(self, *a, **k)

    ens = GPUBayesianEnsemble(input_dim=F_in, n_models=M, device=torch.device("cpu"), batch_size=32)
    ref_mod.BayesianNetwork.__init__ = init_wrap
    ref_mod.BayesianNetwork.forward = fwd_wrap
    ref_mod.peak_weighted_loss = loss_wrap
    torch.nn.Dropout.forward = drop_wrap
    torch.optim.AdamW.step = step_wrap
    try:
        torch.manual_seed(4242)
        with EpsRecorder() as rec:
            ens.train(X, y, epochs=epochs, num_samples=S)
    finally:
        ref_mod.BayesianNetwork.__init__ = orig_init
        ref_mod.BayesianNetwork.forward = orig_fwd
        ref_mod.peak_weighted_loss = orig_loss
        torch.nn.Dropout.forward = orig_drop
        torch.optim.AdamW.step = orig_step
    P = orc.n_params(F_in)
    n_steps = len(losses)               # M * epochs * 2 batches
    assert n_steps == M * epochs * 2 and len(xbs) == n_steps * S and len(masks) == n_steps * S * 3
    eps = torch.cat(rec.draws).reshape(n_steps, S, P)
    out = dict(X=X.numpy(), y=y.numpy(), M=M, S=S, epochs=epochs, steps_per_member=n_steps // M,
               lr=np.array(lrs), beta1=np.array(beta1s), loss=torch.stack(losses).numpy())
    for m in range(M):
        out[f"mu0_{m}"], out[f"rho0_{m}"] = (t.numpy()[0] for t in inits[m])
        mu1, rho1 = orc.pack_members([ens.models[m]])
        out[f"mu1_{m}"], out[f"rho1_{m}"] = mu1.numpy()[0], rho1.numpy()[0]
    for i in range(n_steps):
        out[f"xb_{i}"] = xbs[i * S].numpy()
        assert all(torch.equal(xbs[i * S], xbs[i * S + s]) for s in range(S))
        out[f"yb_{i}"] = ybs[i].numpy()
        out[f"eps_{i}"] = eps[i].numpy()
        for l in range(3):
            out[f"mask{l}_{i}"] = torch.stack([masks[(i * S + s) * 3 + l] for s in range(S)]).numpy()
    np.savez_compressed(os.path.join(OUT, "train_steps.npz"), **out)


def gen_tft():
    from models.temporal_fusion import TFTModel  # reference
    from oracle import tft_oracle as tft
    _, _, Xs, ys, _, _ = make_cfg1_inputs(seed=2025)
    F_in = Xs.shape[1]
    torch.manual_seed(909)
    ref = TFTModel(num_features=F_in, seq_length=30, batch_size=32, device="cpu")
    with torch.no_grad():  # fresh LayerNorms are (1, 0) and biases small: make every term generic
        for n, q in ref.model.named_parameters():
            if "layer_norm" in n or n.endswith("bias"):
                q.add_(0.1 * torch.randn_like(q))
    flat = tft.flatten(dict(ref.model.named_parameters()), F_in)
    preds = ref.predict(Xs)
    seqs = ref.create_sequences(Xs).rename(None)
    ref.model.eval()
    with torch.no_grad():
        raw = ref.model(seqs[100:105]).rename(None)
    np.savez_compressed(os.path.join(OUT, "tft_fwd.npz"), X=Xs.numpy(), flat=flat.numpy(), predict=np.asarray(preds, dtype=np.float32),
                        raw_batch=seqs[100:105].numpy(), raw_out=raw.numpy())

    # ---- training trajectory: the reference's own train() with everything its RNG decided recorded
    N, epochs = 70, 2                       # 41 sequences -> batches of 32 and 9
    Xt, yt = Xs[:N].clone(), ys[:N].clone()
    torch.manual_seed(910)
    ref = TFTModel(num_features=F_in, seq_length=30, batch_size=32, device="cpu")
    flat0 = tft.flatten(dict(ref.model.named_parameters()), F_in)
    import models.temporal_fusion as ref_mod
    import torch.nn.functional as Fn
    masks, xbs, ybs, losses, lrs, beta1s = [], [], [], [], [], []
    orig_drop, orig_loss, orig_step, orig_fwd = Fn.dropout, ref_mod.peak_weighted_loss, torch.optim.AdamW.step, ref.model.forward

    def drop_wrap(x, p=0.5, training=True, inplace=False):
        out = orig_drop(x, p, training, inplace)
        if training and p > 0:
            masks.append(torch.where(x != 0, (out != 0), torch.ones_like(out, dtype=torch.bool)).to(torch.uint8).clone())
        return out

    def loss_wrap(pr, t, *a, **k):
        L = orig_loss(pr, t, *a, **k)
        ybs.append(t.detach().clone())
        losses.append(L.detach().clone())
        return L

    def step_wrap(self, *a, **k):
        lrs.append(float(self.param_groups[0]["lr"]))
        beta1s.append(float(self.param_groups[0]["betas"][0]))
        return orig_step(self, *a, **k)

    def fwd_wrap(x):
        xbs.append(x.rename(None).detach().clone())
        return orig_fwd(x)

    Fn.dropout, ref_mod.peak_weighted_loss, torch.optim.AdamW.step, ref.model.forward = drop_wrap, loss_wrap, step_wrap, fwd_wrap
    try:
        torch.manual_seed(911)
        ref.train(Xt, yt, epochs=epochs)
    finally:
        Fn.dropout, ref_mod.peak_weighted_loss, torch.optim.AdamW.step, ref.model.forward = orig_drop, orig_loss, orig_step, orig_fwd
    n_steps = len(losses)
    assert n_steps == epochs * 2 and len(masks) == 4 * n_steps and len(xbs) == n_steps
    out = dict(X=Xt.numpy(), y=yt.numpy(), flat0=flat0.numpy(), flat1=tft.flatten(dict(ref.model.named_parameters()), F_in).numpy(),
               lr=np.array(lrs), beta1=np.array(beta1s), loss=torch.stack(losses).numpy(), n_steps=n_steps)
    for i in range(n_steps):
        out[f"xb_{i}"], out[f"yb_{i}"] = xbs[i].numpy(), ybs[i].numpy()
        for j in range(4):
            out[f"mask{j}_{i}"] = masks[4 * i + j].numpy()
    np.savez_compressed(os.path.join(OUT, "tft_train.npz"), **out)


def gen_glue():
    from sklearn.preprocessing import StandardScaler
    from itertools import combinations
    from utils import evaluate_predictions as ref_eval  # reference
    X_raw, y_raw, _, _, _, _ = make_cfg1_inputs(seed=2025)
    features, temporal = X_raw[:, :5], X_raw[:, 5:]
    combos = [tuple()]
    for r in range(1, 6):
        combos.extend(combinations(range(5), r))
    out = dict(features=features.numpy(), temporal=temporal.numpy(), target=y_raw.numpy(), n_combos=len(combos),
               combo_order=np.array(["_".join(map(str, c)) if c else "baseline" for c in combos]))
    torch.manual_seed(77)
    for tag, combo in (("baseline", ()), ("0", (0,)), ("1_3", (1, 3)), ("0_1_2_3_4", (0, 1, 2, 3, 4))):
        f = features.index_select(1, torch.tensor(list(combo))) if combo else features
        X = torch.cat([f, temporal], dim=1)
        sx, sy = StandardScaler(), StandardScaler()
        Xs = torch.tensor(sx.fit_transform(X.numpy()))
        ys = torch.tensor(sy.fit_transform(y_raw.numpy().reshape(-1, 1))).squeeze(-1)
        fake_mean = ys + 0.2 * torch.randn(ys.shape)   # stands in for the ensemble's scaled-space output
        fake_std = 0.1 + 0.05 * torch.rand(ys.shape)
        m = sy.inverse_transform(fake_mean.reshape(-1, 1)).flatten()
        sd = fake_std * sy.scale_[0]
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            met = ref_eval(y_raw.numpy(), m)
        out.update({f"Xs_{tag}": Xs.numpy(), f"ys_{tag}": ys.numpy(), f"fake_mean_{tag}": fake_mean.numpy(),
                    f"fake_std_{tag}": fake_std.numpy(), f"mean_{tag}": np.asarray(m), f"std_{tag}": sd.numpy(),
                    f"rmse_{tag}": met["rmse"], f"mape_{tag}": met["mape"]})
    np.savez_compressed(os.path.join(OUT, "glue_cfg1.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gens = dict(predict_small=gen_predict_small, predict_cfg1=gen_predict_cfg1, predict_trained=gen_predict_trained,
                train_fwd=gen_train_fwd, loss=gen_loss, train_steps=gen_train_steps, glue=gen_glue, tft=gen_tft)
    for name in (sys.argv[1:] or list(gens)):  # no argument: every fixture
        gens[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
