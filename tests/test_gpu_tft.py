"""GPU parity of the TFT leg against oracle/tft_oracle.py and the reference's own vectors."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tft_oracle as tft

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _key_bias(F):
    off = 0
    for n, s in tft.param_spec(F):
        if n == "attention.in_proj_bias":
            return slice(off + 64, off + 128)
        off += int(np.prod(s))


def _engine(flat, F, T=30, **kw):
    from mcmc_amd.temporal_fusion import TFTEngine
    return TFTEngine(torch.as_tensor(flat).to(DEV), F, T, **kw)


def _windows(xb):
    """[B,T,F] batch -> (X [B*T,F], start [B]) on the device."""
    B, T, F = xb.shape
    return xb.reshape(B * T, F).contiguous().to(DEV), (torch.arange(B, dtype=torch.int32) * T).to(DEV)


def test_tft_eval_forward_and_predict_match_reference(golden):
    z = golden("tft_fwd.npz")
    F = z["X"].shape[1]
    te = _engine(z["flat"], F)
    X, start = _windows(torch.from_numpy(z["raw_batch"]))
    got = te.forward(X, start).cpu().numpy()
    np.testing.assert_allclose(got, z["raw_out"][:, -1, 0], rtol=0, atol=1e-5)
    # the whole predict(): 336 windows in one launch + the reference's padding
    from mcmc_amd.temporal_fusion import TFTModel
    m = TFTModel(num_features=F, device=DEV)
    m.model.load_flat(torch.from_numpy(z["flat"]).to(DEV))
    pred = m.predict(torch.from_numpy(z["X"]))
    assert pred.shape == (365,) and pred.dtype == np.float32 and np.all(pred[:29] == pred[29])
    np.testing.assert_allclose(pred, z["predict"], rtol=0, atol=1e-5)
    with pytest.raises(ValueError):
        m.predict(torch.from_numpy(z["X"])[:, :5])
    # the module's own forward: the reference's whole [batch, time, 1] output (temporal_fusion.py:120-153)
    full = m.model.eval()(torch.from_numpy(z["raw_batch"]).to(DEV))
    assert full.names == ("batch", "time", "features") and tuple(full.shape) == z["raw_out"].shape
    np.testing.assert_allclose(full.rename(None).cpu().numpy(), z["raw_out"], rtol=0, atol=1e-5)
    with pytest.raises(ValueError):
        m.model(torch.zeros(2, 30, F + 1, device=DEV))
    # forward() keeps its engine while the parameters are untouched, rebuilds it when one is written, and gives the same
    # numbers when the batch is cut into bounded window chunks
    eng0 = m.model._forward_engine(DEV, 30)
    assert m.model._forward_engine(DEV, 30) is eng0
    try:
        type(m.model).FORWARD_WINDOWS = 60   # two batch items per launch: 5 items -> 3 launches
        chunked = m.model(torch.from_numpy(z["raw_batch"]).to(DEV))
    finally:
        type(m.model).FORWARD_WINDOWS = 65536
    assert torch.equal(chunked.rename(None), full.rename(None)) and m.model._forward_engine(DEV, 30) is eng0
    with torch.no_grad():
        m.model.output_projection.module.bias.add_(0.25)
    assert m.model._forward_engine(DEV, 30) is not eng0
    moved = m.model(torch.from_numpy(z["raw_batch"]).to(DEV))
    np.testing.assert_allclose(moved.rename(None).cpu().numpy(), z["raw_out"] + 0.25, rtol=0, atol=1e-5)


@pytest.mark.parametrize("step", [0, 1])
def test_tft_train_forward_loss_and_gradient_match_oracle(golden, step):
    """Reference's recorded batch + dropout masks: prediction, loss, every parameter gradient, clip norm."""
    z = golden("tft_train.npz")
    F = z["X"].shape[1]
    xb, yb = torch.from_numpy(z[f"xb_{step}"]), torch.from_numpy(z[f"yb_{step}"])
    masks = [torch.from_numpy(z[f"mask{j}_{step}"]) for j in range(4)]
    o = tft.TFTTrainOracle(torch.from_numpy(z["flat0"]), F)
    loss, grad, total, pred = o.step(xb, yb, masks, lr=1e-3, beta1=0.9)
    te = _engine(z["flat0"], F)
    X, start = _windows(xb)
    got_loss = te.step(X, start, yb.to(DEV), 1e-3, 0.9, masks=masks).cpu()
    np.testing.assert_allclose(te.pred[:xb.shape[0]].cpu().numpy(), pred.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(got_loss.item(), loss.item(), rtol=2e-5)
    g = te.grad.cpu().numpy()
    ref = grad.numpy()
    off = 0
    for name, shape in tft.param_spec(F):
        k = int(np.prod(shape))
        a, b = g[off:off + k], ref[off:off + k]
        if name == "attention.in_proj_bias":  # key-bias gradient is exactly zero in exact arithmetic: noise on both sides
            a, b = np.concatenate([a[:64], a[128:]]), np.concatenate([b[:64], b[128:]])
            assert np.abs(g[off + 64:off + 128]).max() <= 1e-6
        scale = max(np.abs(b).max(), 1e-6)
        assert np.abs(a - b).max() <= 3e-5 * scale + 1e-7, (name, np.abs(a - b).max(), scale)
        off += k
    np.testing.assert_allclose(te.gnorm.item(), total.item(), rtol=3e-5)


@pytest.mark.parametrize("F,T,B", [(5, 12, 7), (24, 30, 3), (1, 2, 32), (11, 32, 5)])
def test_tft_other_widths_and_window_lengths(F, T, B):
    """Feature counts across the input-projection range and short / full-tile windows (the MFMA tiles always span 32 rows:
    rows past seq_len must stay inert): eval forward, loss and every parameter gradient against the oracle's autograd."""
    g = torch.Generator().manual_seed(100 * F + T)
    flat = torch.cat([(torch.randn(int(np.prod(sh)), generator=g) * (0.15 if len(sh) > 1 else 0.05)) + (1.0 if n.endswith("layer_norm.weight") else 0.0)
                      for n, sh in tft.param_spec(F)])
    xb = torch.randn(B, T, F, generator=g)
    yb = torch.randn(B, generator=g)
    o = tft.TFTTrainOracle(flat.clone(), F)
    loss, grad, total, pred = o.step(xb, yb, None, lr=1e-3, beta1=0.9)   # masks None: dropout off
    te = _engine(flat.numpy(), F, T)
    X, start = _windows(xb)
    np.testing.assert_allclose(te.forward(X, start).cpu().numpy(), pred.numpy(), rtol=0, atol=2e-5)
    # eval-mode step: no masks, no seed-driven dropout -> host masks of ones
    ones = [torch.ones(B, T, 64, dtype=torch.uint8), torch.ones(B, T, 64, dtype=torch.uint8), torch.ones(B * 4, T, T, dtype=torch.uint8),
            torch.ones(B, T, 64, dtype=torch.uint8)]
    te.p_drop = 0.0
    got_loss = te.step(X, start, yb.to(DEV), 1e-3, 0.9, masks=ones, update=False).cpu()
    np.testing.assert_allclose(got_loss.item(), loss.item(), rtol=5e-5)
    from mcmc_amd import _lib
    from mcmc_amd.engine import _p, _stream
    _lib.check(te.lib.mcmc_flat_adamw(te.h, _p(te.P.clone()), te.n, _p(te.dP), B, _p(te.grad), _p(te.adam_m.clone()), _p(te.adam_v.clone()),
                                      _p(te.norm_scratch), 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 1, _p(te.gnorm), _stream()), "adamw")
    gg, ref = te.grad.cpu().numpy(), grad.numpy()
    off = 0
    for name, shape in tft.param_spec(F):
        k = int(np.prod(shape))
        a, b = gg[off:off + k], ref[off:off + k]
        if name == "attention.in_proj_bias":
            a, b = np.concatenate([a[:64], a[128:]]), np.concatenate([b[:64], b[128:]])
        scale = max(np.abs(b).max(), 1e-6)
        assert np.abs(a - b).max() <= 5e-5 * scale + 1e-7, (name, np.abs(a - b).max(), scale)
        off += k


def test_tft_replays_reference_trajectory(golden):
    z = golden("tft_train.npz")
    F = z["X"].shape[1]
    te = _engine(z["flat0"], F)
    for i in range(int(z["n_steps"])):
        xb, yb = torch.from_numpy(z[f"xb_{i}"]), torch.from_numpy(z[f"yb_{i}"])
        masks = [torch.from_numpy(z[f"mask{j}_{i}"]) for j in range(4)]
        X, start = _windows(xb)
        loss = te.step(X, start, yb.to(DEV), float(z["lr"][i]), float(z["beta1"][i]), masks=masks).cpu()
        np.testing.assert_allclose(loss.item(), z["loss"][i], rtol=5e-5)
    d = np.abs(te.P.cpu().numpy() - z["flat1"])
    kb = _key_bias(F)
    rest = np.ones(d.size, dtype=bool)
    rest[kb] = False
    lr_sum = float(z["lr"].sum())
    # Adam's sign-like update: elements whose gradient is ~adam_eps may differ by a part of lr; all others tightly
    assert d[rest].max() <= 2.0 * lr_sum and np.quantile(d[rest], 0.999) <= 3e-6, (d[rest].max(), np.quantile(d[rest], 0.999))
    assert d[kb].max() <= 2.02 * lr_sum


def test_tft_philox_dropout_matches_oracle_masks(golden):
    z = golden("tft_fwd.npz")
    F = z["X"].shape[1]
    te = _engine(z["flat"], F)
    X = torch.from_numpy(z["X"]).to(DEV)
    keys = [3, 100, 200, 335]
    start = torch.tensor(keys, dtype=torch.int32, device=DEV)
    seed = 0xABCDEF0123456789
    got = te.forward(X, start, seed=seed).cpu().numpy()
    masks = tft.philox_masks(seed, keys, 30)
    xb = torch.stack([torch.from_numpy(z["X"][k:k + 30]) for k in keys])
    with torch.no_grad():
        want = tft.forward(xb, tft.unflatten(torch.from_numpy(z["flat"]), F), masks)[:, -1, 0].numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
    evalp = te.forward(X, start).cpu().numpy()
    assert np.abs(evalp - got).max() > 1e-3      # dropout really was live


def test_tft_saved_activations_equal_recompute(golden):
    """step() (train-mode forward that saves the windows' activations + ONE backward launch that derives the loss and
    dLoss/dpred in its prologue and reloads the activations: mcmc_tft_backward_loss) == the plain forward / loss / backward
    trio whose backward recomputes the forward pass: predictions, loss, the epoch-loss sum and every window's parameter
    gradient, bit for bit, with Philox dropout live and with a ragged batch after a full one."""
    from mcmc_amd import _lib
    from mcmc_amd.engine import _p, _stream
    z = golden("tft_fwd.npz")
    F = z["X"].shape[1]
    X = torch.from_numpy(z["X"]).to(DEV)
    te = _engine(z["flat"], F)
    lib, h = te.lib, te.h
    g = torch.Generator().manual_seed(2)
    for n_win, seed in ((32, 77), (13, 78)):
        start = torch.randperm(336, generator=g)[:n_win].to(dtype=torch.int32, device=DEV)
        yb = torch.randn(n_win, generator=g).to(DEV)
        acc = torch.full((1,), 0.25, dtype=torch.float64, device=DEV)
        loss = te.step(X, start, yb, 1e-3, 0.9, seed=seed, update=False, loss_acc=acc).clone()
        assert float(acc) == 0.25 + float(loss)
        assert torch.equal(te.step(X, start, yb, 1e-3, 0.9, seed=seed, update=False), loss)  # without the epoch sum
        pred = torch.empty(n_win, device=DEV)
        delta = torch.empty(n_win, device=DEV)
        l2 = torch.empty(1, device=DEV)
        dP = torch.full((n_win, te.n), float("nan"), device=DEV)
        none = [_p(None)] * 4
        st = _stream()
        _lib.check(lib.mcmc_tft_forward(h, _p(X), X.shape[0], _p(start), n_win, 30, F, _p(te.P), _p(te.PT), 2, *none, seed, 0.1, _p(pred), st), "f")
        _lib.check(lib.mcmc_train_loss(h, _p(pred), _p(yb), 1, 1, n_win, 2.0, 0.5, _p(delta), _p(l2), st), "l")
        _lib.check(lib.mcmc_tft_backward(h, _p(X), X.shape[0], _p(start), n_win, 30, F, _p(te.P), _p(te.PT), 2, *none, seed, 0.1, _p(delta),
                                         _p(dP), st), "b")
        assert torch.equal(te.pred[:n_win], pred) and torch.equal(loss, l2)
        assert torch.equal(te.dP[:n_win], dP)
    # an updating step keeps the transposed image in step with the parameters (mcmc_tft_adamw writes both)
    te.step(X, start, yb, 1e-3, 0.9, seed=5)
    fresh = torch.empty_like(te.PT)
    _lib.check(lib.mcmc_tft_transpose(h, _p(te.P), _p(fresh), F, _stream()), "t")
    assert torch.equal(te.PT, fresh) and not torch.equal(te.P.cpu(), torch.from_numpy(z["flat"]))


def test_tft_model_trains_and_predicts_end_to_end(golden):
    from mcmc_amd.temporal_fusion import TFTModel
    z = golden("predict_cfg1.npz")
    X, y = torch.from_numpy(z["X"]), torch.from_numpy(z["y"])
    torch.manual_seed(4)
    m = TFTModel(num_features=X.shape[1], seq_length=30, batch_size=32, device=DEV, seed=9)
    before = m.predict(X)
    m.train(X.refine_names("time", "features"), y.refine_names("time"), epochs=25)
    after = m.predict(X)
    mse0 = float(((before[29:] - y.numpy()[29:]) ** 2).mean())
    mse1 = float(((after[29:] - y.numpy()[29:]) ** 2).mean())
    assert m.epochs_trained == 25 and mse1 < 0.5 * mse0 and mse1 < 0.6, (mse0, mse1)
    Xs, ys = m.create_sequences(X.refine_names("time", "features"), y.refine_names("time"))
    assert Xs.names == ("batch", "time", "features") and tuple(Xs.shape) == (336, 30, X.shape[1]) and ys.names == ("batch",)
    assert torch.equal(Xs.rename(None)[7], X[7:37]) and ys.rename(None)[7] == y[36]


def test_tft_capi_rejects_bad_arguments():
    from mcmc_amd.engine import Engine
    eng = Engine(torch.device(DEV))
    lib, h = eng.lib, eng.ctx.handle
    buf = torch.zeros(1 << 18, device=DEV)
    p, z = C.c_void_p(buf.data_ptr()), C.c_void_p(0)
    assert lib.mcmc_tft_forward(h, p, 4096, p, 4, 33, 11, p, p, 0, z, z, z, z, 0, 0.1, p, z) == -1     # seq_len > 32
    assert lib.mcmc_tft_forward(h, p, 4096, p, 4, 30, 25, p, p, 0, z, z, z, z, 0, 0.1, p, z) == -1     # too many features
    assert lib.mcmc_tft_forward(h, p, 4096, p, 4, 30, 11, p, p, 1, p, p, z, p, 0, 0.1, p, z) == -1     # host masks incomplete
    assert lib.mcmc_tft_forward(h, p, 20, p, 4, 30, 11, p, p, 0, z, z, z, z, 0, 0.1, p, z) == -1   # fewer rows than seq_len
    assert lib.mcmc_tft_transpose(h, p, p, 11, z) == -1                                          # aliased
    assert lib.mcmc_flat_adamw(h, p, 10, p, 1, p, p, p, p, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, 0, z, z) == -1
    assert lib.mcmc_tft_param_count(11) == 67585 and lib.mcmc_tft_param_count(0) == -1


def test_pipeline_task_with_tft_leg(golden):
    """main.py's two GPU legs for one feature subset: record carries 'tft' and 'bayesian' with the reference's keys."""
    from mcmc_amd import pipeline
    z = golden("glue_cfg1.npz")
    feats, temp, target = (torch.from_numpy(z[k]) for k in ("features", "temporal", "target"))
    torch.manual_seed(2)
    name, rec = pipeline.process_bayesian_task(feats, temp, target, (0, 1), torch.device(DEV), epochs=30, predict_samples=50, seed=5,
                                               include_tft=True, log=lambda *_: None)
    assert name == "0_1" and set(rec) == {"tft", "bayesian"}
    assert set(rec["tft"]) == {"predictions", "metrics"} and len(rec["tft"]["predictions"]) == 365
    assert rec["tft"]["metrics"]["mape"] < 15.0 and rec["bayesian"]["metrics"]["mape"] < 15.0, (rec["tft"]["metrics"], rec["bayesian"]["metrics"])
