"""Host-side mirror of /root/reference/models/temporal_fusion.py for the MI355X path (row f4).

``TFTModel`` keeps the reference's constructor, ``create_sequences``, ``train`` and ``predict``
(temporal_fusion.py:156-300) and a ``.model`` whose parameters carry the reference's names and
initialisation, so checkpoints and ``main.py:141-165`` work unchanged.  All arithmetic runs in
libmcmc_hip.so (``mcmc_tft_*``, ``mcmc_train_loss``, ``mcmc_flat_adamw``): a fused per-window
forward / backward that evaluates only what the reference reads - the last timestep of each window.
There is no CPU path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import Engine, _p, _stream, device_guarded
from .train import one_cycle_schedule

HIDDEN, HEADS = 64, 4
_GOLDEN = 0x9E3779B97F4A7C15
_U64 = 0xFFFFFFFFFFFFFFFF


class TimeDistributed(nn.Module):
    """Parameter holder with the reference's attribute name (temporal_fusion.py:10-13)."""

    def __init__(self, module):
        super().__init__()
        self.module = module


class GatedResidualNetwork(nn.Module):
    """temporal_fusion.py:41-60 (parameters only; the arithmetic is in tft.hip)."""

    def __init__(self, input_size: int, hidden_size: int, output_size: int, dropout: float = 0.1):
        super().__init__()
        if not (input_size == hidden_size == output_size == HIDDEN):
            raise ValueError(f"the HIP kernels are built for {HIDDEN}-wide GRNs")
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        self.fc1 = TimeDistributed(nn.Linear(input_size, hidden_size))
        self.fc2 = TimeDistributed(nn.Linear(hidden_size, output_size))
        self.skip = None
        self.gate = TimeDistributed(nn.Linear(input_size + output_size, output_size))
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(output_size)


class TemporalFusionTransformer(nn.Module):
    """temporal_fusion.py:91-118: same sub-module names, parameter order and default initialisation."""

    def __init__(self, num_features: int, hidden_size: int = HIDDEN, num_heads: int = HEADS, dropout: float = 0.1):
        super().__init__()
        if hidden_size != HIDDEN or num_heads != HEADS:
            raise ValueError(f"the HIP kernels are built for hidden_size {HIDDEN}, {HEADS} heads")
        self.num_features = num_features
        self.hidden_size = hidden_size
        self.p_drop = dropout
        self.input_projection = TimeDistributed(nn.Linear(num_features, hidden_size))
        self.feature_grn = GatedResidualNetwork(hidden_size, hidden_size, hidden_size, dropout)
        self.temporal_grn = GatedResidualNetwork(hidden_size, hidden_size, hidden_size, dropout)
        self.attention = nn.MultiheadAttention(hidden_size, num_heads, dropout=dropout)
        self.final_grn = GatedResidualNetwork(hidden_size, hidden_size, hidden_size, dropout)
        self.output_projection = TimeDistributed(nn.Linear(hidden_size, 1))

    def flat_parameters(self) -> torch.Tensor:
        """One fp32 vector in named_parameters() order - the layout of include/mcmc_hip.h."""
        return torch.cat([p.detach().reshape(-1).float() for p in self.parameters()])

    def load_flat(self, flat: torch.Tensor):
        with torch.no_grad():
            off = 0
            for p in self.parameters():
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """temporal_fusion.py:120-153: x [batch, time, features] -> [batch, time, 1], eval mode, on the HIP kernels.

        The kernels evaluate the LAST timestep of a window (all that train() and predict() read, :249,:293).  The model has
        no positional term - the GRNs act per timestep and the attention treats its keys as a set - so the output at
        timestep t equals the last-timestep output of the same window with rows t and T-1 exchanged: one kernel launch
        over batch x time such windows gives the reference's whole [batch, time, 1] tensor (attention sums in another
        key order: equal to ~1e-7)."""
        if self.training and self.p_drop > 0:
            raise RuntimeError("TemporalFusionTransformer.forward runs in eval mode on the HIP path (call .eval()); "
                               "train-mode steps go through TFTModel.train / TFTEngine.step")
        x = x.rename(None) if x.has_names() else x
        if x.dim() != 3 or x.shape[-1] != self.num_features:
            raise ValueError(f"Expected {self.num_features} features, got {x.shape[-1]}")
        if not x.is_cuda:
            raise RuntimeError(f"x must live on a HIP device (got {x.device}); mcmc_amd has no CPU path")
        B, T, F = x.shape
        idx = torch.arange(T, device=x.device).repeat(T, 1)      # idx[t] = window order for output timestep t
        t = torch.arange(T, device=x.device)
        idx[t, t], idx[t, T - 1] = T - 1, t                      # rows t and T-1 exchanged
        te = self._forward_engine(x.device, T)
        out = torch.empty(B * T, dtype=torch.float32, device=x.device)
        # the T-fold blow-up of the input ([batch, out timestep, row, feature]) is built for a bounded number of batch items
        # at a time, and the int32 window starts stay far below 2^31
        per = max(1, self.FORWARD_WINDOWS // T)
        xf = x.to(torch.float32)
        for b0 in range(0, B, per):
            nb = min(per, B - b0)
            windows = xf[b0:b0 + nb][:, idx, :].reshape(nb * T * T, F).contiguous()
            start64 = torch.arange(nb * T, device=x.device, dtype=torch.int64) * T
            assert nb * T * T < 2 ** 31
            te.forward(windows, start64.to(torch.int32), out=out[b0 * T:(b0 + nb) * T])
        return out.view(B, T, 1).refine_names("batch", "time", "features")

    FORWARD_WINDOWS = 65536  # windows per launch of forward(): 1.9 M rows x F floats of swapped windows at T = 30

    def _forward_engine(self, device, T: int) -> "TFTEngine":
        """The engine (device copy of the flat parameters + its transposed image + a context) forward() runs on, rebuilt
        only when a parameter tensor was replaced, moved or written since the last call (data_ptr / _version)."""
        key = (str(device), T, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        cached = self.__dict__.get("_fwd_engine")
        if cached is None or cached[0] != key:
            cached = (key, TFTEngine(self.flat_parameters().to(device), self.num_features, T, p_drop=self.p_drop, max_windows=1))
            self.__dict__["_fwd_engine"] = cached
        return cached[1]


@device_guarded
class TFTEngine:
    """Device state of one TFT: flat parameters, AdamW moments, work buffers; typed calls into the C ABI."""

    def __init__(self, flat: torch.Tensor, num_features: int, seq_length: int = 30, *, p_drop: float = 0.1, max_windows: int = 32,
                 weight_decay: float = 0.01, beta2: float = 0.999, adam_eps: float = 1e-8, max_norm: float = 1.0):
        self.eng = Engine(flat.device)
        self.lib, self.h = self.eng.lib, self.eng.ctx.handle
        self.F, self.T = num_features, seq_length
        self.n = _lib.check(self.lib.mcmc_tft_param_count(num_features), "mcmc_tft_param_count")
        if flat.numel() != self.n:
            raise ValueError(f"expected {self.n} parameters for {num_features} features, got {flat.numel()}")
        dev = self.eng.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.P = flat.detach().to(**f32).contiguous().clone()
        self.PT = torch.empty_like(self.P)
        self.grad = torch.empty_like(self.P)
        self.adam_m = torch.zeros_like(self.P)
        self.adam_v = torch.zeros_like(self.P)
        self.norm_scratch = torch.empty(self.lib.mcmc_flat_norm_blocks(), dtype=torch.float64, device=dev)
        self.gnorm = torch.empty(1, **f32)
        self.loss = torch.empty(1, **f32)
        self.max_windows = max_windows
        self.dP = torch.empty(max_windows, self.n, **f32)
        self.pred = torch.empty(max_windows, **f32)
        # the train-mode forward leaves every window's activations here for the backward kernel (124 KB per window)
        self.act = torch.empty(max_windows * int(self.lib.mcmc_tft_act_floats()), **f32)
        self.p_drop, self.weight_decay, self.beta2, self.adam_eps, self.max_norm = p_drop, weight_decay, beta2, adam_eps, max_norm
        self.t = 0
        self._pt_stale = True

    def _transpose(self):
        if self._pt_stale:
            _lib.check(self.lib.mcmc_tft_transpose(self.h, _p(self.P), _p(self.PT), self.F, _stream()), "mcmc_tft_transpose")
            self._pt_stale = False

    def _check(self, X: torch.Tensor, start: torch.Tensor):
        if not (X.is_cuda and start.is_cuda):
            raise RuntimeError("X / start must live on the HIP device; mcmc_amd has no CPU path")
        if X.dim() != 2 or X.shape[1] != self.F:
            raise ValueError(f"Expected {self.F} features, got {X.shape[-1]}")
        if start.dtype != torch.int32:
            raise ValueError("start must be int32")

    def forward(self, X: torch.Tensor, start: torch.Tensor, *, masks: Optional[Sequence[torch.Tensor]] = None,
                seed: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Last-timestep prediction of every window X[start[b] : start[b]+T].  Eval mode unless
        ``masks`` (4 uint8 keep-masks in the reference's draw order) or ``seed`` (Philox dropout) is given."""
        self._check(X, start)
        B = self._B = start.numel()
        self._transpose()
        pred = out if out is not None else torch.empty(B, dtype=torch.float32, device=X.device)
        mode, m = self._mask_args(masks, seed)
        _lib.check(self.lib.mcmc_tft_forward(self.h, _p(X), X.shape[0], _p(start), B, self.T, self.F, _p(self.P), _p(self.PT), mode, *m,
                                             (seed or 0) & _U64, self.p_drop, _p(pred), _stream()), "mcmc_tft_forward")
        return pred

    def _mask_args(self, masks, seed):
        if masks is not None:
            ms = [t.to(device=self.eng.device, dtype=torch.uint8).contiguous() for t in masks]
            B, T = self._B, self.T
            for t, want in zip(ms, (B * T * HIDDEN, B * T * HIDDEN, B * HEADS * T * T, B * T * HIDDEN)):
                if t.numel() != want:
                    raise ValueError(f"mask of {t.numel()} elements, expected {want} (reference shapes [B,T,64] / [B*4,T,T])")
            self._keep = ms  # keep alive across the launches
            return 1, [_p(t) for t in ms]
        none = [_p(None)] * 4
        return (2, none) if seed is not None else (0, none)

    def step(self, X: torch.Tensor, start: torch.Tensor, yb: torch.Tensor, lr: float, beta1: float = 0.9, *,
             masks: Optional[Sequence[torch.Tensor]] = None, seed: int = 0, alpha: float = 2.0, beta: float = 0.5,
             update: bool = True, loss_acc: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step (temporal_fusion.py:247-256) on the windows ``start``; returns the loss [1] (device).
        ``loss_acc``: optional fp64 [1] on the device, loss_acc += loss (the epoch sum of :257)."""
        self._check(X, start)
        B = start.numel()
        if B > self.max_windows:
            raise ValueError(f"batch of {B} windows exceeds max_windows = {self.max_windows}")
        yb = yb.to(device=self.eng.device, dtype=torch.float32).contiguous()
        X = X.contiguous()
        B = self._B = start.numel()
        self._transpose()
        st = _stream()
        mode, m = self._mask_args(masks, None if masks is not None else seed)
        pred = self.pred
        _lib.check(self.lib.mcmc_tft_forward_save(self.h, _p(X), X.shape[0], _p(start), B, self.T, self.F, _p(self.P), _p(self.PT), mode, *m,
                                                  seed & _U64, self.p_drop, _p(pred), _p(self.act), st), "mcmc_tft_forward_save")
        if loss_acc is not None and (loss_acc.dtype != torch.float64 or loss_acc.numel() != 1 or not loss_acc.is_cuda):
            raise ValueError("loss_acc must be an fp64 [1] device tensor")
        # the loss (utils.py:7-28 over the batch's predictions) runs in the prologue of the backward kernel: one launch fewer
        _lib.check(self.lib.mcmc_tft_backward_loss(self.h, _p(X), X.shape[0], _p(start), B, self.T, self.F, _p(self.P), _p(self.PT), mode, *m,
                                                   seed & _U64, self.p_drop, _p(pred), _p(yb), alpha, beta, _p(self.loss), _p(loss_acc),
                                                   _p(self.dP), _p(self.act), st), "mcmc_tft_backward_loss")
        if update:
            self.t += 1
            # clip + AdamW; the same kernel refreshes the transposed image PT (it was current: _transpose() above)
            _lib.check(self.lib.mcmc_tft_adamw(self.h, _p(self.P), _p(self.PT), self.F, _p(self.dP), B, _p(self.grad), _p(self.adam_m),
                                               _p(self.adam_v), _p(self.norm_scratch), lr, beta1, self.beta2, self.adam_eps,
                                               self.weight_decay, self.max_norm, self.t, _p(self.gnorm), st), "mcmc_tft_adamw")
        return self.loss


class TFTModel:
    """Drop-in for the reference class (temporal_fusion.py:156-300)."""

    def __init__(self, num_features: int, seq_length: int = 30, batch_size: int = 32, device: str = "cuda", *, seed: Optional[int] = None):
        self.model = TemporalFusionTransformer(num_features=num_features, hidden_size=64, num_heads=4).to(device)
        self.seq_length = seq_length
        self.batch_size = batch_size
        self.device = device
        self._seed = seed
        self.epochs_trained = 0

    def create_sequences(self, X: torch.Tensor, y: torch.Tensor = None) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """temporal_fusion.py:167-203 (materialised windows, for callers that want them; train/predict index X directly)."""
        X = X.refine_names("time", "features").rename(None)
        n = X.shape[0] - self.seq_length + 1
        idx = torch.arange(n, device=X.device).unsqueeze(1) + torch.arange(self.seq_length, device=X.device).unsqueeze(0)
        X_seq = X[idx].refine_names("batch", "time", "features")
        if y is not None:
            y = y.refine_names("time").rename(None)
            return X_seq, y[self.seq_length - 1:].refine_names("batch")
        return X_seq

    def _engine(self) -> TFTEngine:
        dev = torch.device(self.device)
        if dev.type != "cuda":
            raise RuntimeError(f"TFTModel needs a HIP device, got {self.device}; there is no CPU path")
        return TFTEngine(self.model.flat_parameters().to(dev), self.model.num_features, self.seq_length, p_drop=self.model.p_drop,
                         max_windows=self.batch_size)

    def train(self, X: torch.Tensor, y: torch.Tensor, epochs: int = 400):
        """Same optimiser (AdamW 1e-3, wd 0.01), OneCycleLR, clipping, shuffling and patience-20 early
        stopping as temporal_fusion.py:205-271; one fused HIP step per batch."""
        X = X.refine_names("time", "features")
        y = y.refine_names("time")
        if X.size("features") != self.model.num_features:
            raise ValueError(f"Expected {self.model.num_features} features, got {X.size('features')}")
        te = self._engine()
        dev = te.eng.device
        Xd = X.rename(None).to(device=dev, dtype=torch.float32).contiguous()
        yd = y.rename(None).to(device=dev, dtype=torch.float32).contiguous()
        n = Xd.shape[0] - self.seq_length + 1
        if n < 1:
            raise ValueError(f"need at least seq_length = {self.seq_length} rows, got {Xd.shape[0]}")
        bounds = [(b, min(b + self.batch_size, n)) for b in range(0, n, self.batch_size)]
        lrs, b1s = one_cycle_schedule(epochs, len(bounds))
        seed0 = self._seed if self._seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        best_loss, patience, patience_counter, it = float("inf"), 20, 0, 0
        for epoch in range(epochs):
            perm = torch.randperm(n).to(device=dev, dtype=torch.int32)
            targets = yd[perm.long() + (self.seq_length - 1)]
            epoch_loss = torch.zeros(1, dtype=torch.float64, device=dev)
            for (b0, b1) in bounds:
                te.step(Xd, perm[b0:b1], targets[b0:b1], lrs[it], b1s[it], seed=(seed0 + _GOLDEN * (it + 1)) & _U64, loss_acc=epoch_loss)
                it += 1
            self.epochs_trained = epoch + 1
            epoch_loss = float(epoch_loss.item())  # one sync per epoch
            if epoch_loss < best_loss:
                best_loss, patience_counter = epoch_loss, 0
            else:
                patience_counter += 1
            if patience_counter >= patience:
                print(f"Early stopping at epoch {epoch}")
                break
            if (epoch + 1) % 20 == 0:
                print(f"Epoch {epoch + 1}/{epochs}, Loss: {epoch_loss:.4f}")
        self.model.load_flat(te.P)

    def predict(self, X: torch.Tensor) -> np.ndarray:
        """temporal_fusion.py:273-300: one prediction per window, the first seq_length-1 points padded with the first."""
        X = X.refine_names("time", "features")
        if X.size("features") != self.model.num_features:
            raise ValueError(f"Expected {self.model.num_features} features, got {X.size('features')}")
        te = self._engine()
        Xd = X.rename(None).to(device=te.eng.device, dtype=torch.float32).contiguous()
        n = Xd.shape[0] - self.seq_length + 1
        start = torch.arange(n, dtype=torch.int32, device=te.eng.device)
        preds = te.forward(Xd, start).cpu().numpy()
        return np.concatenate([np.full(self.seq_length - 1, preds[0]), preds])
