"""CPU oracle for the Monte Carlo predictive path of the Bayesian ensemble.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mcmc_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and there only as the checker / the timed CPU baseline, never as the product path.

It is a restatement (torch-CPU / numpy, op for op) of the reference's algorithm:

* ``BayesianLinear.forward``    /root/reference/models/bayesian_ensemble.py:33-49
* ``BayesianNetwork.forward``   /root/reference/models/bayesian_ensemble.py:73-92
* ``GPUBayesianEnsemble.predict`` /root/reference/models/bayesian_ensemble.py:212-252
* ``peak_weighted_loss``        /root/reference/utils.py:7-28

Pinning: ``oracle/gen_golden.py`` imports the reference itself (in the build container
only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those vectors bit for bit.  The Philox / histogram-quantile parts have no
counterpart in the reference (the reference draws noise with torch's CPU generator and
computes no quantiles): Philox4x32-10 follows the published Random123 algorithm
(Salmon et al., SC'11) and is pinned by its known-answer vectors; the quantile oracle
is ``torch.quantile`` over the same samples.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HIDDEN = (256, 128, 64)  # bayesian_ensemble.py:53
P_DROP = 0.1             # bayesian_ensemble.py:71


# --------------------------------------------------------------------------- layout
def layer_dims(input_dim: int) -> List[Tuple[int, int]]:
    """(in, out) per BayesianLinear, in forward order (bayesian_ensemble.py:61-70)."""
    dims = [(input_dim, HIDDEN[0])]
    for i in range(len(HIDDEN) - 1):
        dims.append((HIDDEN[i], HIDDEN[i + 1]))
    dims.append((HIDDEN[-1], 1))
    return dims


def param_offsets(input_dim: int) -> List[Tuple[int, int, int, int]]:
    """Flat layout of one member: per layer (w_off, b_off, in, out).

    Order = the order in which the reference draws noise: weight then bias of each
    layer (bayesian_ensemble.py:43-44), layers in forward order.  W is [out, in]
    row-major exactly like ``weight_mu`` (bayesian_ensemble.py:17).
    """
    offs, o = [], 0
    for k_in, k_out in layer_dims(input_dim):
        offs.append((o, o + k_in * k_out, k_in, k_out))
        o += k_in * k_out + k_out
    return offs


def n_params(input_dim: int) -> int:
    return sum(i * o + o for i, o in layer_dims(input_dim))


def pack_members(models: Sequence[torch.nn.Module]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Modules exposing hidden_layers[i].{weight_mu,...} / output_layer.* -> mu[M,P], rho[M,P]."""
    mus, rhos = [], []
    for m in models:
        layers = list(m.hidden_layers) + [m.output_layer]
        mu = torch.cat([torch.cat([l.weight_mu.detach().reshape(-1), l.bias_mu.detach().reshape(-1)]) for l in layers])
        rho = torch.cat([torch.cat([l.weight_rho.detach().reshape(-1), l.bias_rho.detach().reshape(-1)]) for l in layers])
        mus.append(mu.float().cpu())
        rhos.append(rho.float().cpu())
    return torch.stack(mus), torch.stack(rhos)


# --------------------------------------------------------------------------- forward
def forward_pair(X: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor,
                 eps: Optional[torch.Tensor], masks: Optional[Sequence[torch.Tensor]] = None,
                 p_drop: float = P_DROP) -> torch.Tensor:
    """One sampled forward pass of one member over the full batch.

    X [B,F] fp32; mu, rho, eps flat [P] (eps None -> sample=False path).
    masks: optional 3 Bernoulli keep-masks [B,256],[B,128],[B,64] (train mode):
    torch's Dropout computes ``x * mask * (1/(1-p))`` with the scale rounded to fp32.
    Follows bayesian_ensemble.py:42-49 (sampling + F.linear) and :83-92.
    """
    F_in = X.shape[1]
    offs = param_offsets(F_in)
    h = X
    scale = torch.tensor(1.0 / (1.0 - p_drop), dtype=torch.float32)
    for li, (wo, bo, k_in, k_out) in enumerate(offs):
        w_mu = mu[wo:wo + k_in * k_out].view(k_out, k_in)
        b_mu = mu[bo:bo + k_out]
        if eps is not None:
            w = w_mu + eps[wo:wo + k_in * k_out].view(k_out, k_in) * torch.exp(rho[wo:wo + k_in * k_out].view(k_out, k_in))
            b = b_mu + eps[bo:bo + k_out] * torch.exp(rho[bo:bo + k_out])
        else:
            w, b = w_mu, b_mu
        h = F.linear(h, w, b)
        if li < len(offs) - 1:
            h = F.elu(h)
            if masks is not None:
                h = h * masks[li].to(h.dtype) * scale
    return h.squeeze(-1)


def predict_all(X: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """all_predictions [M*S, B] (bayesian_ensemble.py:225-242). eps [M,S,P]."""
    M, S, _ = eps.shape
    out = torch.empty(M * S, X.shape[0], dtype=torch.float32)
    with torch.no_grad():
        for m in range(M):
            for s in range(S):
                out[m * S + s] = forward_pair(X, mu[m], rho[m], eps[m, s])
    return out


def reduce_mean_std(all_predictions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """bayesian_ensemble.py:245-246: mean(0) and unbiased std(0)."""
    return all_predictions.mean(dim=0), all_predictions.std(dim=0)


def reference_rng_eps(seed: int, M: int, S: int, input_dim: int) -> torch.Tensor:
    """eps [M,S,P] exactly as the reference's predict() would draw it after
    ``torch.manual_seed(seed)``: per member, per sample, per layer randn_like(W) then
    randn_like(b) (bayesian_ensemble.py:43-44, loops :225-231) on the CPU generator."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    P = n_params(input_dim)
    eps = torch.empty(M, S, P, dtype=torch.float32)
    offs = param_offsets(input_dim)
    for m in range(M):
        for s in range(S):
            for (wo, bo, k_in, k_out) in offs:
                eps[m, s, wo:wo + k_in * k_out] = torch.randn(k_out, k_in, generator=g).reshape(-1)
                eps[m, s, bo:bo + k_out] = torch.randn(k_out, generator=g)
    return eps


def predict_reference_semantics(X: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor, S: int,
                                generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The reference predict() op sequence with noise drawn on the fly (timed CPU baseline).

    Per pass: randn for W then b per layer -> exp/mul/add -> linear -> elu; stack; mean;
    unbiased std (bayesian_ensemble.py:212-252)."""
    F_in = X.shape[1]
    offs = param_offsets(F_in)
    M = mu.shape[0]
    views = []
    for m in range(M):
        per = []
        for (wo, bo, k_in, k_out) in offs:
            per.append((mu[m, wo:wo + k_in * k_out].view(k_out, k_in), rho[m, wo:wo + k_in * k_out].view(k_out, k_in),
                        mu[m, bo:bo + k_out], rho[m, bo:bo + k_out]))
        views.append(per)
    with torch.no_grad():
        all_preds = []
        for m in range(M):
            batch_preds = []
            for _ in range(S):
                h = X
                for li, (w_mu, w_rho, b_mu, b_rho) in enumerate(views[m]):
                    w = w_mu + torch.randn(w_mu.shape, generator=generator) * torch.exp(w_rho)
                    b = b_mu + torch.randn(b_mu.shape, generator=generator) * torch.exp(b_rho)
                    h = F.linear(h, w, b)
                    if li < len(offs) - 1:
                        h = F.elu(h)
                batch_preds.append(h.squeeze(-1))
            all_preds.append(torch.stack(batch_preds, dim=0))
        ens = torch.stack(all_preds, dim=0)
        allp = ens.reshape(-1, ens.size(-1))
        return allp.mean(dim=0), allp.std(dim=0)


# --------------------------------------------------------------------------- loss
def peak_weighted_loss(y_pred: torch.Tensor, y_true: torch.Tensor, alpha: float = 2.0, beta: float = 0.5) -> torch.Tensor:
    """utils.py:7-28, op for op (autograd-capable)."""
    if y_true.dim() == 1:
        y_true = y_true.view(-1, 1)
    if y_pred.dim() == 1:
        y_pred = y_pred.view(-1, 1)
    base_loss = F.mse_loss(y_pred, y_true, reduction="none")
    mean = y_true.mean()
    std = y_true.std()
    peak_mask = (y_true > (mean + beta * std)).float()
    pred_error = torch.abs(y_pred - y_true)
    error_weight = 1.0 + torch.sigmoid(pred_error - pred_error.mean())
    total_weight = 1.0 + (alpha - 1.0) * peak_mask * error_weight
    return (base_loss * total_weight).mean()


def peak_weighted_loss_closed_form(p: np.ndarray, t: np.ndarray, alpha: float = 2.0, beta: float = 0.5):
    """fp64 closed form of the loss and dL/dp.

    dL/dp_j = (1/N) [ 2 d_j w_j + (alpha-1) sign(d_j) (k_j d_j^2 g'_j - G/N) ],
    g' = g(1-g), G = sum_i k_i d_i^2 g'_i.
    """
    p = p.astype(np.float64).reshape(-1)
    t = t.astype(np.float64).reshape(-1)
    N = p.size
    d = p - t
    m = t.mean()
    s = t.std(ddof=1) if N > 1 else np.nan
    with np.errstate(invalid="ignore"):
        k = (t > m + beta * s).astype(np.float64)
    e = np.abs(d)
    g = 1.0 / (1.0 + np.exp(-(e - e.mean())))
    w = 1.0 + (alpha - 1.0) * k * (1.0 + g)
    L = float(np.mean(d * d * w))
    gp = g * (1.0 - g)
    G = float(np.sum(k * d * d * gp))
    grad = (2.0 * d * w + (alpha - 1.0) * np.sign(d) * (k * d * d * gp - G / N)) / N
    return L, grad


# --------------------------------------------------------------------------- Philox4x32-10
_PH_M0 = np.uint64(0xD2511F53)
_PH_M1 = np.uint64(0xCD9E8D57)
_PH_W0 = np.uint32(0x9E3779B9)
_PH_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox4x32-10 (Random123).  ctr [...,4] uint32, key [...,2] uint32 -> [...,4] uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0 = np.array(key[..., 0], dtype=np.uint32)
    k1 = np.array(key[..., 1], dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PH_M0 * c[0]
            p1 = _PH_M1 * c[2]
            hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
            n0 = (hi1 ^ c[1] ^ k0.astype(np.uint64)) & _MASK32
            n2 = (hi0 ^ c[3] ^ k1.astype(np.uint64)) & _MASK32
            c = [n0, lo1, n2, lo0]
            k0 = (k0 + _PH_W0).astype(np.uint32)
            k1 = (k1 + _PH_W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def _u01(x: np.ndarray) -> np.ndarray:
    """uint32 -> fp32 uniform in (0,1): (2*(x>>9)+1) * 2^-24, exact in fp32."""
    return ((x >> np.uint32(9)).astype(np.float32) * np.float32(2.0) + np.float32(1.0)) * np.float32(2.0 ** -24)


def philox_normals(seed: int, pair: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Standard normals for parameter index ``idx`` of (member,sample) pair ``pair``.

    Counter = (idx>>2, pair, 0x4D434D43 ['MCMC'], 0), key = (seed_lo, seed_hi).  The four
    32-bit outputs make two Box-Muller pairs: (x0,x1)->(z0,z1), (x2,x3)->(z2,z3) with
    r = sqrt(-2 ln u_a), theta = 2 pi u_b, z_even = r cos(theta), z_odd = r sin(theta),
    all in fp32.  Component idx&3 is returned.  Same keying as the device kernel
    (mcmc_amd/csrc/philox.hpp), so a device dump agrees to a few fp32 ulp.
    """
    pair = np.asarray(pair, dtype=np.uint32)
    idx = np.asarray(idx, dtype=np.uint32)
    pair, idx = np.broadcast_arrays(pair, idx)
    ctr = np.stack([idx >> np.uint32(2), pair, np.full_like(idx, 0x4D434D43), np.zeros_like(idx)], axis=-1)
    key = np.stack([np.full_like(idx, seed & 0xFFFFFFFF), np.full_like(idx, (seed >> 32) & 0xFFFFFFFF)], axis=-1)
    x = philox4x32_10(ctr, key)
    comp = idx & np.uint32(3)
    half = (comp >> np.uint32(1)).astype(np.int64)  # which Box-Muller pair
    xa = np.take_along_axis(x, (2 * half)[..., None], axis=-1)[..., 0]
    xb = np.take_along_axis(x, (2 * half + 1)[..., None], axis=-1)[..., 0]
    ua, ub = _u01(xa), _u01(xb)
    r = np.sqrt(np.float32(-2.0) * np.log(ua)).astype(np.float32)
    theta = (np.float32(2.0 * math.pi) * ub).astype(np.float32)
    z = np.where((comp & np.uint32(1)) == 0, r * np.cos(theta).astype(np.float32), r * np.sin(theta).astype(np.float32))
    return z.astype(np.float32)


def philox_eps(seed: int, pair_ids: Sequence[int], P: int) -> np.ndarray:
    """eps [len(pair_ids), P] fp32 for whole pairs."""
    pair = np.asarray(pair_ids, dtype=np.uint32)[:, None]
    idx = np.arange(P, dtype=np.uint32)[None, :]
    return philox_normals(seed, pair, idx)



def philox_keep_masks(seed: int, pair: int, rows: int, p_drop: float = P_DROP, row_begin: int = 0):
    """Dropout keep-masks [rows,256],[rows,128],[rows,64] of one pair from the 'MASK' Philox stream, for the global rows
    row_begin .. row_begin + rows - 1.

    16 bits per element: one Philox block serves eight elements.  elem runs over the 448 hidden activations
    (256 | 128 | 64); counter = (((elem>>4)<<1) | ((elem>>2)&1), pair, 0x4D41534B ['MASK'], row), key = seed;
    slot = 4*((elem>>3)&1) + (elem&3) selects the 16-bit half slot&1 of output word slot>>1;
    keep iff that half < floor(fp32(1-p) * 65536).  Same definition as mcmc_amd/csrc/dropmask.hpp:drop16.
    """
    n_el = sum(HIDDEN)
    row = (np.arange(rows, dtype=np.int64) + row_begin).astype(np.uint32)[:, None]
    elem = np.arange(n_el, dtype=np.uint32)[None, :]
    row, elem = np.broadcast_arrays(row, elem)
    blk = ((elem >> np.uint32(4)) << np.uint32(1)) | ((elem >> np.uint32(2)) & np.uint32(1))
    slot = ((elem >> np.uint32(3)) & np.uint32(1)) * np.uint32(4) + (elem & np.uint32(3))
    ctr = np.stack([blk, np.full_like(elem, pair), np.full_like(elem, 0x4D41534B), row], axis=-1)
    key = np.stack([np.full_like(elem, seed & 0xFFFFFFFF), np.full_like(elem, (seed >> 32) & 0xFFFFFFFF)], axis=-1)
    x = philox4x32_10(ctr, key)
    word = np.take_along_axis(x, (slot >> np.uint32(1)).astype(np.int64)[..., None], axis=-1)[..., 0]
    half = (word >> (np.uint32(16) * (slot & np.uint32(1)))) & np.uint32(0xFFFF)
    thr = np.uint32(int(np.float32(np.float32(1.0) - np.float32(p_drop)) * np.float32(65536.0)))
    keep = (half < thr).astype(np.uint8)
    a, b = HIDDEN[0], HIDDEN[0] + HIDDEN[1]
    return keep[:, :a], keep[:, a:b], keep[:, b:]


# --------------------------------------------------------------------------- training step (row f1)
class TrainOracle:
    """One member's optimisation state, stepped exactly like the reference's inner loop
    (bayesian_ensemble.py:137-150 optimiser + schedule, :165-181 step): S train-mode forwards with
    the given eps / dropout masks, mean over samples, peak_weighted_loss, backward,
    clip_grad_norm_(1.0), torch's own AdamW.  Parameters are kept flat ([P] mu, [P] rho): the
    global-norm clip and the element-wise AdamW do not depend on how parameters are grouped."""

    def __init__(self, mu: torch.Tensor, rho: torch.Tensor, lr: float = 0.001, weight_decay: float = 0.01):
        self.mu = torch.nn.Parameter(mu.detach().clone().float())
        self.rho = torch.nn.Parameter(rho.detach().clone().float())
        self.opt = torch.optim.AdamW([self.mu, self.rho], lr=lr, weight_decay=weight_decay)

    def one_cycle(self, epochs: int, steps_per_epoch: int):
        """The reference's schedule (bayesian_ensemble.py:143-150)."""
        self.sched = torch.optim.lr_scheduler.OneCycleLR(self.opt, max_lr=0.01, epochs=epochs, steps_per_epoch=steps_per_epoch,
                                                         pct_start=0.3, anneal_strategy="cos")
        return self.sched

    def step(self, xb: torch.Tensor, yb: torch.Tensor, eps: torch.Tensor, masks=None, lr: Optional[float] = None,
             beta1: Optional[float] = None, max_norm: float = 1.0):
        """eps [S,P]; masks: None or 3 tensors [S,rows,256/128/64].  Returns (loss, grad_mu, grad_rho,
        total_norm) with the gradients as they were BEFORE clipping."""
        g = self.opt.param_groups[0]
        if lr is not None:
            g["lr"] = lr
        if beta1 is not None:
            g["betas"] = (beta1, g["betas"][1])
        self.opt.zero_grad()
        preds = []
        for s in range(eps.shape[0]):
            mk = None if masks is None else [torch.as_tensor(m[s]) for m in masks]
            preds.append(forward_pair(xb, self.mu, self.rho, eps[s], mk))
        pred_mean = torch.stack(preds, dim=0).mean(0)
        loss = peak_weighted_loss(pred_mean, yb)
        loss.backward()
        gmu, grho = self.mu.grad.detach().clone(), self.rho.grad.detach().clone()
        total = torch.nn.utils.clip_grad_norm_([self.mu, self.rho], max_norm)
        self.opt.step()
        if getattr(self, "sched", None) is not None and lr is None:
            self.sched.step()
        return loss.detach(), gmu, grho, total.detach()

# --------------------------------------------------------------------------- streaming stats + histogram
def hist_edges_from_pilot(n0: int, s1: np.ndarray, s2: np.ndarray, nbins: int, range_sigmas: float):
    """Per-row histogram placement from pilot moments (fp64 in, fp32 lo / inv_w out).

    Interior bins 1..nbins-2 tile [mean-k*std, mean+k*std); bin 0 / nbins-1 catch under- /
    overflow.  Restates mcmc_amd/csrc/reduce.hip:set_edges_kernel."""
    mean = s1 / n0
    var = np.maximum((s2 - s1 * s1 / n0) / max(n0 - 1, 1), 0.0)
    half = range_sigmas * np.sqrt(var)
    tiny = np.maximum(np.abs(mean) * 1e-6, 1e-12)
    half = np.where(half > tiny, half, tiny)
    lo = (mean - half).astype(np.float32)
    inv_w = ((nbins - 2) / (2.0 * half)).astype(np.float32)
    return lo, inv_w


def hist_bin_index(y: np.ndarray, lo: np.ndarray, inv_w: np.ndarray, nbins: int) -> np.ndarray:
    """fp32: t = (y - lo) * inv_w (two roundings); clamp to [-1, nbins-2]; floor; +1."""
    y = y.astype(np.float32)
    t = ((y - lo.astype(np.float32)).astype(np.float32) * inv_w.astype(np.float32)).astype(np.float32)
    t = np.where(np.isnan(t), np.float32(-1.0), t)
    t = np.minimum(np.maximum(t, np.float32(-1.0)), np.float32(nbins - 2))
    return np.floor(t).astype(np.int64) + 1


def hist_accumulate(Y: np.ndarray, lo: np.ndarray, inv_w: np.ndarray, nbins: int) -> np.ndarray:
    """Y [C,B] -> counts [B,nbins] int64."""
    C, B = Y.shape
    idx = hist_bin_index(Y, lo[None, :], inv_w[None, :], nbins)
    hist = np.zeros((B, nbins), dtype=np.int64)
    rows = np.broadcast_to(np.arange(B)[None, :], (C, B))
    np.add.at(hist, (rows.reshape(-1), idx.reshape(-1)), 1)
    return hist


def finalize_from_state(n: int, s1: np.ndarray, s2: np.ndarray, hist: Optional[np.ndarray], lo, inv_w,
                        q_levels: Sequence[float]):
    """(n, sum, sumsq, hist) -> mean, unbiased std, histogram quantiles.

    Quantile rule (restates reduce.hip:finalize_kernel): torch.quantile's 'linear' rule
    interpolates order statistics k = floor(q*(n-1)) and k+1; each order statistic j is
    estimated inside its own bin b (cum[b-1] <= j < cum[b]) as
    edge_lo(b) + (j + 0.5 - cum[b-1]) / hist[b] * width, so every estimate - and their
    interpolation - is within one bin width of the exact value; under-/overflow bins return
    the range edge."""
    mean = s1 / n
    if n > 1:
        var = np.maximum((s2 - s1 * s1 / n) / (n - 1), 0.0)
        std = np.sqrt(var)
    else:
        std = np.full_like(mean, np.nan)
    qs = None
    if hist is not None:
        B, nbins = hist.shape
        cum = np.cumsum(hist, axis=1)
        qs = np.empty((len(q_levels), B), dtype=np.float64)
        width = 1.0 / inv_w.astype(np.float64)
        lo64 = lo.astype(np.float64)
        def order_stat(j):
            b = (cum <= j).sum(axis=1)  # first bin with cum > j
            b = np.minimum(b, nbins - 1)
            prev = np.where(b > 0, np.take_along_axis(cum, np.maximum(b - 1, 0)[:, None], axis=1)[:, 0], 0)
            cnt = np.take_along_axis(hist, b[:, None], axis=1)[:, 0]
            val = lo64 + ((b - 1) + (j + 0.5 - prev) / np.maximum(cnt, 1)) * width
            val = np.where(b == 0, lo64, val)
            return np.where(b == nbins - 1, lo64 + (nbins - 2) * width, val)

        for qi, q in enumerate(q_levels):
            pos = q * (n - 1)
            k0 = math.floor(pos)
            frac = pos - k0
            val = order_stat(float(k0))
            if frac > 0.0 and k0 + 1 < n:
                val = val + frac * (order_stat(float(k0 + 1)) - val)
            qs[qi] = val
    return mean, std, qs
