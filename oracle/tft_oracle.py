"""CPU oracle of the Temporal Fusion Transformer leg.

TEST INFRASTRUCTURE ONLY - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import this package; the product (mcmc_amd/) never does.

Restates /root/reference/models/temporal_fusion.py op for op on explicit parameters:
  GatedResidualNetwork.forward   :62-89
  TemporalFusionTransformer      :91-153  (nn.MultiheadAttention(64, 4, dropout=0.1), need_weights
                                 path of torch's multi_head_attention_forward: q scaled by
                                 sqrt(1/head_dim) before q k^T, softmax, dropout, attn v, out_proj)
  TFTModel.create_sequences      :167-203
  TFTModel.train inner step      :242-256 (last-timestep prediction, peak_weighted_loss,
                                 clip_grad_norm_(1.0), AdamW, OneCycleLR)
  TFTModel.predict               :273-300 (first seq_length-1 points padded with the first prediction)
Pinned by tests/golden/tft_*.npz, which oracle/gen_golden.py records from the reference itself.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .bnn_oracle import peak_weighted_loss

HIDDEN = 64      # temporal_fusion.py:160
HEADS = 4        # :161
SEQ_LEN = 30     # :157
P_DROP = 0.1     # :92


def param_spec(num_features: int) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) in the reference module's named_parameters() order - the flat layout."""
    H = HIDDEN
    spec = [("input_projection.module.weight", (H, num_features)), ("input_projection.module.bias", (H,))]

    def grn(prefix):
        return [(f"{prefix}.fc1.module.weight", (H, H)), (f"{prefix}.fc1.module.bias", (H,)),
                (f"{prefix}.fc2.module.weight", (H, H)), (f"{prefix}.fc2.module.bias", (H,)),
                (f"{prefix}.gate.module.weight", (H, 2 * H)), (f"{prefix}.gate.module.bias", (H,)),
                (f"{prefix}.layer_norm.weight", (H,)), (f"{prefix}.layer_norm.bias", (H,))]

    spec += grn("feature_grn") + grn("temporal_grn")
    spec += [("attention.in_proj_weight", (3 * H, H)), ("attention.in_proj_bias", (3 * H,)),
             ("attention.out_proj.weight", (H, H)), ("attention.out_proj.bias", (H,))]
    spec += grn("final_grn")
    spec += [("output_projection.module.weight", (1, H)), ("output_projection.module.bias", (1,))]
    return spec


def n_params(num_features: int) -> int:
    return sum(int(np.prod(s)) for _, s in param_spec(num_features))


def flatten(named: Dict[str, torch.Tensor], num_features: int) -> torch.Tensor:
    return torch.cat([named[n].detach().reshape(-1).float() for n, _ in param_spec(num_features)])


def unflatten(flat: torch.Tensor, num_features: int) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for n, s in param_spec(num_features):
        k = int(np.prod(s))
        out[n] = flat[off:off + k].view(s)
        off += k
    return out


def create_sequences(X: torch.Tensor, y: Optional[torch.Tensor] = None, seq_length: int = SEQ_LEN):
    """temporal_fusion.py:167-203: sliding windows; the target is the window's last step."""
    n = X.shape[0] - seq_length + 1
    Xs = torch.stack([X[i:i + seq_length] for i in range(n)], dim=0)
    if y is None:
        return Xs
    return Xs, torch.stack([y[i + seq_length - 1] for i in range(n)], dim=0)


def _grn(x, p, prefix, mask):
    """:62-89 with input_size == output_size (no skip projection)."""
    h = F.elu(F.linear(x, p[f"{prefix}.fc1.module.weight"], p[f"{prefix}.fc1.module.bias"]))
    if mask is not None:
        h = h * mask.to(h.dtype) * torch.tensor(1.0 / (1.0 - P_DROP), dtype=torch.float32)
    h = F.elu(F.linear(h, p[f"{prefix}.fc2.module.weight"], p[f"{prefix}.fc2.module.bias"]))
    gate = torch.sigmoid(F.linear(torch.cat([x, h], dim=-1), p[f"{prefix}.gate.module.weight"], p[f"{prefix}.gate.module.bias"]))
    w = gate * h + (1 - gate) * x
    return F.layer_norm(w, (HIDDEN,), p[f"{prefix}.layer_norm.weight"], p[f"{prefix}.layer_norm.bias"])


def forward(x: torch.Tensor, p: Dict[str, torch.Tensor], masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """x [B,T,F] -> [B,T,1] (:120-153).  masks (train mode): keep-masks in the reference's draw
    order: feature_grn [B,T,64], temporal_grn [B,T,64], attention [B*heads,T,T], final_grn [B,T,64]."""
    B, T, _ = x.shape
    m = masks if masks is not None else (None, None, None, None)
    h = F.linear(x, p["input_projection.module.weight"], p["input_projection.module.bias"])
    h = _grn(h, p, "feature_grn", m[0])
    h = _grn(h, p, "temporal_grn", m[1])
    # nn.MultiheadAttention on [T,B,H] (:139-143)
    d = HIDDEN // HEADS
    qkv = F.linear(h.permute(1, 0, 2), p["attention.in_proj_weight"], p["attention.in_proj_bias"])  # [T,B,3H]
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(T, B * HEADS, d).transpose(0, 1)
    k = k.reshape(T, B * HEADS, d).transpose(0, 1)
    v = v.reshape(T, B * HEADS, d).transpose(0, 1)
    a = torch.softmax(torch.bmm(q * math.sqrt(1.0 / float(d)), k.transpose(-2, -1)), dim=-1)
    if m[2] is not None:
        a = a * m[2].to(a.dtype) * torch.tensor(1.0 / (1.0 - P_DROP), dtype=torch.float32)
    o = torch.bmm(a, v).transpose(0, 1).contiguous().view(T * B, HIDDEN)
    o = F.linear(o, p["attention.out_proj.weight"], p["attention.out_proj.bias"]).view(T, B, HIDDEN).permute(1, 0, 2)
    h = _grn(o, p, "final_grn", m[3])
    return F.linear(h, p["output_projection.module.weight"], p["output_projection.module.bias"])


def predict(X: torch.Tensor, p: Dict[str, torch.Tensor], batch_size: int = 32, seq_length: int = SEQ_LEN) -> np.ndarray:
    """:273-300."""
    Xs = create_sequences(X, None, seq_length)
    preds = []
    with torch.no_grad():
        for i in range(0, Xs.shape[0], batch_size):
            preds.extend(forward(Xs[i:i + batch_size], p)[:, -1, 0].numpy())
    return np.concatenate([np.full(seq_length - 1, preds[0]), preds])


class TFTTrainOracle:
    """One optimisation step like :242-256 on a flat parameter vector with torch's own AdamW."""

    def __init__(self, flat: torch.Tensor, num_features: int, lr: float = 0.001, weight_decay: float = 0.01):
        self.F = num_features
        self.flat = torch.nn.Parameter(flat.detach().clone().float())
        self.opt = torch.optim.AdamW([self.flat], lr=lr, weight_decay=weight_decay)

    def step(self, xb, yb, masks=None, lr=None, beta1=None, max_norm: float = 1.0):
        g = self.opt.param_groups[0]
        if lr is not None:
            g["lr"] = lr
        if beta1 is not None:
            g["betas"] = (beta1, g["betas"][1])
        self.opt.zero_grad()
        out = forward(xb, unflatten(self.flat, self.F), masks)
        pred = out[:, -1, 0]
        loss = peak_weighted_loss(pred, yb)
        loss.backward()
        grad = self.flat.grad.detach().clone()
        total = torch.nn.utils.clip_grad_norm_([self.flat], max_norm)
        self.opt.step()
        return loss.detach(), grad, total.detach(), pred.detach()


def philox_masks(seed: int, keys: Sequence[int], T: int, p_drop: float = P_DROP):
    """The device's Philox dropout decisions as reference-shaped keep-masks for windows keyed ``keys``.

    Stream 'MASK' (mcmc_amd/csrc/philox.hpp:philox_mask_uniform), counter (elem>>2, key, 'MASK', row),
    component elem&3; elem = grn_index*64 + feature for the three GRNs (feature 0, temporal 1, final 2),
    192 + head*32 + key_position for the attention row of the last query.  Entries the device never
    draws (attention rows of other queries) are 1."""
    from .bnn_oracle import _u01, philox4x32_10
    B = len(keys)

    def uniform(key, row, elem):
        row, elem = np.broadcast_arrays(np.asarray(row, dtype=np.uint32), np.asarray(elem, dtype=np.uint32))
        ctr = np.stack([elem >> np.uint32(2), np.full_like(elem, key), np.full_like(elem, 0x4D41534B), row], axis=-1)
        k = np.stack([np.full_like(elem, seed & 0xFFFFFFFF), np.full_like(elem, (seed >> 32) & 0xFFFFFFFF)], axis=-1)
        x = philox4x32_10(ctr, k)
        return _u01(np.take_along_axis(x, (elem & np.uint32(3)).astype(np.int64)[..., None], axis=-1)[..., 0])

    keep = np.float32(1.0 - p_drop)
    g = [np.ones((B, T, HIDDEN), dtype=np.uint8) for _ in range(3)]
    at = np.ones((B * HEADS, T, T), dtype=np.uint8)
    rows = np.arange(T)[:, None]
    feats = np.arange(HIDDEN)[None, :]
    for b, key in enumerate(keys):
        for which in range(3):
            g[which][b] = (uniform(key, rows, which * HIDDEN + feats) < keep).astype(np.uint8)
        for h in range(HEADS):
            at[b * HEADS + h, T - 1, :] = (uniform(key, T - 1, 3 * HIDDEN + h * 32 + np.arange(T)) < keep).astype(np.uint8)
    return [torch.from_numpy(g[0]), torch.from_numpy(g[1]), torch.from_numpy(at), torch.from_numpy(g[2])]
