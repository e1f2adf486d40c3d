"""Host driver of the ensemble training step.

Replaces the reference's sequential member loop and its per-batch torch graph
(/root/reference/models/bayesian_ensemble.py:135-202) with six launches per step that advance
ALL members at once:

    train_sample -> train_forward -> train_loss -> train_backward -> train_update (grad + AdamW)

At the reference's batch size the five calls are one (mcmc_train_step), which also takes the batch as
a strided slice of the epoch's shuffled tensors and keeps the epoch-loss sum on the device.

All arithmetic is in libmcmc_hip.so (include/mcmc_hip.h, mcmc_train_*); torch owns the device
buffers, the batch gather and - on the host, on a parameter-less dummy optimiser - the
OneCycleLR schedule, so lr / beta1 per step are the numbers the reference's scheduler produces.
There is no CPU path.
"""
from __future__ import annotations

import ctypes
import functools
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import NOISE_INJECTED, NOISE_NONE, NOISE_PHILOX, TrainBuffers, TrainHyper
from .engine import HIDDEN, Engine, _p, _stream, all_reduce_sum, device_guarded, param_count

MASK_NONE, MASK_HOST, MASK_PHILOX = 0, 1, 2
_GOLDEN = 0x9E3779B97F4A7C15
_U64 = 0xFFFFFFFFFFFFFFFF


def _dp_world(group) -> int:
    """Ranks the step's rows are sharded over: 1 without torch.distributed or with group=False."""
    import torch.distributed as dist
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


@device_guarded
class EnsembleTrainer:
    """Device state of M members trained in lock-step: parameters, AdamW moments, work buffers.

    ``step`` is one pass of bayesian_ensemble.py:165-181 for every member on its own mini-batch.
    Pair index = member * num_samples + sample keys the Philox noise / mask streams."""

    # The forward kernel's activations are kept for the backward kernel (autograd's saved tensors) while they fit this many
    # bytes: 1.8 KB per pair and row (fp32 kernels; 0.92 KB for the bf16 kernels) - 2.9 MB at the reference's 50 pairs x 32 rows,
    # 9.5 GB at 80 pairs x 65,536 rows, 79 GB (bf16) at 80 pairs x 1,048,576 rows: sized for the 288 GB of an MI355X.
    # Larger batches recompute the forward tile in the backward kernel (the same numbers, ~1.2-1.3x the step time).
    # The bound is the smaller of this constant (constructor argument ``act_max_bytes``) and ACT_FREE_FRACTION of the HBM that
    # is free when the buffer has to grow; an allocation that fails anyway falls back to the recompute path.
    ACT_MAX_BYTES = 128 << 30
    ACT_FREE_FRACTION = 0.5
    # bf16 kernels: cost model of _bf16_row_blocks in rows (a block's fixed cost; one more partial in the gradient sum)
    BF16_BLOCK_COST_ROWS = 256
    BF16_PARTIAL_COST_ROWS = 128

    def __init__(self, mu: torch.Tensor, rho: torch.Tensor, input_dim: int, num_samples: int = 10, *, p_drop: float = 0.1,
                 weight_decay: float = 0.01, beta2: float = 0.999, adam_eps: float = 1e-8, max_norm: float = 1.0,
                 alpha: float = 2.0, beta: float = 0.5, act_max_bytes: Optional[int] = None, compute_dtype: str = "fp32"):
        """``compute_dtype``: "fp32" (default: exact fp32 MFMA, within 2e-5 of the reference's autograd) or "bf16" (opt-in, batches
        of more than one row block only - 2,048 rows and up: the forward / backward GEMMs on bf16 MFMA operands with fp32
        accumulation, fp32 master weights and AdamW; loss within 2e-2, gradients within 2e-2 of each tensor's largest entry)."""
        if compute_dtype not in ("fp32", "bf16"):
            raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self.bf16_row_blocks = 0  # 0: chosen by _bf16_row_blocks; a positive value pins it (measurements)
        self.act_max_bytes = self.ACT_MAX_BYTES if act_max_bytes is None else int(act_max_bytes)
        self.eng = Engine(mu.device)
        dev = self.eng.device
        self.M, self.P = mu.shape
        if self.P != param_count(input_dim):
            raise ValueError(f"mu has {self.P} parameters per member, input_dim {input_dim} needs {param_count(input_dim)}")
        if rho.shape != mu.shape:
            raise ValueError("mu and rho must have the same shape")
        self.F, self.S = input_dim, num_samples
        self.p_drop, self.weight_decay, self.beta2, self.adam_eps, self.max_norm = p_drop, weight_decay, beta2, adam_eps, max_norm
        self.alpha, self.beta = alpha, beta
        f32 = dict(dtype=torch.float32, device=dev)
        self.mu = mu.detach().to(**f32).contiguous().clone()
        self.rho = rho.detach().to(**f32).contiguous().clone()
        self.sigma = torch.empty_like(self.rho)
        pairs = self.M * self.S
        self.eps = torch.empty(pairs, self.P, **f32)
        self.W = torch.empty(pairs, self.P, **f32)
        self.WT = torch.empty(pairs, self.P, **f32)
        self.dW = torch.empty(pairs, self.P, **f32)  # [pairs * row_blocks, P] once a large batch arrives
        self.grad = torch.empty(2, self.M, self.P, **f32)  # one buffer: the data-parallel step all-reduces it in one call
        self.gmu, self.grho = self.grad[0], self.grad[1]
        self.adam_m = torch.zeros(self.M, 2 * self.P, **f32)
        self.adam_v = torch.zeros(self.M, 2 * self.P, **f32)
        self.norm_scratch = torch.empty(self.M * self.eng.lib.mcmc_train_norm_blocks(), dtype=torch.float64, device=dev)
        self.gnorm = torch.empty(self.M, **f32)
        self.loss = torch.empty(self.M, **f32)
        self.t = 0  # AdamW step count (shared: all members start together; frozen members never resume)
        self._Y = self._delta = self._Ybuf = self._dbuf = self._dpbuf = self._bufs = self._act = None
        self._rows = -1
        self._hyper = TrainHyper(p_drop, alpha, beta, beta2, adam_eps, weight_decay, max_norm)
        h = self.eng.ctx.handle
        _lib.check(self.eng.lib.mcmc_train_sigma(h, _p(self.rho), _p(self.sigma), self.rho.numel(), _stream()), "mcmc_train_sigma")

    def _work(self, rows: int):
        """Work buffers for a batch of ``rows`` rows: Y [M*S, rows], delta [M, rows] (views of buffers that only grow,
        so the ragged last batch of an epoch re-allocates nothing) and the saved-activation scratch."""
        pairs = self.M * self.S
        if self._Ybuf is None or self._Ybuf.numel() < pairs * rows:
            self._Ybuf = torch.empty(pairs * rows, dtype=torch.float32, device=self.eng.device)
            self._dbuf = torch.empty(self.M * rows, dtype=torch.float32, device=self.eng.device)
        if rows != self._rows:
            self._rows = rows
            self._Y = self._Ybuf[:pairs * rows].view(pairs, rows)
            self._delta = self._dbuf[:self.M * rows].view(self.M, rows)
        # saved activations for the backward kernel of small batches (the reference's 32 rows: 2.9 MB for 50 pairs)
        need = pairs * int(self.eng.lib.mcmc_train_act_floats(rows))
        self._save_act = 4 * need <= self.act_max_bytes
        if self._save_act and (self._act is None or self._act.numel() < need):
            self._act = None  # release before growing
            free, _ = torch.cuda.mem_get_info(self.eng.device)
            if 4 * need > self.ACT_FREE_FRACTION * free:
                self._save_act = False  # the device is shared (e.g. with mc_predict's 8 GiB sample buffer): recompute instead
            else:
                try:
                    self._act = torch.empty(need, dtype=torch.float32, device=self.eng.device)
                except torch.cuda.OutOfMemoryError:
                    self._save_act = False
        if not self._save_act and self._act is not None and 4 * self._act.numel() > (64 << 20):
            self._act = None  # a batch too large to save: a big buffer of earlier, smaller batches is handed back
        return self._Y, self._delta

    def _bf16_row_blocks(self, rows: int) -> int:
        """Row blocks per pair for the bf16 kernels (one 16-wave workgroup per CU, 64-row tiles; any count is valid): the count
        that minimises (rounds of workgroups over the CUs) x (rows per block + a block's fixed cost: weight staging, the
        write of its dW partial) + the gradient sum's read of one partial per block."""
        if self.bf16_row_blocks:
            return int(self.bf16_row_blocks)
        key = (rows, self.M * self.S)
        if getattr(self, "_nb16_key", None) == key:
            return self._nb16
        cus = torch.cuda.get_device_properties(self.eng.device).multi_processor_count
        best, best_cost = 1, None
        for nb in range(1, 65):
            per = -(-(-(-rows // nb)) // 64) * 64
            if (nb - 1) * per >= rows:
                continue  # would leave empty blocks
            rounds = -(-(self.M * self.S * nb) // cus)
            cost = rounds * (per + self.BF16_BLOCK_COST_ROWS) + self.BF16_PARTIAL_COST_ROWS * nb
            if best_cost is None or cost < best_cost:
                best, best_cost = nb, cost
        self._nb16_key, self._nb16 = key, best
        return best

    def _act_bf16(self, rows: int) -> Optional[torch.Tensor]:
        """Saved-activation buffer of the bf16 variant (0.9 KB per pair and row), under the same bounds as the fp32 one."""
        need = self.M * self.S * int(self.eng.lib.mcmc_train_act_bytes_bf16(rows))
        buf = getattr(self, "_act16", None)
        if need > self.act_max_bytes:
            self._act16 = None
            return None
        if buf is None or buf.numel() < need:
            self._act16 = None
            free, _ = torch.cuda.mem_get_info(self.eng.device)
            if need > self.ACT_FREE_FRACTION * free:
                return None
            try:
                self._act16 = torch.empty(need, dtype=torch.uint8, device=self.eng.device)
            except torch.cuda.OutOfMemoryError:
                return None
        return self._act16

    def _buffers(self) -> TrainBuffers:
        """mcmc_train_buffers of the current allocations (rebuilt when any tensor was re-allocated or re-bound)."""
        key = tuple(t.data_ptr() for t in (self.mu, self.rho, self.sigma, self.eps, self.W, self.WT, self.dW, self.gmu, self.grho,
                                           self.adam_m, self.adam_v, self.norm_scratch, self._Y, self._delta, self.loss, self.gnorm))
        key += (self._act.data_ptr(), self._act.numel()) if self._act is not None else (None, 0)
        if self._bufs is None or self._bufs_key != key:
            self._bufs, self._bufs_key = TrainBuffers(*key), key
        return self._bufs

    def step(self, xb: torch.Tensor, yb: torch.Tensor, lr: float, beta1: float = 0.9, *, noise: str = "philox",
             eps: Optional[torch.Tensor] = None, seed: int = 0, masks: Optional[Sequence[torch.Tensor]] = None,
             mask_seed: Optional[int] = None, dropout: bool = True, active: Optional[torch.Tensor] = None,
             update: bool = True, group=False, global_rows: Optional[int] = None,
             loss_acc: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step of every (active) member.

        xb [M,rows,F], yb [M,rows] fp32 on the device.  noise: 'philox' (keyed by ``seed``),
        'injected' (eps [M,S,P]) or 'none'.  Dropout: host uint8 keep-masks (3 tensors
        [M*S,rows,256/128/64]) when ``masks`` is given, else Philox masks keyed by ``mask_seed``
        (default: seed), or none with dropout=False.  ``active`` [M] uint8 on the device freezes
        members whose flag is 0 (early-stopped; they must not be re-activated later, the AdamW step
        count is shared).  Returns the per-member loss [M] (device tensor, no sync).
        xb / yb may be slices [:, b0:b1] of [M,N,F] / [M,N] tensors (rows contiguous, any member stride).
        ``loss_acc``: optional fp64 [M] on the device, loss_acc += loss (the epoch sum of :181).

        Data-parallel (BASELINE config 5): with torch.distributed initialised and
        ``group`` None (the default process group) or a group, xb / yb are THIS rank's rows of every
        member's batch.  Noise is keyed by (seed, pair) so every rank draws the same weights; the loss statistics (mean / std of y,
        mean |e|, G) are global-batch through two all-reduces of [M,4] fp64 partials, each rank
        back-propagates its rows, and (gmu, grho) are sum-all-reduced (2*M*P fp32 = 354 KB per member)
        before the clip and the AdamW update, which every rank then applies identically.
        ``global_rows``: total rows per member over all ranks (saves the sync that counts them).
        Philox dropout masks are keyed by the rank-local row index (use distinct mask_seed per rank)."""
        lib, h = self.eng.lib, self.eng.ctx.handle
        M, S, F = self.M, self.S, self.F
        if xb.dim() != 3 or xb.shape[0] != M or xb.shape[2] != F:
            raise ValueError(f"xb must be [{M}, rows, {F}], got {tuple(xb.shape)}")
        rows = xb.shape[1]
        if tuple(yb.shape) != (M, rows):
            raise ValueError(f"yb must be [{M}, {rows}], got {tuple(yb.shape)}")
        if not (xb.is_cuda and yb.is_cuda):
            raise RuntimeError("xb / yb must live on the HIP device; mcmc_amd has no CPU path")
        xb, yb = xb.to(torch.float32), yb.to(torch.float32)
        if rows == 1 or xb.stride(2) != 1 or xb.stride(1) != F or yb.stride(1) != 1 or xb.stride(0) < rows * F or yb.stride(0) < rows:
            xb, yb = xb.contiguous(), yb.contiguous()
        if loss_acc is not None and (loss_acc.dtype != torch.float64 or loss_acc.numel() != M or not loss_acc.is_cuda
                                     or not loss_acc.is_contiguous()):
            raise ValueError(f"loss_acc must be a contiguous fp64 [{M}] device tensor")
        noise_mode = {"philox": NOISE_PHILOX, "injected": NOISE_INJECTED, "none": NOISE_NONE}[noise]
        if noise_mode == NOISE_INJECTED:
            if eps is None or eps.numel() != M * S * self.P:
                raise ValueError(f"noise='injected' needs eps [{M},{S},{self.P}]")
            eps = eps.to(device=self.eng.device, dtype=torch.float32).contiguous()
        m0 = m1 = m2 = None
        if masks is not None:
            mask_mode = MASK_HOST
            m0, m1, m2 = (m.to(device=self.eng.device, dtype=torch.uint8).contiguous() for m in masks)
            for m, w in zip((m0, m1, m2), HIDDEN):
                if m.numel() != M * S * rows * w:
                    raise ValueError(f"mask must be [{M * S},{rows},{w}], got {tuple(m.shape)}")
        else:
            mask_mode = MASK_PHILOX if dropout else MASK_NONE
        mseed = (seed if mask_seed is None else mask_seed) & _U64
        Y, delta = self._work(rows)
        st = _stream()
        dp = _dp_world(group) > 1
        if update and not dp and int(lib.mcmc_train_row_blocks(rows)) == 1:  # the reference's batch sizes: one call
            self.t += 1
            _lib.check(lib.mcmc_train_step(h, ctypes.byref(self._buffers()), ctypes.byref(self._hyper), xb.data_ptr(), xb.stride(0),
                                           yb.data_ptr(), yb.stride(0), rows, F, M, S, noise_mode, _p(eps), seed & _U64, mask_mode,
                                           _p(m0), _p(m1), _p(m2), mseed, _p(active), lr, beta1, self.t, _p(loss_acc), st),
                       "mcmc_train_step")
            return self.loss
        if xb.stride(0) != rows * F or yb.stride(0) != rows:
            xb, yb = xb.contiguous(), yb.contiguous()
        _lib.check(lib.mcmc_train_sample(h, _p(self.mu), _p(self.sigma), M, S, F, noise_mode, _p(eps), seed & _U64, _p(self.eps),
                                         _p(self.W), _p(self.WT), st), "mcmc_train_sample")
        nb = int(lib.mcmc_train_row_blocks(rows))  # > 1 for large batches: each pair's rows run on nb workgroups
        bf16 = self.compute_dtype == "bf16" and nb > 1
        if bf16:
            nb = self._bf16_row_blocks(rows)
        save = self._save_act and not bf16  # the forward pass keeps its activations for the backward kernel (else: recomputed there)
        if bf16:
            act16 = self._act_bf16(rows)  # saved bf16 activations (None: the backward kernel recomputes)
            _lib.check(lib.mcmc_train_forward_bf16(h, _p(xb), rows, F, M, S, _p(self.W), mask_mode, _p(m0), _p(m1), _p(m2), mseed,
                                                   self.p_drop, nb, _p(Y), _p(act16), st), "mcmc_train_forward_bf16")
        elif save:
            _lib.check(lib.mcmc_train_forward_save(h, _p(xb), rows, F, M, S, _p(self.WT), mask_mode, _p(m0), _p(m1), _p(m2), mseed,
                                                   self.p_drop, _p(Y), _p(self._act), st), "mcmc_train_forward_save")
        else:
            _lib.check(lib.mcmc_train_forward(h, _p(xb), rows, F, M, S, _p(self.WT), mask_mode, _p(m0), _p(m1), _p(m2), mseed, self.p_drop,
                                              _p(Y), st), "mcmc_train_forward")
        if dp or rows >= 4096:  # large batches: the grid-wide piece-3 kernels instead of one workgroup per member
            self._dp_loss(Y, yb, rows, delta, group if dp else False, global_rows if dp else rows)
        else:
            _lib.check(lib.mcmc_train_loss(h, _p(Y), _p(yb), M, S, rows, self.alpha, self.beta, _p(delta), _p(self.loss), st),
                       "mcmc_train_loss")
        if self.dW.shape[0] < M * S * nb:
            self.dW = torch.empty(M * S * nb, self.P, dtype=torch.float32, device=self.eng.device)
        if bf16:
            _lib.check(lib.mcmc_train_backward_bf16(h, _p(xb), rows, F, M, S, _p(self.W), mask_mode, _p(m0), _p(m1), _p(m2), mseed,
                                                    self.p_drop, _p(delta), nb, _p(self.dW), _p(act16), st), "mcmc_train_backward_bf16")
        elif save:
            _lib.check(lib.mcmc_train_backward_saved(h, _p(xb), rows, F, M, S, _p(self.W), _p(self.WT), mask_mode, _p(m0), _p(m1), _p(m2),
                                                     mseed, self.p_drop, _p(delta), nb, _p(self.dW), _p(self._act), st),
                       "mcmc_train_backward_saved")
        else:
            _lib.check(lib.mcmc_train_backward_blocks(h, _p(xb), rows, F, M, S, _p(self.W), _p(self.WT), mask_mode, _p(m0), _p(m1), _p(m2),
                                                      mseed, self.p_drop, _p(delta), nb, _p(self.dW), st), "mcmc_train_backward_blocks")
        if update and (dp or nb > 1):
            self.t += 1
            _lib.check(lib.mcmc_train_grad(h, M, S, nb, F, _p(self.eps), _p(self.sigma), _p(self.dW), _p(self.gmu), _p(self.grho),
                                           _p(self.norm_scratch), st), "mcmc_train_grad")
            if dp:
                all_reduce_sum(self.grad, group)
            _lib.check(lib.mcmc_train_apply(h, _p(self.mu), _p(self.rho), _p(self.sigma), M, F, _p(self.gmu), _p(self.grho),
                                            _p(self.adam_m), _p(self.adam_v), _p(self.norm_scratch), _p(active), lr, beta1, self.beta2,
                                            self.adam_eps, self.weight_decay, self.max_norm, self.t, _p(self.gnorm), st),
                       "mcmc_train_apply")
        elif update:
            self.t += 1
            _lib.check(lib.mcmc_train_update(h, _p(self.mu), _p(self.rho), _p(self.sigma), M, S, F, _p(self.eps), _p(self.dW),
                                             _p(self.gmu), _p(self.grho), _p(self.adam_m), _p(self.adam_v), _p(self.norm_scratch),
                                             _p(active), lr, beta1, self.beta2, self.adam_eps, self.weight_decay, self.max_norm,
                                             self.t, _p(self.gnorm), st), "mcmc_train_update")
        if loss_acc is not None:
            loss_acc += self.loss.double()
        return self.loss

    def run_batches(self, Xe: torch.Tensor, ye: torch.Tensor, bounds: Sequence[Tuple[int, int]], lrs: Sequence[float],
                    b1s: Sequence[float], seeds: Sequence[int], *, active: Optional[torch.Tensor] = None,
                    loss_acc: Optional[torch.Tensor] = None) -> None:
        """The batches [b0, b1) of an epoch, one ``mcmc_train_step`` call each, addressed in place in the epoch's
        shuffled tensors Xe [M,N,F] / ye [M,N] (contiguous fp32 on the device): ``step(Xe[:, b0:b1], ye[:, b0:b1], ...)``
        with Philox noise and Philox dropout for every batch, without the per-step tensor slicing and argument checks.
        Batches of more than one row block fall back to ``step``."""
        lib, h = self.eng.lib, self.eng.ctx.handle
        M, S, F = self.M, self.S, self.F
        if tuple(Xe.shape[::2]) != (M, F) or tuple(ye.shape) != (M, Xe.shape[1]):
            raise ValueError(f"Xe must be [{M}, N, {F}] and ye [{M}, N], got {tuple(Xe.shape)} / {tuple(ye.shape)}")
        if not (Xe.is_cuda and ye.is_cuda and Xe.is_contiguous() and ye.is_contiguous() and Xe.dtype == ye.dtype == torch.float32):
            raise ValueError("Xe / ye must be contiguous fp32 tensors on the HIP device")
        if not (len(bounds) == len(lrs) == len(b1s) == len(seeds)):
            raise ValueError("bounds, lrs, b1s and seeds must have one entry per batch")
        N = Xe.shape[1]
        if any(not (0 <= b0 < b1 <= N) for b0, b1 in bounds):
            raise ValueError(f"batch bounds outside [0, {N}]")
        if loss_acc is not None and (loss_acc.dtype != torch.float64 or loss_acc.numel() != M or not loss_acc.is_cuda
                                     or not loss_acc.is_contiguous()):
            raise ValueError(f"loss_acc must be a contiguous fp64 [{M}] device tensor")
        max_rows = max(b1 - b0 for b0, b1 in bounds)
        if int(lib.mcmc_train_row_blocks(max_rows)) != 1:
            for (b0, b1), lr, b1m, sd in zip(bounds, lrs, b1s, seeds):
                self.step(Xe[:, b0:b1], ye[:, b0:b1], lr, b1m, seed=sd, active=active, loss_acc=loss_acc)
            return
        self._work(max_rows)  # buffers sized for the largest batch serve every batch of the epoch
        bufs, hyper, st = ctypes.byref(self._buffers()), ctypes.byref(self._hyper), _stream()
        xp, yp, ap, lp = Xe.data_ptr(), ye.data_ptr(), _p(active), _p(loss_acc)
        for (b0, b1), lr, b1m, sd in zip(bounds, lrs, b1s, seeds):
            self.t += 1
            sd &= _U64
            _lib.check(lib.mcmc_train_step(h, bufs, hyper, xp + 4 * b0 * F, N * F, yp + 4 * b0, N, b1 - b0, F, M, S, NOISE_PHILOX, None, sd,
                                           MASK_PHILOX, None, None, None, sd, ap, lr, b1m, self.t, lp, st), "mcmc_train_step")
        self._work(bounds[-1][1] - bounds[-1][0])  # predictions() describes the last batch

    def _dp_loss(self, Y: torch.Tensor, yb: torch.Tensor, rows: int, delta: torch.Tensor, group, global_rows: Optional[int]):
        """peak_weighted_loss with global-batch statistics on the piece-3 kernels (mcmc_loss_*): per member
        p = mean over samples of this rank's rows; partial sums all-reduced between the passes (group False:
        single process, no collective - the path large local batches take)."""
        lib, h, st = self.eng.lib, self.eng.ctx.handle, _stream()
        M, S, dev = self.M, self.S, self.eng.device
        rp = (rows + 3) // 4 * 4  # 16-byte aligned member rows
        if self._dpbuf is None or self._dpbuf.shape[2] != rp:
            self._dpbuf = torch.zeros(3, M, rp, dtype=torch.float32, device=dev)
        pbuf, tbuf, dbuf = self._dpbuf[0], self._dpbuf[1], self._dpbuf[2]
        pbuf[:, :rows] = Y.view(M, S, rows).mean(1)  # bayesian_ensemble.py:173-174
        tbuf[:, :rows] = yb
        acc = torch.zeros(2, M, 4, dtype=torch.float64, device=dev)
        stats, part = acc[0], acc[1]
        stats[:, 3] = rows
        for m in range(M):
            _lib.check(lib.mcmc_loss_stats(h, _p(pbuf[m]), _p(tbuf[m]), rows, _p(stats[m]), st), "mcmc_loss_stats")
        if group is not False:
            all_reduce_sum(stats, group)
        n_total = int(global_rows) if global_rows is not None else int(round(float(stats[0, 3].item())))
        for m in range(M):
            _lib.check(lib.mcmc_loss_forward(h, _p(pbuf[m]), _p(tbuf[m]), rows, n_total, _p(stats[m]), self.alpha, self.beta,
                                             _p(part[m]), st), "mcmc_loss_forward")
        if group is not False:
            all_reduce_sum(part, group)
        self.loss.copy_(part[:, 0] / n_total)
        for m in range(M):
            _lib.check(lib.mcmc_loss_backward(h, _p(pbuf[m]), _p(tbuf[m]), rows, n_total, _p(stats[m]), _p(part[m]), self.alpha,
                                              self.beta, 1.0 / S, _p(dbuf[m]), st), "mcmc_loss_backward")
        delta.copy_(dbuf[:, :rows])

    def predictions(self) -> torch.Tensor:
        """Y [M*S, rows] of the last step's train-mode forwards."""
        return self._Y

    def unpack_into(self, models: Sequence[torch.nn.Module]):
        """Copy (mu, rho) back into modules with the reference's attribute names."""
        with torch.no_grad():
            for m, net in enumerate(models):
                off = 0
                for l in list(net.hidden_layers) + [net.output_layer]:
                    nw, nb = l.weight_mu.numel(), l.bias_mu.numel()
                    l.weight_mu.copy_(self.mu[m, off:off + nw].view_as(l.weight_mu))
                    l.weight_rho.copy_(self.rho[m, off:off + nw].view_as(l.weight_rho))
                    off += nw
                    l.bias_mu.copy_(self.mu[m, off:off + nb])
                    l.bias_rho.copy_(self.rho[m, off:off + nb])
                    off += nb


@functools.lru_cache(maxsize=16)
def _one_cycle_schedule(epochs: int, steps_per_epoch: int) -> Tuple[Tuple[float, ...], Tuple[float, ...]]:
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([dummy], lr=0.001, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.01, epochs=epochs, steps_per_epoch=steps_per_epoch, pct_start=0.3,
                                                anneal_strategy="cos")
    opt.step()  # once, so the scheduler sees an optimiser that has stepped; the schedule depends on the step index alone
    lrs, b1s = [], []
    total = epochs * steps_per_epoch
    g = opt.param_groups[0]
    for i in range(total):
        lrs.append(float(g["lr"]))
        b1s.append(float(g["betas"][0]))
        if i + 1 < total:
            sched.step()
    return tuple(lrs), tuple(b1s)


def one_cycle_schedule(epochs: int, steps_per_epoch: int) -> Tuple[List[float], List[float]]:
    """(lr, beta1) for every step of the reference's AdamW(lr=1e-3) + OneCycleLR(max_lr=0.01,
    pct_start=0.3, cos) (bayesian_ensemble.py:137-150), produced by torch's scheduler itself on a
    throw-away host parameter (cached per (epochs, steps_per_epoch): main.py fits one ensemble per feature subset)."""
    lrs, b1s = _one_cycle_schedule(int(epochs), int(steps_per_epoch))
    return list(lrs), list(b1s)


def fit_ensemble(trainer: EnsembleTrainer, *args, **kw):
    """``_fit_ensemble`` with the trainer's device current (its early-stop kernel and streams need it)."""
    with torch.cuda.device(trainer.eng.device):
        return _fit_ensemble(trainer, *args, **kw)


def _fit_ensemble(trainer: EnsembleTrainer, X: torch.Tensor, y: torch.Tensor, epochs: int = 400, batch_size: int = 32, *,
                 seed: int = 0, patience: int = 25, log=print, generator: Optional[torch.Generator] = None, lookahead: int = 8):
    """The reference's training loop (bayesian_ensemble.py:152-200) for all members at once.

    Every member draws its own shuffle of the rows per epoch (the reference builds one shuffling
    DataLoader and iterates it once per member and epoch, :122-128,160), the ragged last batch is kept
    (drop_last=False), epoch losses are summed per member, and a member stops updating after
    ``patience`` epochs without improvement (:185-193) while the others go on.

    The stream never drains at an epoch boundary: the early-stopping rule runs on the device (it only has
    to produce the kernels' ``active`` flags), the epoch losses come back through pinned memory, and the
    host applies the same rule to the same fp64 numbers up to ``lookahead`` epochs later for the log
    lines, the epoch counts and the end of the loop.  Epochs queued past the point where every member had
    stopped update nothing (all members frozen).  Returns the list of per-member epoch counts actually trained."""
    from collections import deque
    dev = trainer.eng.device
    M = trainer.M
    X = X.rename(None).to(device=dev, dtype=torch.float32)
    y = y.rename(None).to(device=dev, dtype=torch.float32)
    N = X.shape[0]
    bounds = [(b, min(b + batch_size, N)) for b in range(0, N, batch_size)]
    lrs, b1s = one_cycle_schedule(epochs, len(bounds))
    # host copy of the rule: logs, epoch counts, termination
    best = torch.full((M,), float("inf"), dtype=torch.float64)
    bad = torch.zeros(M, dtype=torch.int64)
    alive = torch.ones(M, dtype=torch.bool)
    trained = [0] * M
    # device copy of the rule (mcmc_train_early_stop): the active flags of the next epoch's steps
    d_best = torch.full((M,), float("inf"), dtype=torch.float64, device=dev)
    d_bad = torch.zeros(M, dtype=torch.int64, device=dev)
    active = torch.ones(M, dtype=torch.uint8, device=dev)
    epoch_losses = torch.zeros(epochs, M, dtype=torch.float64, device=dev)  # row e: the sum of epoch e's batch losses
    Xe = torch.empty(M, N, X.shape[1], dtype=torch.float32, device=dev)      # the epoch's shuffled rows, per member
    ye = torch.empty(M, N, dtype=torch.float32, device=dev)
    lib, h = trainer.eng.lib, trainer.eng.ctx.handle
    depth = max(0, min(int(lookahead), (1 << 24) // max(1, M * N)))  # the pinned permutation ring stays below ~1.3 GB for huge N
    slots = depth + 2  # an epoch's pinned slot is re-used only after that epoch has been settled
    perm_pin = torch.empty(slots, M, N, dtype=torch.int64).pin_memory()
    loss_pin = torch.empty(slots, M, dtype=torch.float64).pin_memory()
    pending = deque()  # (epoch, slot, event after the epoch's last copy, AdamW step count after the epoch)

    def settle(epoch: int, el: torch.Tensor) -> bool:
        """bayesian_ensemble.py:183-200 for one finished epoch; False once no member is left."""
        for m in range(M):
            if not alive[m]:
                continue
            trained[m] = epoch + 1
            if el[m] < best[m]:
                best[m], bad[m] = el[m], 0
            else:
                bad[m] += 1
            if bad[m] >= patience:
                alive[m] = False
                log(f"Model {m + 1}/{M}: Early stopping at epoch {epoch}")
            elif (epoch + 1) % 20 == 0:
                log(f"Model {m + 1}/{M}, Epoch {epoch + 1}/{epochs}, Loss: {float(el[m]):.4f}")
        return bool(alive.any())

    def settle_ready(limit: int) -> bool:
        """Settle epochs whose losses have arrived, and the oldest ones beyond ``limit`` queued epochs."""
        while pending and (len(pending) > limit or pending[0][2].query()):
            epoch, slot, ev, t_after = pending.popleft()
            ev.synchronize()
            if not settle(epoch, loss_pin[slot]):
                trainer.t = t_after  # later epochs were queued with every member frozen: they applied no step
                pending.clear()
                return False
        return True

    it = 0
    for epoch in range(epochs):
        slot = epoch % slots
        for m in range(M):  # per-member permutation, drawn on the host generator like DataLoader's sampler
            torch.randperm(N, generator=generator, out=perm_pin[slot, m])
        perm = perm_pin[slot].to(dev, non_blocking=True).view(-1)
        torch.index_select(X, 0, perm, out=Xe.view(M * N, -1))  # Xe = X[perm]: [M, N, F]
        torch.index_select(y, 0, perm, out=ye.view(M * N))      # ye = y[perm]: [M, N]
        epoch_loss = epoch_losses[epoch]
        nb = len(bounds)
        trainer.run_batches(Xe, ye, bounds, lrs[it:it + nb], b1s[it:it + nb], [seed + _GOLDEN * (k + 1) for k in range(it, it + nb)],
                            active=active, loss_acc=epoch_loss)
        it += nb
        _lib.check(lib.mcmc_train_early_stop(h, _p(epoch_loss), _p(d_best), _p(d_bad), _p(active), M, int(patience), _stream()),
                   "mcmc_train_early_stop")
        loss_pin[slot].copy_(epoch_loss, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((epoch, slot, ev, trainer.t))
        if not settle_ready(depth):
            break
    settle_ready(0)
    return trained
