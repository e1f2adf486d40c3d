"""peak_weighted_loss on the fused HIP kernels (mirror of /root/reference/utils.py:7-28).

Same name, argument meaning and defaults as the reference function; differentiable w.r.t.
``y_pred`` through the closed-form backward.  ``y_true`` gets no gradient
(the reference's callers never ask for one: bayesian_ensemble.py:176, temporal_fusion.py:251).
With ``group`` set (torch.distributed, data-parallel over rows) the statistics are global-batch
like the reference's: three tiny all-reduces (4 + 2 scalars) around the passes.
"""
from __future__ import annotations


import torch

from . import _lib
from .engine import Engine, _p, _require_gpu, _stream, all_reduce_sum


def _is_dist() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _allreduce(t: torch.Tensor, group):
    """group False: local statistics (reference semantics); None / a group: global-batch."""
    if group is False or not _is_dist():
        return
    all_reduce_sum(t, group)


_WS = {}


def _workspace(eng: Engine) -> torch.Tensor:
    """Workspace of mcmc_loss_fused (40 KB) per (device, stream): calls on one stream run in order, so one buffer serves
    them all; calls on different streams may overlap and get their own."""
    key = (eng.device.index, torch.cuda.current_stream(eng.device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(int(eng.lib.mcmc_loss_workspace_bytes()), dtype=torch.uint8, device=eng.device)
    return ws


class _PeakWeightedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, alpha, beta, group):
        dev = _require_gpu(y_pred, "y_pred")
        _require_gpu(y_true, "y_true")
        eng = Engine(torch.device("cuda", dev))
        p = y_pred.detach().to(torch.float32).contiguous().view(-1)
        t = y_true.detach().to(torch.float32).contiguous().view(-1)
        if p.data_ptr() % 16:
            p = p.clone()
        if t.data_ptr() % 16:
            t = t.clone()
        if p.numel() != t.numel():
            raise ValueError(f"y_pred has {p.numel()} elements, y_true {t.numel()}")
        n = p.numel()
        if (group is False or not _is_dist()) and n >= 1:
            # local statistics (the reference's semantics): one call, loss and gradient together when one will be asked for
            want_grad = bool(ctx.needs_input_grad[0])
            with torch.cuda.device(eng.device):
                ws = _workspace(eng)
                loss = torch.empty(1, dtype=torch.float32, device=eng.device)
                grad = torch.empty_like(p) if want_grad else None
                _lib.check(eng.lib.mcmc_loss_fused(eng.ctx.handle, _p(p), _p(t), n, float(alpha), float(beta), 1.0, _p(loss), _p(grad),
                                                   _p(ws), _stream()), "mcmc_loss_fused")
            ctx.fused, ctx.fused_grad = True, grad
            ctx.meta = (eng, n, n, float(alpha), float(beta), y_pred.shape, y_pred.dtype)
            return loss[0]
        ctx.fused, ctx.fused_grad = False, None
        with torch.cuda.device(eng.device):
            stats = torch.zeros(4, dtype=torch.float64, device=eng.device)
            stats[3] = n
            _lib.check(eng.lib.mcmc_loss_stats(eng.ctx.handle, _p(p), _p(t), n, _p(stats), _stream()), "mcmc_loss_stats")
            _allreduce(stats, group)
            n_total = int(stats[3].item()) if group is not False and _is_dist() else n
            part = torch.zeros(4, dtype=torch.float64, device=eng.device)
            _lib.check(eng.lib.mcmc_loss_forward(eng.ctx.handle, _p(p), _p(t), n, n_total, _p(stats), float(alpha), float(beta),
                                                 _p(part), _stream()), "mcmc_loss_forward")
            _allreduce(part, group)
            loss = (part[0] / n_total).to(torch.float32)
        ctx.save_for_backward(p, t, stats, part)
        ctx.meta = (eng, n, n_total, float(alpha), float(beta), y_pred.shape, y_pred.dtype)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        eng, n, n_total, alpha, beta, shape, dtype = ctx.meta
        if ctx.fused:  # the one-call path computed the gradient with the loss (or none was asked for: y_true gets no gradient)
            g = ctx.fused_grad
            return (None if g is None else (g * grad_out.to(torch.float32)).view(shape).to(dtype)), None, None, None, None
        p, t, stats, part = ctx.saved_tensors
        with torch.cuda.device(eng.device):
            grad = torch.empty_like(p)
            # grad_out is a device scalar: fold it in afterwards to avoid a host sync
            _lib.check(eng.lib.mcmc_loss_backward(eng.ctx.handle, _p(p), _p(t), n, n_total, _p(stats), _p(part), alpha, beta, 1.0,
                                                  _p(grad), _stream()), "mcmc_loss_backward")
            grad = (grad * grad_out.to(torch.float32)).view(shape).to(dtype)
        return grad, None, None, None, None


def peak_weighted_loss(y_pred, y_true, alpha=2.0, beta=0.5, *, group=False):
    """Custom loss function for handling demand spikes (utils.py:7).

    ``group=False`` (default): statistics over this call's tensors, as the reference.
    ``group=None`` / a process group: rows are sharded over ranks; statistics are global."""
    y_pred = y_pred.rename(None) if y_pred.has_names() else y_pred
    y_true = y_true.rename(None) if y_true.has_names() else y_true
    return _PeakWeightedLoss.apply(y_pred, y_true, alpha, beta, group)
