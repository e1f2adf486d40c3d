"""Host driver of the MC predictive path: chunk loop over (member, sample) pairs.

Replaces the two Python loops + stack + mean/std of ``GPUBayesianEnsemble.predict``
(/root/reference/models/bayesian_ensemble.py:221-246) with

    pack_x -> per chunk { sample_weights -> forward -> reduce_update } -> [RCCL combine] -> finalize

All compute is in libmcmc_hip.so (include/mcmc_hip.h); torch only owns device memory, the
stream and (for world_size > 1) the collective.  There is no CPU path: inputs must live on a
HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_BF16_FAST, DT_F32, NOISE_INJECTED, NOISE_NONE, NOISE_PHILOX

HIDDEN = (256, 128, 64)
_DTYPES = {"fp32": DT_F32, "f32": DT_F32, "float32": DT_F32, torch.float32: DT_F32,
           "bf16": DT_BF16, "bfloat16": DT_BF16, torch.bfloat16: DT_BF16,
           "bf16_fast": DT_BF16_FAST}


def resolve_dtype(dtype) -> int:
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise ValueError(f"dtype must be 'fp32', 'bf16' or 'bf16_fast', got {dtype!r}") from None


def param_count(input_dim: int) -> int:
    """Noise draws per pair (44,289 at input_dim 11; bayesian_ensemble.py:17-22,61-70)."""
    n, k = 0, input_dim
    for h in HIDDEN + (1,):
        n += k * h + h
        k = h
    return n


def pack_members(models: Sequence[torch.nn.Module], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """mu[M,P], rho[M,P] fp32 on ``device`` in the flat layout: per layer W [out,in] row-major
    then b, layers in forward order (the reference's noise draw order, bayesian_ensemble.py:43-44).
    Reads the members through the reference's own attribute names (bayesian_ensemble.py:58-70)."""
    mus, rhos = [], []
    for m in models:
        layers = list(m.hidden_layers) + [m.output_layer]
        mus.append(torch.cat([torch.cat([l.weight_mu.detach().reshape(-1), l.bias_mu.detach().reshape(-1)]) for l in layers]))
        rhos.append(torch.cat([torch.cat([l.weight_rho.detach().reshape(-1), l.bias_rho.detach().reshape(-1)]) for l in layers]))
    mu = torch.stack(mus).to(device=device, dtype=torch.float32).contiguous()
    rho = torch.stack(rhos).to(device=device, dtype=torch.float32).contiguous()
    return mu, rho


def device_guarded(cls):
    """Class decorator: every method of a class that owns an ``eng`` (Engine) runs with that engine's device current.
    The C ABI rejects a context whose device is not the current one and takes the stream of the current device, so an
    object built for cuda:1 must work while cuda:0 is current (main.py:236-238 assigns task i to device i % n)."""
    import functools

    def guard(fn):
        @functools.wraps(fn)
        def wrapped(self, *args, **kw):
            eng = getattr(self, "eng", None)
            dev = eng.device if eng is not None else next((a.device for a in args if isinstance(a, torch.Tensor)), None)
            if dev is None or dev.type != "cuda":
                return fn(self, *args, **kw)  # Engine() names the missing HIP device
            with torch.cuda.device(dev):
                return fn(self, *args, **kw)
        return wrapped

    for name, fn in list(vars(cls).items()):
        if callable(fn) and not isinstance(fn, (staticmethod, classmethod, type)) and (name == "__init__" or not name.startswith("__")):
            setattr(cls, name, guard(fn))
    return cls


def _require_gpu(t: torch.Tensor, what: str) -> int:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a HIP device (got {t.device}); mcmc_amd has no CPU path")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def shard_pairs(n_pairs: int, rank: int, world: int, weights: Optional[Sequence[float]] = None) -> Tuple[int, int]:
    """Contiguous q-range of ``rank``: [begin, end).  Balanced by default; with ``weights`` (one positive number per rank,
    identical on every rank) the shares are proportional to them - MI355X devices sustain clocks up to ~10 % apart under an
    MFMA-dense kernel, and a sharded predict takes as long as its slowest rank.  Results do not depend on the split: noise is
    keyed by the global pair id and the pilot pairs are q < n_pilot whoever owns them."""
    if weights is None:
        base, rem = divmod(n_pairs, world)
        begin = rank * base + min(rank, rem)
        return begin, begin + base + (1 if rank < rem else 0)
    w = [float(x) for x in weights]
    if len(w) != world or not all(x > 0.0 and x == x and x != float("inf") for x in w):
        raise ValueError(f"shard weights must be {world} positive finite numbers, got {list(weights)}")
    total, acc, bounds = sum(w), 0.0, [0]
    for x in w[:-1]:
        acc += x
        bounds.append(min(n_pairs, max(bounds[-1], int(round(n_pairs * acc / total)))))
    bounds.append(n_pairs)
    return bounds[rank], bounds[rank + 1]


def pair_global_ids(begin: int, count: int, M: int, S: int, device=None) -> torch.Tensor:
    """Processing index q -> global pair id g = member*S + sample, member = q % M, sample = q // M."""
    q = torch.arange(begin, begin + count, device=device)
    return (q % M) * S + q // M


@dataclass
class ReduceState:
    """Per-row partial statistics (the payload of the multi-GPU combine)."""
    n: int
    sum: torch.Tensor                    # [B] fp64
    sumsq: torch.Tensor                  # [B] fp64
    hist: Optional[torch.Tensor] = None  # [B, nbins] uint16 counts, stored as int16
    lo: Optional[torch.Tensor] = None    # [B] fp32
    inv_w: Optional[torch.Tensor] = None  # [B] fp32
    nbins: int = 0


def _single_process(group) -> bool:
    """No exchange needed: torch.distributed is not initialised or the group has one rank.
    MCMC_FORCE_COLLECTIVES=1 (test aid) sends a one-rank group through the collectives anyway, so a one-GPU box
    can exercise the RCCL calls (dtypes, split sizes) of the multi-rank path."""
    import torch.distributed as dist
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("MCMC_FORCE_COLLECTIVES") != "1"


def _via_host(t: torch.Tensor, group) -> bool:
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks of ``group``.  RCCL ('nccl') reduces device tensors directly over xGMI.
    Under gloo (CPU tests; rehearsals where several ranks share one GPU) a device tensor goes through host
    memory here, explicitly: gloo's own staging of large device tensors hung when two ranks shared a card."""
    import torch.distributed as dist
    if _via_host(t, group):
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def combine_partials(state: ReduceState, group=None, n_total: Optional[int] = None) -> ReduceState:
    """Sum-combine the partials of all ranks: fp64 moments and the histogram
    (uint16 pairs viewed as int32: no carry crosses a half while total counts stay <= 65,535).
    RCCL over xGMI when the tensors are on GPUs (backend 'nccl'), gloo on CPU tensors in tests.
    ``n_total``: the combined sample count when the caller knows it (mc_predict does: the shards are a pure function of
    rank and world) - saves the count's all-reduce and the host sync that reading it costs."""
    import torch.distributed as dist
    if _single_process(group):
        return state
    moments = torch.stack([state.sum, state.sumsq])
    all_reduce_sum(moments, group)
    state.sum, state.sumsq = moments[0].contiguous(), moments[1].contiguous()
    if n_total is None:
        count = torch.tensor([state.n], dtype=torch.int64, device=state.sum.device)
        all_reduce_sum(count, group)
        n_total = int(count.item())
    state.n = int(n_total)
    if state.hist is not None:
        if state.n > 65535:
            raise ValueError(f"uint16 histogram holds at most 65,535 samples per row (n = {state.n})")
        all_reduce_sum(state.hist.view(torch.int32), group)
    return state


def row_slice(B: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Rows [begin, end) that ``rank`` finalizes, and the padded slice length (same on all ranks)."""
    per = (B + world - 1) // world
    return min(rank * per, B), min((rank + 1) * per, B), per


def scatter_combine(state: ReduceState, group=None, merge=None, n_total: Optional[int] = None) -> Tuple[ReduceState, int, int]:
    """Reduce-scatter over rows: every rank ends up with the fully combined partials
    of its own row slice only.  Built on all_to_all_single of raw bytes + ``merge`` (Engine.reduce_merge =
    mcmc_reduce_merge: the sum of the world's slices in rank order, in the library): the same wire traffic as
    a reduce-scatter, (W-1)/W of the payload per rank, available on RCCL and gloo alike, so the
    ~B*nbins*2-byte histogram crosses xGMI once instead of twice as in an all-reduce.
    Returns (state of the slice, row_begin, row_end); lo / inv_w are sliced locally (identical on
    all ranks by construction: they come from the all-reduced pilot).
    ``merge(sum_parts [W,rows] f64, sumsq_parts, hist_parts [W,rows,nbins] int16 | None, nbins) -> (sum, sumsq, hist)``;
    the CPU tests of the exchange logic pass a numpy stand-in, the product always the HIP kernel."""
    import torch.distributed as dist
    B = state.sum.shape[0]
    if _single_process(group):
        return state, 0, B
    if merge is None:
        raise RuntimeError("scatter_combine needs the library's merge (Engine.reduce_merge); mcmc_amd has no CPU path")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = [row_slice(B, r, world)[:2] for r in range(world)]
    r0, r1 = bounds[rank]
    mine = r1 - r0
    send_rows = [e - b for b, e in bounds]

    def exchange(t: torch.Tensor) -> torch.Tensor:
        # rows travel as raw bytes: no arithmetic happens on the wire, and gloo's all_to_all has no int16
        src = t.contiguous().view(B, -1).view(torch.uint8)
        host = _via_host(t, group)  # gloo rehearsal on a shared GPU only (see all_reduce_sum)
        if host:
            src = src.cpu()
        out = torch.empty(world * mine, src.shape[1], dtype=torch.uint8, device=src.device)
        dist.all_to_all_single(out, src, output_split_sizes=[mine] * world, input_split_sizes=send_rows, group=group)
        if host:
            out = out.to(t.device)
        return out.view(t.dtype).view((world, mine) + tuple(t.shape[1:]))

    if n_total is None:  # the combined sample count, unless the caller knows it (see combine_partials)
        count = torch.tensor([state.n], dtype=torch.int64, device=state.sum.device)
        all_reduce_sum(count, group)
        n_total = int(count.item())
    n = int(n_total)
    if state.hist is not None and n > 65535:
        raise ValueError(f"uint16 histogram holds at most 65,535 samples per row (n = {n})")
    s1, s2, hist = merge(exchange(state.sum), exchange(state.sumsq), exchange(state.hist) if state.hist is not None else None, state.nbins)
    out = ReduceState(n, s1, s2, nbins=state.nbins)
    if state.hist is not None:
        out.hist = hist
        out.lo, out.inv_w = state.lo[r0:r1].contiguous(), state.inv_w[r0:r1].contiguous()
    return out, r0, r1


def gather_rows(parts: Sequence[Optional[torch.Tensor]], B: int, group=None) -> List[Optional[torch.Tensor]]:
    """All-gather per-slice fp32 results ([rows_of_my_slice] or [k, rows_of_my_slice]) back to full
    [B] / [k, B] tensors on every rank: the parts are packed into one [K, per] block so the step is a
    single all_gather_into_tensor."""
    import torch.distributed as dist
    if _single_process(group):
        return list(parts)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1, per = row_slice(B, rank, world)
    present = [t for t in parts if t is not None]
    # the row count of each part comes from its rank, not from reshape(-1, rows): an empty slice (B < world) packs as [k, 0]
    blocks = [(t.reshape(1, r1 - r0) if t.dim() == 1 else t).to(torch.float32) for t in present]
    K = sum(b.shape[0] for b in blocks)
    packed = torch.zeros(K, per, dtype=torch.float32, device=present[0].device)
    if r1 > r0:
        packed[:, :r1 - r0] = torch.cat(blocks, dim=0)
    host = _via_host(packed, group)
    if host:
        packed = packed.cpu()
    gathered = torch.empty(world * K, per, dtype=torch.float32, device=packed.device)  # concatenated along dim 0
    dist.all_gather_into_tensor(gathered, packed, group=group)
    if host:
        gathered = gathered.to(present[0].device)
    full = gathered.view(world, K, per).permute(1, 0, 2).reshape(K, world * per)[:, :B]
    outs, k = [], 0
    for t in parts:
        if t is None:
            outs.append(None)
            continue
        rows = 1 if t.dim() == 1 else t.shape[0]
        blk = full[k:k + rows]
        outs.append(blk[0].contiguous() if t.dim() == 1 else blk.contiguous())
        k += rows
    return outs


class Engine:
    """Thin typed wrapper over the C ABI for one device."""

    def __init__(self, device: torch.device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"mcmc_amd needs a HIP device, got {device}; there is no CPU path")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.ctx = _lib.Context.for_device(self.device.index)
        self.lib = self.ctx.lib

    # ---- piece 1 -------------------------------------------------------------------------
    def pack_x(self, X: torch.Tensor, dtype: int) -> torch.Tensor:
        B, F = X.shape
        nbytes = _lib.check(self.lib.mcmc_xpack_bytes(B, F, dtype), "mcmc_xpack_bytes")
        xp = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mcmc_pack_x(self.ctx.handle, _p(X), B, F, dtype, _p(xp), _stream()), "mcmc_pack_x")
        return xp

    def wpack_bytes(self, F: int, dtype: int) -> int:
        return _lib.check(self.lib.mcmc_wpack_bytes(F, dtype), "mcmc_wpack_bytes")

    def sample_weights(self, mu, rho, S: int, F: int, pair_begin: int, pair_count: int, noise_mode: int,
                       eps: Optional[torch.Tensor], seed: int, dtype: int, wpack: torch.Tensor):
        _lib.check(self.lib.mcmc_sample_weights(self.ctx.handle, _p(mu), _p(rho), mu.shape[0], S, F, pair_begin, pair_count,
                                                noise_mode, _p(eps), seed & 0xFFFFFFFFFFFFFFFF, dtype, _p(wpack), _stream()),
                   "mcmc_sample_weights")

    def philox_eps(self, seed: int, g_begin: int, pair_count: int, F: int) -> torch.Tensor:
        eps = torch.empty(pair_count, param_count(F), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mcmc_philox_eps(self.ctx.handle, seed & 0xFFFFFFFFFFFFFFFF, g_begin, pair_count, F, _p(eps), _stream()),
                   "mcmc_philox_eps")
        return eps

    def forward(self, dtype: int, xpack, B: int, F: int, wpack, pair_count: int, Y: torch.Tensor, ldY: int,
                masks=None, keep_scale: float = 1.0, philox_dropout: Optional[Tuple[int, float, int, int, int]] = None):
        """masks: three host keep-masks [pair_count, B, 256/128/64] uint8 (train-mode forward);
        philox_dropout: (mask_seed, p_drop, n_members, n_samples, pair_begin[, row_begin]) for counter-based masks;
        row_begin = global index of this launch's first row (the masks are keyed by the global row)."""
        if philox_dropout is not None:
            mseed, p_drop, M, S, pair_begin = philox_dropout[:5]
            row_begin = philox_dropout[5] if len(philox_dropout) > 5 else 0
            _lib.check(self.lib.mcmc_forward_dropout(self.ctx.handle, dtype, _p(xpack), B, F, _p(wpack), M, S, pair_begin, pair_count,
                                                     mseed & 0xFFFFFFFFFFFFFFFF, float(p_drop), row_begin, _p(Y), ldY, _stream()),
                       "mcmc_forward_dropout")
            return
        m0 = m1 = m2 = None
        if masks is not None:
            m0, m1, m2 = masks
        _lib.check(self.lib.mcmc_forward(self.ctx.handle, dtype, _p(xpack), B, F, _p(wpack), pair_count, _p(m0), _p(m1), _p(m2),
                                         keep_scale, _p(Y), ldY, _stream()), "mcmc_forward")

    # ---- piece 2 -------------------------------------------------------------------------
    def new_state(self, B: int, nbins: int = 0) -> ReduceState:
        """Zeroed per-row partials: one allocation and one fill, viewed as (sum, sumsq[, lo, inv_w, hist])."""
        Bp = (B + 1) // 2 * 2  # keeps every view 8-byte aligned
        nbytes = 16 * Bp + (8 * Bp + 2 * B * nbins if nbins else 0)
        buf = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        f64 = buf[:16 * Bp].view(torch.float64)
        st = ReduceState(0, f64[:B], f64[Bp:Bp + B])
        if nbins:
            st.nbins = nbins
            f32 = buf[16 * Bp:24 * Bp].view(torch.float32)
            st.lo, st.inv_w = f32[:B], f32[Bp:Bp + B]
            st.hist = buf[24 * Bp:24 * Bp + 2 * B * nbins].view(torch.int16).view(B, nbins)
        return st

    def reduce_update(self, Y: torch.Tensor, ldY: int, n_pairs: int, B: int, st: ReduceState, with_hist: bool = True):
        use_hist = with_hist and st.hist is not None
        _lib.check(self.lib.mcmc_reduce_update(self.ctx.handle, _p(Y), ldY, n_pairs, B, _p(st.sum), _p(st.sumsq),
                                               _p(st.lo if use_hist else None), _p(st.inv_w if use_hist else None),
                                               _p(st.hist if use_hist else None), st.nbins if use_hist else 0, _stream()),
                   "mcmc_reduce_update")
        st.n += n_pairs

    def reduce_merge(self, sum_parts: torch.Tensor, sumsq_parts: torch.Tensor, hist_parts: Optional[torch.Tensor], nbins: int):
        """Sum of the W partial states of one row slice, in rank order (mcmc_reduce_merge)."""
        W, rows = sum_parts.shape
        s1 = torch.empty(rows, dtype=torch.float64, device=self.device)
        s2 = torch.empty(rows, dtype=torch.float64, device=self.device)
        hist = torch.empty(rows, nbins, dtype=torch.int16, device=self.device) if hist_parts is not None else None
        over = torch.zeros(1, dtype=torch.int32, device=self.device) if hist_parts is not None else None
        _lib.check(self.lib.mcmc_reduce_merge(self.ctx.handle, W, rows, _p(sum_parts.contiguous()), _p(sumsq_parts.contiguous()),
                                              _p(hist_parts.contiguous() if hist_parts is not None else None), nbins, _p(s1), _p(s2),
                                              _p(hist), _p(over), _stream()), "mcmc_reduce_merge")
        if over is not None and int(over.item()):
            raise ValueError("a histogram count exceeded 65,535 in the combine (uint16 counts)")
        return s1, s2, hist

    def set_edges(self, B: int, n0: int, psum, psumsq, st: ReduceState, range_sigmas: float):
        _lib.check(self.lib.mcmc_reduce_set_edges(self.ctx.handle, B, n0, _p(psum), _p(psumsq), st.nbins, range_sigmas,
                                                  _p(st.lo), _p(st.inv_w), _stream()), "mcmc_reduce_set_edges")

    def finalize(self, B: int, st: ReduceState, q_levels: Sequence[float] = ()):
        mean = torch.empty(B, dtype=torch.float32, device=self.device)
        std = torch.empty(B, dtype=torch.float32, device=self.device)
        nq = len(q_levels)
        quant = torch.empty(nq, B, dtype=torch.float32, device=self.device) if nq else None
        qarr = (C.c_double * max(nq, 1))(*[float(q) for q in q_levels])
        _lib.check(self.lib.mcmc_reduce_finalize(self.ctx.handle, B, st.n, _p(st.sum), _p(st.sumsq),
                                                 _p(st.hist if nq else None), _p(st.lo if nq else None),
                                                 _p(st.inv_w if nq else None), st.nbins, qarr, nq, _p(mean), _p(std), _p(quant),
                                                 _stream()), "mcmc_reduce_finalize")
        return mean, std, quant


class _timed:
    """Bracket a launch with events on the current (= launch) stream."""

    def __init__(self, timers: Optional[dict], name: str, units: int = 0):
        self.timers, self.name, self.units = timers, name, units

    def __enter__(self):
        if self.timers is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record()

    def __exit__(self, *exc):
        if self.timers is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self.timers.setdefault(self.name, []).append((self.t0, t1, self.units))


@dataclass
class PredictResult:
    mean: torch.Tensor
    std: torch.Tensor
    quantiles: Optional[torch.Tensor] = None       # [n_q, B]
    bin_width: Optional[torch.Tensor] = None       # [B] histogram bin width (quantile resolution)
    samples: Optional[torch.Tensor] = None         # [M*S, B] in the reference's all_predictions order
    n_samples: int = 0


PILOT_PAIRS = 128


def pilot_pairs(cnt: int, M: int) -> int:
    """Pairs whose moments place the histogram: the first ~128 in processing order (whole rounds of members),
    q = 0 .. pilot_pairs-1 whichever rank owns them, so the bin edges - and with them the quantiles - do not
    depend on the world size.  The range is mean +- range_sigmas * std with catch-all end bins, so a 6 % error
    of the pilot's std moves the edges by a fraction of a sigma; reducing a whole chunk twice would cost a full
    extra pass."""
    want = max(M, PILOT_PAIRS // M * M)
    return min(cnt, want)


CHUNK_BYTES = 8 << 30  # default bound on the sample buffer Y (the only intermediate that touches HBM)
ROW_GRANULE = 16384     # the forward kernels' row superblock: row blocks are whole superblocks


def plan_chunks(n_pairs_max: int, B: int, chunk_bytes: int, chunk_pairs: Optional[int], chunk_rows: Optional[int],
                M: int) -> Tuple[int, int]:
    """(pairs per chunk, rows per block) of the sample buffer Y[pairs, rows] fp32 within ``chunk_bytes``.

    Default: ROW blocks that hold every pair of this rank's shard - a row's samples then meet in ONE reduce
    launch, so the per-row state (moments + histogram, 288 B per row at 128 bins) is written once per predict
    instead of read-modify-written once per pair chunk (at 1,024-pair chunks that state pass was 12 % of the
    reduction's bytes).  Only when even one 16,384-row superblock of all pairs exceeds the bound are the pairs
    chunked too.  Derived from the LARGEST shard of the group and constants only, so every rank of a sharded
    predict walks the same row blocks (the pilot all-reduce happens once per block)."""
    Bp = (B + 63) // 64 * 64
    if chunk_pairs:
        cn = max(1, min(chunk_pairs, max(n_pairs_max, 1)))
    else:
        cn = max(1, n_pairs_max)
        if cn * min(Bp, ROW_GRANULE) * 4 > chunk_bytes:
            cn = max(1, chunk_bytes // (min(Bp, ROW_GRANULE) * 4))
            if cn > M:
                cn -= cn % M  # whole rounds of members per chunk
    if chunk_rows:
        rb = max(ROW_GRANULE, chunk_rows // ROW_GRANULE * ROW_GRANULE)
    elif chunk_pairs:
        rb = Bp  # an explicit pair chunking keeps whole rows (the pre-round-2 shape)
    else:
        rb = max(ROW_GRANULE, chunk_bytes // (cn * 4) // ROW_GRANULE * ROW_GRANULE)
    if rb >= B:
        return cn, B
    nb = (B + rb - 1) // rb
    per = (B + nb - 1) // nb  # balanced blocks of whole superblocks
    return cn, (per + ROW_GRANULE - 1) // ROW_GRANULE * ROW_GRANULE


@dataclass(frozen=True)
class RankPlan:
    """What one rank of a (possibly sharded) predict does."""
    q_begin: int          # its contiguous range of processing indices q: [q_begin, q_end)
    q_end: int
    pairs_per_chunk: int  # pairs per forward / reduce launch (Cn): the whole shard unless pairs are chunked
    hoisted: bool         # one pair chunk: the shard's weights are sampled once and serve every row block
    pilot_share: int      # pilot pairs this rank reduces per row block: those of q < n_pilot inside its FIRST chunk

    @property
    def n_local(self) -> int:
        return self.q_end - self.q_begin

    def chunks(self) -> List[Tuple[int, int]]:
        """(first q, count) of every pair chunk, in launch order."""
        return [(b, min(self.pairs_per_chunk, self.q_end - b)) for b in range(self.q_begin, self.q_end, self.pairs_per_chunk)]


@dataclass(frozen=True)
class ShardPlan:
    """The shard / chunk / pilot plan of one predict: a pure function of sizes, world and constants, identical on every
    rank (no rank learns anything from another one to build it)."""
    n_pairs: int                          # N = M * S
    world: int
    chunk_pairs: int                      # Cplan: pairs per chunk as planned for the LARGEST shard
    rows_per_block: int                   # Rb: every rank walks the same row blocks
    row_blocks: Tuple[Tuple[int, int], ...]  # (first row, rows)
    n_pilot: int                          # the pilot pairs are the global q < n_pilot, whoever owns them
    n_pilot_total: int                    # sum of the ranks' pilot shares (== n_pilot: checked by the CPU tests)
    ranks: Tuple[RankPlan, ...]


def plan_predict(N: int, M: int, B: int, world: int = 1, shard_weights: Optional[Sequence[float]] = None, *,
                 chunk_bytes: int = CHUNK_BYTES, chunk_pairs: Optional[int] = None, chunk_rows: Optional[int] = None) -> ShardPlan:
    """Shards (contiguous q-ranges, equal or weighted), pair chunks, row blocks and the pilot's split over the ranks.

    Invariants (tests/test_host_logic.py, world 1..8, equal / weighted / empty shards, chunked and un-chunked pairs):
    the q-ranges tile [0, N) in rank order; every rank gets the same row blocks; the pilot pairs q < n_pilot lie in
    their owner's first pair chunk, so each row block's pilot is complete after the first forward launch of every
    rank; the shares add up to n_pilot, which every rank knows without an exchange."""
    sw = shard_weights if world > 1 else None
    bounds = [shard_pairs(N, r, world, sw) for r in range(world)]
    n_max = max(e - b for b, e in bounds)  # the largest shard: the chunk plan must not differ between ranks
    Cplan, Rb = plan_chunks(n_max, B, chunk_bytes, chunk_pairs, chunk_rows, M)
    # bounded by the pair chunk only when pairs ARE chunked, so that the pilot pairs sit in every owner's first chunk
    # and the choice does not depend on the world size
    n_pilot = min(pilot_pairs(N, M), Cplan if Cplan < n_max else N)
    ranks = []
    for b, e in bounds:
        n_local = e - b
        cn = max(1, min(Cplan, max(n_local, 1)))
        ranks.append(RankPlan(b, e, cn, cn >= n_local, max(0, min(min(Cplan, n_local), n_pilot - b))))
    blocks = tuple((r0, min(Rb, B - r0)) for r0 in range(0, B, max(Rb, 1)))
    return ShardPlan(N, world, Cplan, Rb, blocks, n_pilot, sum(r.pilot_share for r in ranks), tuple(ranks))


def mc_predict(X: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor, num_samples: int, *, dtype="fp32",
               noise: str = "philox", eps: Optional[torch.Tensor] = None, seed: int = 0,
               q_levels: Sequence[float] = (), nbins: int = 128, range_sigmas: float = 6.0,
               chunk_pairs: Optional[int] = None, chunk_rows: Optional[int] = None, chunk_bytes: int = CHUNK_BYTES,
               group=False, shard_weights: Optional[Sequence[float]] = None, return_samples: bool = False,
               timers: Optional[dict] = None, mc_dropout: bool = False, p_drop: float = 0.1,
               mask_seed: Optional[int] = None) -> PredictResult:
    """Monte Carlo predictive statistics of the ensemble over X.

    X [B,F] fp32 on a HIP device; mu, rho [M,P] fp32 (pack_members).  dtype: 'fp32' (the parity path), 'bf16'
    (bf16 MFMA operands, exact ELU) or 'bf16_fast' (bf16 operands, exponent-field ELU: include/mcmc_hip.h).  noise: 'philox'
    (counter-based, keyed by seed and the global pair id: independent of chunking and world
    size), 'injected' (eps [M*S,P] fp32 in the reference's draw order) or 'none' (sample=False).
    ``group=False`` (default): this process computes all M*S pairs, as the reference does - also under
    torch.distributed, where ranks may be predicting different inputs.  ``group=None`` (the default process
    group; backend 'nccl' = RCCL) or a group: the M*S pairs are sharded over its ranks - every rank must
    call with the same X, mu, rho and seed; the partial (sum, sumsq, histogram) are reduce-scattered over
    rows, each rank finalizes its row slice, and the results are all-gathered: every rank returns the full
    result.  ``timers``: optional dict; per kernel name it collects
    (start, end) torch.cuda.Event pairs recorded on the launch stream around each launch.
    ``mc_dropout`` (extension; the reference's predict() runs in eval mode, bayesian_ensemble.py:215-216):
    keep the members' Dropout(p_drop) live, with counter-based Philox masks keyed by (mask_seed or seed;
    global pair id, GLOBAL row, element) applied inside the MFMA kernels' activation stage: like the weight noise, the
    masks do not depend on how the pairs and rows are cut into launches, nor on the world size.
    """
    import torch.distributed as dist
    dev_index = _require_gpu(X, "X")
    eng = Engine(torch.device("cuda", dev_index))
    dt = resolve_dtype(dtype)
    X = X.detach().rename(None).to(torch.float32).contiguous()
    if X.dim() != 2:
        raise ValueError(f"X must be [rows, features], got {tuple(X.shape)}")
    B, F = X.shape
    M, P = mu.shape
    if P != param_count(F):
        raise ValueError(f"Expected {F} input features to give {param_count(F)} parameters per member, got {P}")
    S = int(num_samples)
    if S < 1:
        raise ValueError("num_samples must be >= 1")
    if mc_dropout and dt == DT_BF16_FAST:
        raise ValueError("dtype 'bf16_fast' is the eval-mode predict only (no dropout-mask stage): use 'bf16' with mc_dropout")
    mode = {"philox": NOISE_PHILOX, "injected": NOISE_INJECTED, "none": NOISE_NONE}.get(noise)
    if mode is None:
        raise ValueError(f"noise must be 'philox', 'injected' or 'none', got {noise!r}")
    if mode == NOISE_INJECTED:
        if eps is None or tuple(eps.shape) != (M * S, P):
            raise ValueError(f"injected eps must be [{M * S}, {P}]")
        _require_gpu(eps, "eps")
        eps = eps.to(torch.float32).contiguous()
    N = M * S
    nq = len(q_levels)
    if nq and N > 65535:
        raise ValueError(f"quantile histograms count in uint16: M*S = {N} exceeds 65,535")

    sharded = group is not False and dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if sharded else 1
    rank = dist.get_rank(group) if world > 1 else 0
    plan = plan_predict(N, M, B, world, shard_weights, chunk_bytes=chunk_bytes, chunk_pairs=chunk_pairs, chunk_rows=chunk_rows)
    me = plan.ranks[rank]
    q_begin, q_end, n_local, Cn, Rb = me.q_begin, me.q_end, me.n_local, me.pairs_per_chunk, plan.rows_per_block
    n_pilot, n_pilot_total = plan.n_pilot, plan.n_pilot_total

    with torch.cuda.device(eng.device):
        xpack = eng.pack_x(X, dt)
        wbytes = eng.wpack_bytes(F, dt)
        xrow_bytes = eng.lib.mcmc_xpack_bytes(32, F, dt) // 32
        ldY = (min(Rb, B) + 63) // 64 * 64
        Y = torch.empty(Cn, ldY, dtype=torch.float32, device=eng.device)
        wpack = torch.empty(Cn * wbytes, dtype=torch.uint8, device=eng.device)
        st = eng.new_state(B, nbins if nq else 0)
        pilot = eng.new_state(B) if nq else None
        samples = torch.empty(N, B, dtype=torch.float32, device=eng.device) if return_samples else None
        if return_samples and world > 1:
            raise ValueError("return_samples is a single-process debugging aid")
        hoisted = me.hoisted  # one pair chunk: its weights are sampled once and serve every row block
        if hoisted and n_local:
            with _timed(timers, "sample_weights"):
                eng.sample_weights(mu, rho, S, F, q_begin, n_local, mode, eps, seed, dt, wpack)

        def rows_of(t: torch.Tensor, r0: int, rb: int):
            return t[r0:r0 + rb]

        for r0, rb in plan.row_blocks:
            xblk = xpack[r0 * xrow_bytes:]  # row blocks start on whole superblocks: 16-byte aligned
            blk = ReduceState(0, rows_of(st.sum, r0, rb), rows_of(st.sumsq, r0, rb), nbins=st.nbins)
            if nq:
                blk.hist, blk.lo, blk.inv_w = st.hist[r0:r0 + rb], rows_of(st.lo, r0, rb), rows_of(st.inv_w, r0, rb)
            first = True
            for begin in range(q_begin, q_end, Cn):
                cnt = min(Cn, q_end - begin)
                if not hoisted:
                    with _timed(timers, "sample_weights"):
                        eng.sample_weights(mu, rho, S, F, begin, cnt, mode, eps, seed, dt, wpack)
                with _timed(timers, "forward", cnt * rb):
                    eng.forward(dt, xblk, rb, F, wpack, cnt, Y, ldY,
                                philox_dropout=(seed if mask_seed is None else mask_seed, p_drop, M, S, begin, r0) if mc_dropout else None)
                if nq and first:
                    # pilot: the moments of global pairs q < n_pilot place every row's histogram.  They sit at the head of
                    # the ranks that own them (rank 0 alone when its shard has >= n_pilot pairs) and in their first chunk;
                    # every rank joins the sum, so all ranks bin identically.
                    with _timed(timers, "pilot"):
                        pblk = ReduceState(0, rows_of(pilot.sum, r0, rb), rows_of(pilot.sumsq, r0, rb))
                        mine = me.pilot_share  # = max(0, min(cnt, n_pilot - begin)) in the rank's first chunk
                        if mine:
                            eng.reduce_update(Y, ldY, mine, rb, pblk)
                        pblk = combine_partials(pblk, group, n_total=n_pilot_total)
                        eng.set_edges(rb, max(pblk.n, 1), pblk.sum, pblk.sumsq, blk, range_sigmas)
                first = False
                with _timed(timers, "reduce_update", cnt * rb):
                    eng.reduce_update(Y, ldY, cnt, rb, blk)
                if samples is not None:
                    samples[pair_global_ids(begin, cnt, M, S, eng.device), r0:r0 + rb] = Y[:cnt, :rb]
            if nq and first:  # a rank without pairs still joins the block's pilot combine
                pblk = combine_partials(ReduceState(0, rows_of(pilot.sum, r0, rb), rows_of(pilot.sumsq, r0, rb)), group, n_total=n_pilot_total)
                eng.set_edges(rb, max(pblk.n, 1), pblk.sum, pblk.sumsq, blk, range_sigmas)
        st.n = n_local

        width = (1.0 / st.inv_w) if nq else None
        # exchange step: reduce-scatter over rows -> finalize the own slice -> all-gather the results
        with _timed(timers, "exchange"):
            part, r0, r1 = scatter_combine(st, group, eng.reduce_merge, n_total=N)
        with _timed(timers, "finalize"):
            mean, std, quant = eng.finalize(r1 - r0, part, q_levels)
        with _timed(timers, "gather"):
            mean, std, quant = gather_rows([mean, std, quant], B, group)
    return PredictResult(mean, std, quant, width, samples, part.n)


class PredictPlan:
    """A captured hipGraph of one whole predict for a fixed shape: the launch-bound regime.

    At the reference's own size (365 rows, 5 x 100 pairs) a predict is ~10 launches of a few
    microseconds each, so the Python / launch overhead is most of the wall time.  The plan allocates
    every buffer once, captures  pack_x -> sample_weights (Philox key read from device memory) ->
    forward -> pilot moments -> histogram edges -> reduce -> finalize  into one graph, and a call is
    three small copies (inputs, seed) + one graph launch.  Single chunk, single rank: all M*S pairs
    must fit the sample buffer (use mc_predict for the big configurations).
    Results are views of the plan's static outputs: valid until the next call."""

    def __init__(self, n_rows: int, input_dim: int, n_members: int, num_samples: int, *, device, dtype="fp32",
                 q_levels: Sequence[float] = (), nbins: int = 128, range_sigmas: float = 6.0):
        self.eng = Engine(torch.device(device))
        dev = self.eng.device
        self.B, self.F, self.M, self.S = n_rows, input_dim, n_members, int(num_samples)
        self.N = self.M * self.S
        self.P = param_count(input_dim)
        self.dt = resolve_dtype(dtype)
        self.q_levels, self.nbins, self.range_sigmas = tuple(q_levels), nbins, range_sigmas
        nq = len(self.q_levels)
        if nq and self.N > 65535:
            raise ValueError(f"quantile histograms count in uint16: M*S = {self.N} exceeds 65,535")
        self.ldY = (n_rows + 63) // 64 * 64
        need = self.N * (self.ldY * 4 + self.eng.wpack_bytes(input_dim, self.dt))
        if need > (8 << 30):
            raise ValueError(f"a PredictPlan holds all {self.N} pairs in one chunk ({need >> 20} MiB); use mc_predict")
        with torch.cuda.device(dev):
            self.X = torch.zeros(n_rows, input_dim, dtype=torch.float32, device=dev)
            self.mu = torch.zeros(n_members, self.P, dtype=torch.float32, device=dev)
            self.rho = torch.zeros(n_members, self.P, dtype=torch.float32, device=dev)
            self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
            self.Y = torch.empty(self.N, self.ldY, dtype=torch.float32, device=dev)
            self.wpack = torch.empty(self.N * self.eng.wpack_bytes(input_dim, self.dt), dtype=torch.uint8, device=dev)
            self.state = self.eng.new_state(n_rows, nbins if nq else 0)
            self.pilot = self.eng.new_state(n_rows)
            self.graph = None
            self.out = None

    def _body(self):
        eng, st, pilot = self.eng, self.state, self.pilot
        xpack = eng.pack_x(self.X, self.dt)
        _lib.check(eng.lib.mcmc_sample_weights_seedptr(eng.ctx.handle, _p(self.mu), _p(self.rho), self.M, self.S, self.F, 0, self.N,
                                                       _p(self.seed), self.dt, _p(self.wpack), _stream()), "mcmc_sample_weights_seedptr")
        eng.forward(self.dt, xpack, self.B, self.F, self.wpack, self.N, self.Y, self.ldY)
        for t in (st.sum, st.sumsq, pilot.sum, pilot.sumsq):
            t.zero_()
        st.n = pilot.n = 0
        if self.q_levels:
            st.hist.zero_()
            eng.reduce_update(self.Y, self.ldY, pilot_pairs(self.N, self.M), self.B, pilot)
            eng.set_edges(self.B, pilot.n, pilot.sum, pilot.sumsq, st, self.range_sigmas)
        eng.reduce_update(self.Y, self.ldY, self.N, self.B, st)
        mean, std, quant = eng.finalize(self.B, st, self.q_levels)
        width = (1.0 / st.inv_w) if self.q_levels else None
        return PredictResult(mean, std, quant, width, None, st.n)

    def capture(self):
        with torch.cuda.device(self.eng.device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up outside the capture (lazy module loads, function attributes)
                self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._body()
        return self

    def __call__(self, X: torch.Tensor, mu: torch.Tensor, rho: torch.Tensor, seed: int) -> PredictResult:
        if tuple(X.shape) != (self.B, self.F) or tuple(mu.shape) != (self.M, self.P) or tuple(rho.shape) != (self.M, self.P):
            raise ValueError(f"plan is for X [{self.B},{self.F}], mu/rho [{self.M},{self.P}]")
        if self.graph is None:
            self.capture()
        self.X.copy_(X.rename(None), non_blocking=True)
        self.mu.copy_(mu, non_blocking=True)
        self.rho.copy_(rho, non_blocking=True)
        s = seed & 0xFFFFFFFFFFFFFFFF
        self.seed.fill_(s - (1 << 64) if s >= (1 << 63) else s)  # same 64 bits, as a signed value
        self.graph.replay()
        return self.out
