"""CPU-only tests: C-ABI surface, host-side mirror of the reference interface, pair sharding,
the world_size-2 combine (gloo), and the batched generator.  No device compute is called."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import bnn_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --------------------------------------------------------------------------- C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcmc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mcmc_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 17
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_ctypes_structs_and_arities_follow_the_header():
    """The ctypes mirrors of the header's structs list the same fields in the same order with the same C types, and every
    bound signature has as many arguments as its declaration."""
    import ctypes
    from mcmc_amd import _lib
    text = open(os.path.join(ROOT, "include", "mcmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for cname, mirror in (("mcmc_train_buffers", _lib.TrainBuffers), ("mcmc_train_hyper", _lib.TrainHyper)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        fields = []
        for decl in filter(None, (d.strip() for d in body.split(";"))):
            ctype, names = re.match(r"((?:const\s+)?[a-z0-9_]+\s*\*?)\s*(.*)", decl).groups()
            for nm in names.split(","):
                nm = nm.strip()
                ptr = "*" in ctype or nm.startswith("*")
                base = ctype.replace("*", "").replace("const", "").strip()
                fields.append((nm.lstrip("* "), ctypes.c_void_p if ptr else {"float": ctypes.c_float, "int64_t": ctypes.c_int64,
                                                                              "int": ctypes.c_int}[base]))
        assert [(n, t) for n, t in mirror._fields_] == fields, cname
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        decl = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, text, flags=re.S).group(1).strip()
        n_args = 0 if decl in ("", "void") else decl.count(",") + 1
        assert len(argtypes) == n_args, (name, len(argtypes), n_args)


def test_layout_queries_without_gpu():
    from mcmc_amd import _lib
    from mcmc_amd.engine import param_count
    lib = _lib.load()
    assert lib.mcmc_abi_version() == _lib.ABI_VERSION
    for F in (1, 7, 9, 11, 15, 16, 24):
        assert lib.mcmc_param_count(F) == param_count(F) == orc.n_params(F)
        assert lib.mcmc_wpack_bytes(F, _lib.DT_BF16) % 256 == 0
        assert lib.mcmc_wpack_bytes(F, _lib.DT_F32) % 256 == 0
    assert lib.mcmc_wpack_bytes(11, _lib.DT_BF16) == 91392
    assert lib.mcmc_wpack_bytes(11, _lib.DT_BF16_FAST) == 95232  # + the layer-3 A tile, - the fp32 w3
    assert lib.mcmc_xpack_bytes(365, 11, _lib.DT_BF16_FAST) == 384 * 16 * 2
    assert lib.mcmc_xpack_bytes(365, 11, _lib.DT_BF16) == 384 * 16 * 2
    assert lib.mcmc_xpack_bytes(365, 11, _lib.DT_F32) == 384 * 12 * 4
    # errors: negative code + message, no exception across the boundary
    assert lib.mcmc_param_count(0) == -1
    assert lib.mcmc_wpack_bytes(25, _lib.DT_BF16) == -1
    assert b"input_dim" in lib.mcmc_last_error()
    assert lib.mcmc_wpack_bytes(11, 7) == -1
    with pytest.raises(ValueError):
        _lib.check(lib.mcmc_xpack_bytes(-1, 11, 0), "mcmc_xpack_bytes")


def test_context_needs_a_gpu():
    from mcmc_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises((_lib.MCMCError, ValueError)):
        _lib.Context(0)


# --------------------------------------------------------------------------- mirror of the reference interface
def test_ensemble_constructor_and_errors_match_reference():
    from mcmc_amd import BayesianLinear, BayesianNetwork, GPUBayesianEnsemble
    ens = GPUBayesianEnsemble(input_dim=11, device=torch.device("cpu"), batch_size=32)
    assert (ens.input_dim, ens.n_models, ens.batch_size, ens.models) == (11, 5, 32, [])
    assert isinstance(ens, torch.nn.Module)
    with pytest.raises(ValueError, match="Input dimension mismatch. Expected 11, got 7"):
        ens.train(torch.zeros(20, 7), torch.zeros(20))  # bayesian_ensemble.py:115-118
    net = BayesianNetwork(11)
    assert sum(p.numel() for p in net.parameters()) == 88578
    assert [tuple(l.weight_mu.shape) for l in net.hidden_layers] == [(256, 11), (128, 256), (64, 128)]
    assert tuple(net.output_layer.weight_rho.shape) == (1, 64)
    assert float(net.hidden_layers[0].weight_rho[0, 0]) == -3.0 and float(net.output_layer.bias_mu[0]) == 0.0
    with pytest.raises(ValueError, match="Expected 11 input features, got 5"):
        net(torch.zeros(3, 5))
    with pytest.raises(ValueError, match="Expected 256 input features"):
        BayesianLinear(256, 128)(torch.zeros(2, 3))
    out = net.eval()(torch.zeros(4, 11), sample=False)
    assert out.names == ("batch",) and out.shape == (4,)
    # same init distribution as the reference: identical under the same seed
    sys.path.insert(0, "/root/reference")
    try:
        from models.bayesian_ensemble import BayesianNetwork as RefNet  # only when the reference is mounted
    except Exception:
        RefNet = None
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "utils"]:
            del sys.modules[k]
    if RefNet is not None:
        torch.manual_seed(3)
        a = RefNet(11)
        torch.manual_seed(3)
        b = BayesianNetwork(11)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb)


def test_predict_has_no_cpu_path():
    from mcmc_amd import BayesianNetwork, GPUBayesianEnsemble, mc_predict, peak_weighted_loss
    ens = GPUBayesianEnsemble(input_dim=11, device=torch.device("cpu"))
    ens.models = [BayesianNetwork(11)]
    with pytest.raises(RuntimeError, match="no CPU path"):
        ens.predict(torch.zeros(5, 11))
    with pytest.raises(RuntimeError, match="no CPU path"):
        mc_predict(torch.zeros(5, 11), torch.zeros(1, 44289), torch.zeros(1, 44289), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        peak_weighted_loss(torch.zeros(4), torch.zeros(4))
    with pytest.raises(RuntimeError, match="HIP device"):
        ens.train(torch.zeros(20, 11), torch.zeros(20))
    empty = GPUBayesianEnsemble(input_dim=11, device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no trained members"):
        empty.predict(torch.zeros(5, 11))


def test_pack_members_layout_matches_oracle():
    from mcmc_amd import BayesianNetwork, pack_members
    torch.manual_seed(0)
    nets = [BayesianNetwork(9) for _ in range(3)]
    mu, rho = pack_members(nets, "cpu")
    mu_o, rho_o = orc.pack_members(nets)
    assert torch.equal(mu, mu_o) and torch.equal(rho, rho_o)
    offs = orc.param_offsets(9)
    wo, bo, k_in, k_out = offs[1]
    assert torch.equal(mu[2, wo:wo + k_in * k_out].view(k_out, k_in), nets[2].hidden_layers[1].weight_mu.detach())
    assert torch.equal(mu[2, bo:bo + k_out], nets[2].hidden_layers[1].bias_mu.detach())


# --------------------------------------------------------------------------- pair sharding
@pytest.mark.parametrize("n,world", [(500, 1), (500, 2), (8192, 8), (7, 4), (3, 8)])
def test_shard_pairs_partitions_exactly(n, world):
    from mcmc_amd.engine import shard_pairs
    cover = []
    sizes = []
    for r in range(world):
        b, e = shard_pairs(n, r, world)
        cover.extend(range(b, e))
        sizes.append(e - b)
    assert cover == list(range(n))
    assert max(sizes) - min(sizes) <= 1


def test_shard_pairs_weighted():
    from mcmc_amd.engine import shard_pairs
    w = [1.0, 0.9, 1.1, 1.0, 1.0, 0.95, 1.05, 1.0]
    spans = [shard_pairs(8192, r, 8, w) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 8192 and all(spans[r][1] == spans[r + 1][0] for r in range(7))
    sizes = [e - b for b, e in spans]
    assert all(abs(sizes[r] - 8192 * w[r] / sum(w)) <= 1.0 for r in range(8))
    assert [shard_pairs(5, r, 4, [1, 1, 1, 1]) for r in range(4)] == [(0, 1), (1, 2), (2, 4), (4, 5)]  # round-half-even bounds, exact cover
    assert [shard_pairs(3, r, 4, [100, 1, 1, 1]) for r in range(4)] == [(0, 3), (3, 3), (3, 3), (3, 3)]
    with pytest.raises(ValueError):
        shard_pairs(10, 0, 2, [1.0, 0.0])
    with pytest.raises(ValueError):
        shard_pairs(10, 0, 2, [1.0])


_PLAN_CASES = [
    # (N, M, B, chunk_bytes, chunk_pairs, chunk_rows)
    (8192, 8, 3_650_000, 8 << 30, None, None),     # config 3 / 4: row blocks of all pairs
    (8192, 8, 3_650_000, 8 << 30, 40, None),       # explicit pair chunks, whole rows
    (8192, 8, 3_650_000, 1 << 26, None, None),     # a tight sample buffer: pairs AND rows chunked
    (500, 5, 365, 8 << 30, None, None),            # config 1 / 2
    (500, 5, 365, 8 << 30, 35, None),
    (40, 8, 40_001, 8 << 30, 12, 16384),           # row blocks x pair chunks, ragged last block
    (6, 3, 10, 8 << 30, None, None),               # fewer pairs than ranks at world 8: empty shards
    (200_000, 8, 50_000, 8 << 30, None, None),     # > 131,072 pairs: chunked by the plan itself
]
_PLAN_WEIGHTS = {1: [None], 2: [None, [1.0, 0.6]], 3: [None, [1.0, 1.0, 0.5]], 4: [None, [1.0, 0.9, 1.1, 1.0]],
                 8: [None, [1.0, 0.9, 1.1, 1.0, 1.0, 0.95, 1.05, 1.0], [100.0, 1, 1, 1, 1, 1, 1, 1e-3]]}


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("case", _PLAN_CASES)
def test_predict_plan_invariants(world, case):
    """The shard / chunk / pilot plan mc_predict consumes (engine.plan_predict), world 1..8, equal and weighted shards
    (including shards of zero pairs), chunked and un-chunked pairs: the q-ranges tile [0, N) exactly; every rank walks the
    same row blocks, which tile [0, B); the pilot pairs q < n_pilot lie in their owner's FIRST pair chunk; the ranks'
    pilot shares add up to n_pilot_total == n_pilot; and the plan is the same whichever rank computes it (a pure function)."""
    from mcmc_amd.engine import ROW_GRANULE, pilot_pairs, plan_predict
    N, M, B, chunk_bytes, chunk_pairs, chunk_rows = case
    for weights in _PLAN_WEIGHTS[world]:
        plan = plan_predict(N, M, B, world, weights, chunk_bytes=chunk_bytes, chunk_pairs=chunk_pairs, chunk_rows=chunk_rows)
        assert plan == plan_predict(N, M, B, world, weights, chunk_bytes=chunk_bytes, chunk_pairs=chunk_pairs, chunk_rows=chunk_rows)
        assert len(plan.ranks) == world
        # q-ranges: exact, contiguous, in rank order
        assert plan.ranks[0].q_begin == 0 and plan.ranks[-1].q_end == N
        assert all(plan.ranks[r].q_end == plan.ranks[r + 1].q_begin for r in range(world - 1))
        assert all(rk.q_begin <= rk.q_end for rk in plan.ranks)
        # row blocks: identical for every rank by construction (one tuple), tile [0, B), whole superblocks except the last
        assert plan.row_blocks[0][0] == 0 and sum(rb for _, rb in plan.row_blocks) == B
        assert all(plan.row_blocks[i][0] + plan.row_blocks[i][1] == plan.row_blocks[i + 1][0] for i in range(len(plan.row_blocks) - 1))
        assert all(rb % ROW_GRANULE == 0 for _, rb in plan.row_blocks[:-1]) and all(rb > 0 for _, rb in plan.row_blocks)
        # the sample buffer of the largest shard respects the bound (one superblock of one pair is the floor)
        n_max = max(rk.n_local for rk in plan.ranks)
        ld = (min(plan.rows_per_block, B) + 63) // 64 * 64
        if chunk_pairs is None and chunk_rows is None:
            assert min(plan.chunk_pairs, n_max) * ld * 4 <= max(chunk_bytes, min(ld, ROW_GRANULE) * 4)
        # pilot
        assert 0 < plan.n_pilot <= min(N, max(M, pilot_pairs(N, M)))
        assert plan.n_pilot_total == plan.n_pilot == sum(rk.pilot_share for rk in plan.ranks)
        for rk in plan.ranks:
            chunks = rk.chunks()
            assert sum(c for _, c in chunks) == rk.n_local and all(0 < c <= rk.pairs_per_chunk for _, c in chunks)
            assert rk.hoisted == (len(chunks) <= 1)
            mine = [q for q in range(rk.q_begin, rk.q_end) if q < plan.n_pilot]
            assert len(mine) == rk.pilot_share
            if mine:  # every pilot pair of this rank sits in its first chunk
                b0, c0 = chunks[0]
                assert b0 <= mine[0] and mine[-1] < b0 + c0
        # the pilot set is world-independent unless the pair chunk bounds it (then it is bounded by constants and n_max only)
        one = plan_predict(N, M, B, 1, None, chunk_bytes=chunk_bytes, chunk_pairs=chunk_pairs, chunk_rows=chunk_rows)
        if plan.chunk_pairs >= n_max and one.chunk_pairs >= N:
            assert plan.n_pilot == one.n_pilot


def test_predict_plan_config4_shapes():
    """BASELINE configs[3]: 8,192 pairs x 3.65 M rows over 2 / 4 / 8 ranks - what each rank allocates and launches."""
    from mcmc_amd.engine import plan_predict
    for world, pairs, blocks in ((1, 8192, 14), (2, 4096, 7), (4, 2048, 4), (8, 1024, 2)):
        plan = plan_predict(8192, 8, 3_650_000, world)
        assert all(rk.n_local == pairs and rk.hoisted and rk.pairs_per_chunk == pairs for rk in plan.ranks)
        assert len(plan.row_blocks) == blocks
        assert plan.n_pilot == 128 and [rk.pilot_share for rk in plan.ranks] == [128] + [0] * (world - 1)
        ld = (plan.rows_per_block + 63) // 64 * 64
        assert pairs * ld * 4 <= 8 << 30
    # weighted shards at world 8 move the boundaries, not the pilot or the blocks
    w = [1.0, 0.9, 1.1, 1.0, 1.0, 0.95, 1.05, 1.0]
    plan = plan_predict(8192, 8, 3_650_000, 8, w)
    assert plan.n_pilot == 128 and plan.ranks[0].pilot_share == 128 and sum(rk.n_local for rk in plan.ranks) == 8192
    assert max(rk.n_local for rk in plan.ranks) * ((plan.rows_per_block + 63) // 64 * 64) * 4 <= 8 << 30


def test_pair_order_is_a_member_interleaved_permutation():
    from mcmc_amd.engine import pair_global_ids
    M, S = 5, 100
    g = pair_global_ids(0, M * S, M, S)
    assert sorted(g.tolist()) == list(range(M * S))
    # any chunk of consecutive q of length M holds every member once
    assert sorted((g[35:40] // S).tolist()) == [0, 1, 2, 3, 4]
    assert g[:6].tolist() == [0, 100, 200, 300, 400, 1]


# --------------------------------------------------------------------------- multi-rank combine over gloo (world 2 .. 8)
def _numpy_merge(sum_parts, sumsq_parts, hist_parts, nbins):
    """Test stand-in for Engine.reduce_merge (mcmc_reduce_merge, checked bit for bit against this rule on the GPU in
    tests/test_gpu_parity.py::test_reduce_merge_bit_exact): rank-order sums, uint16 counts."""
    s1, s2 = sum_parts[0].clone(), sumsq_parts[0].clone()
    for p in range(1, sum_parts.shape[0]):
        s1 += sum_parts[p]
        s2 += sumsq_parts[p]
    hist = None
    if hist_parts is not None:
        tot = hist_parts.numpy().view(np.uint16).astype(np.int64).sum(0)
        assert tot.max(initial=0) <= 65535
        hist = torch.from_numpy(tot.astype(np.uint16).view(np.int16))
    return s1, s2, hist


def _combine_worker(rank, world, port, tmp, N, B):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mcmc_amd.engine import ReduceState, combine_partials, gather_rows, row_slice, scatter_combine, shard_pairs
        rng = np.random.default_rng(0)
        nbins, n_pilot = 128, min(16, N)
        Y = (rng.normal(size=(N, B)) * 1.5 + rng.normal(size=B)).astype(np.float32)  # all "samples", same on every rank
        b, e = shard_pairs(N, rank, world)
        # pilot = global pairs q < n_pilot, whichever ranks own them (mc_predict's rule), combined
        pil = Y[b:max(b, min(e, n_pilot))].astype(np.float64)
        pilot = ReduceState(pil.shape[0], torch.from_numpy(pil.sum(0)), torch.from_numpy((pil ** 2).sum(0)))
        pilot = combine_partials(pilot)
        assert pilot.n == n_pilot
        ref_pil = Y[:n_pilot].astype(np.float64)
        assert np.allclose(pilot.sum.numpy(), ref_pil.sum(0), rtol=1e-14)
        lo, inv_w = orc.hist_edges_from_pilot(pilot.n, pilot.sum.numpy(), pilot.sumsq.numpy(), nbins, 6.0)
        mine = Y[b:e]  # empty on ranks beyond the pair count
        hist = orc.hist_accumulate(mine, lo, inv_w, nbins).astype(np.uint16).view(np.int16)
        st = ReduceState(e - b, torch.from_numpy(mine.astype(np.float64).sum(0)),
                         torch.from_numpy((mine.astype(np.float64) ** 2).sum(0)), torch.from_numpy(hist.copy()),
                         torch.from_numpy(lo), torch.from_numpy(inv_w), nbins)
        part0 = ReduceState(st.n, st.sum.clone(), st.sumsq.clone(), st.hist.clone(), st.lo, st.inv_w, nbins)
        st = combine_partials(st)
        assert st.n == N
        full_hist = orc.hist_accumulate(Y, lo, inv_w, nbins)
        got_hist = st.hist.numpy().view(np.uint16).astype(np.int64)
        assert np.array_equal(got_hist, full_hist)
        mean, std, q = orc.finalize_from_state(st.n, st.sum.numpy(), st.sumsq.numpy(), got_hist, lo, inv_w, (0.025, 0.975))
        tm, ts = orc.reduce_mean_std(torch.from_numpy(Y))
        assert np.allclose(mean, tm.numpy(), atol=1e-6) and np.allclose(std, ts.numpy(), rtol=1e-5)
        tq = torch.quantile(torch.from_numpy(Y), torch.tensor([0.025, 0.975]), dim=0).numpy()
        if N >= 32:  # with a handful of samples the pilot's sigma is too rough for the one-bin guarantee
            assert np.all(np.abs(q - tq) <= 1.0 / inv_w[None, :])
        # the product's exchange: reduce-scatter over rows -> finalize own slice -> all-gather
        part, r0, r1 = scatter_combine(part0, None, _numpy_merge)
        assert (r0, r1) == row_slice(B, rank, world)[:2] and part.n == N and part.sum.shape[0] == r1 - r0
        # fp64 sums: same addends, possibly another association than the all-reduce's
        assert torch.allclose(part.sum, st.sum[r0:r1], rtol=1e-14, atol=0) and torch.allclose(part.sumsq, st.sumsq[r0:r1], rtol=1e-14, atol=0)
        assert torch.equal(part.hist, st.hist[r0:r1]) and part.hist.dtype == torch.int16
        if r1 > r0:
            pm, ps, pq = orc.finalize_from_state(part.n, part.sum.numpy(), part.sumsq.numpy(),
                                                 part.hist.numpy().view(np.uint16).astype(np.int64), part.lo.numpy(), part.inv_w.numpy(),
                                                 (0.025, 0.975))
        else:  # B < world: this rank finalizes no rows and still joins the gather
            pm, ps, pq = np.zeros(0), np.zeros(0), np.zeros((2, 0))
        gm, gs, gq, none = gather_rows([torch.from_numpy(np.asarray(pm, dtype=np.float32)), torch.from_numpy(np.asarray(ps, dtype=np.float32)),
                                        torch.from_numpy(np.asarray(pq, dtype=np.float32)), None], B)
        assert none is None and gm.shape == (B,) and gq.shape == (2, B)
        assert np.allclose(gm.numpy(), mean, rtol=1e-6, atol=1e-7) and np.allclose(gs.numpy(), std, rtol=1e-6)
        assert np.array_equal(gq.numpy(), np.asarray(q, dtype=np.float32))   # histogram counts are exact
        # without the library's merge there is no combine: no CPU path
        with pytest.raises(RuntimeError, match="no CPU path"):
            scatter_combine(part0)
        # group=False: single-process semantics under an initialised process group
        same, a0, a1 = scatter_combine(part0, False)
        assert same is part0 and (a0, a1) == (0, B)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


# (world, pairs, rows): even and ragged row slices, B not divisible by 8, fewer pairs than ranks (idle ranks still join every
# collective), fewer rows than ranks (empty row slices)
@pytest.mark.parametrize("world,N,B", [(2, 64, 33), (3, 64, 33), (8, 64, 33), (8, 5, 70), (8, 64, 9), (3, 16, 1), (3, 16, 2)])
def test_combine_partials_gloo(tmp_path, world, N, B):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_combine_worker, args=(world, port, str(tmp_path), N, B), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_combine_is_identity_without_process_group():
    from mcmc_amd.engine import ReduceState, combine_partials
    st = ReduceState(3, torch.ones(4, dtype=torch.float64), torch.ones(4, dtype=torch.float64))
    assert combine_partials(st) is st


# --------------------------------------------------------------------------- generator (row f3)
def test_generator_philox_matches_oracle():
    """The generator's Philox (integer torch ops) against the oracle's, which the published known-answer vectors pin."""
    from mcmc_amd.data import philox_u32
    seed, stream, n = 0xFEDCBA9876543210, 5, 1003
    got = philox_u32(seed, stream, n, "cpu").numpy().astype(np.uint32)
    ctr = np.zeros(((n + 3) // 4, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(ctr.shape[0])
    ctr[:, 1] = stream
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    ref = orc.philox4x32_10(ctr, np.broadcast_to(key, (ctr.shape[0], 2))).reshape(-1)[:n]
    assert np.array_equal(got, ref)


def test_series_batch_generator():
    from mcmc_amd.data import calendar_features, make_series_batch
    X, y = make_series_batch(50, seed=4, standardize=False)
    assert X.shape == (50 * 365, 11) and y.shape == (50 * 365,)
    Xs = X.view(50, 365, 11)
    assert torch.all(Xs[:, :, 0].sum(1) == 24) and torch.all(Xs[:, :, 1].sum(1) == 30)  # data.py:57,74
    assert torch.all(Xs[:, :, 2:5] == 0)                                                # data.py:127-129
    assert torch.allclose(Xs[0, :, 5:], calendar_features())
    assert float(y.min()) >= 0 and 900 < float(y.mean()) < 1400
    X2, y2 = make_series_batch(50, seed=4, standardize=False)
    assert torch.equal(X, X2) and torch.equal(y, y2)
    X3, _ = make_series_batch(50, seed=5, standardize=False)
    assert not torch.equal(X, X3)
    Xn, yn = make_series_batch(50, seed=4)
    assert torch.allclose(Xn.mean(0), torch.zeros(11), atol=1e-4)
    assert torch.allclose(Xn.std(0, unbiased=False)[[0, 1, 5, 6, 7, 8, 9, 10]], torch.ones(8), atol=1e-3)
    assert torch.all(Xn[:, 2:5] == 0)  # StandardScaler: zero-variance columns -> 0 (main.py:116-118)


def test_calendar_features_match_reference_generator(golden):
    """Same six sin/cos columns as data.py:155-167 (config-1 fixture's raw X, columns 5..10)."""
    from mcmc_amd.data import calendar_features
    g = golden("predict_cfg1.npz")
    assert np.allclose(calendar_features().numpy(), g["X_raw"][:, 5:], atol=1e-6)


# ----------------------------------------------------------------------------- main.py glue (row f2)
GLUE_TAGS = {"baseline": (), "0": (0,), "1_3": (1, 3), "0_1_2_3_4": (0, 1, 2, 3, 4)}


def test_pipeline_feature_combinations_match_reference_order(golden):
    from mcmc_amd import pipeline
    z = golden("glue_cfg1.npz")
    combos = pipeline.prepare_feature_combinations(5)
    assert len(combos) == int(z["n_combos"]) == 32
    assert [pipeline.combo_name(c) for c in combos] == [str(s) for s in z["combo_order"]]


@pytest.mark.parametrize("tag", sorted(GLUE_TAGS))
def test_pipeline_scaling_matches_sklearn(golden, tag):
    from mcmc_amd import pipeline
    z = golden("glue_cfg1.npz")
    X = pipeline.select_features(torch.from_numpy(z["features"]), torch.from_numpy(z["temporal"]), GLUE_TAGS[tag])
    assert X.names == ("time", "features") and X.shape[1] == len(GLUE_TAGS[tag] or range(5)) + 6
    Xs, ys, sx, sy = pipeline.scale_inputs(X, torch.from_numpy(z["target"]))
    assert Xs.names == ("time", "features") and ys.names == ("time",) and Xs.dtype == torch.float32
    # float32 results of float64 statistics: at most one float32 ulp from sklearn's
    np.testing.assert_allclose(Xs.rename(None).numpy(), z[f"Xs_{tag}"], rtol=3e-7, atol=3e-7)
    np.testing.assert_allclose(ys.rename(None).numpy(), z[f"ys_{tag}"], rtol=3e-7, atol=3e-7)
    mean, std = pipeline.descale_prediction(torch.from_numpy(z[f"fake_mean_{tag}"]), torch.from_numpy(z[f"fake_std_{tag}"]), sy)
    assert mean.dtype == torch.float64 and std.dtype == torch.float32
    np.testing.assert_allclose(mean.numpy(), z[f"mean_{tag}"], rtol=1e-12)
    np.testing.assert_allclose(std.numpy(), z[f"std_{tag}"], rtol=3e-7)
    met = pipeline.evaluate_predictions(z["target"], z[f"mean_{tag}"])
    assert abs(met["rmse"] - float(z[f"rmse_{tag}"])) <= 1e-6 * float(z[f"rmse_{tag}"])
    assert abs(met["mape"] - float(z[f"mape_{tag}"])) <= 1e-6 * float(z[f"mape_{tag}"])


def test_pipeline_record_schema_and_constant_columns(golden, tmp_path):
    import json
    from mcmc_amd import pipeline
    z = golden("glue_cfg1.npz")
    X = pipeline.select_features(torch.from_numpy(z["features"]), torch.from_numpy(z["temporal"]), ())
    Xs, ys, sx, sy = pipeline.scale_inputs(X, torch.from_numpy(z["target"]))
    # the three placeholder indicator columns are constant zero (data.py:127-129): they stay 0, not NaN
    assert torch.equal(Xs.rename(None)[:, 2:5], torch.zeros(365, 3))
    y = torch.from_numpy(z["target"])
    rec = pipeline.bayesian_record(ys.rename(None), torch.full((365,), 0.1), y, sy, ys.rename(None) - 0.2, ys.rename(None) + 0.2)
    assert set(rec) == {"predictions", "uncertainty", "metrics", "lower", "upper"} and set(rec["metrics"]) == {"rmse", "mape"}
    assert len(rec["predictions"]) == len(rec["uncertainty"]) == 365
    np.testing.assert_allclose(rec["predictions"], z["target"], rtol=1e-5)   # inverse of the forward scaling
    assert all(lo < m < hi for lo, m, hi in zip(rec["lower"], rec["predictions"], rec["upper"]))
    rec2 = pipeline.bayesian_record(ys.rename(None), torch.full((365,), 0.1), y, sy)
    assert set(rec2) == {"predictions", "uncertainty", "metrics"}            # main.py:192-196 keys only
    res = {"feature_dates": {}, "actual": y.tolist(), "predictions": {"baseline": {"bayesian": rec2}}}
    path = tmp_path / "results" / "model_results.json"
    pipeline.save_results(res, str(path))
    back = json.loads(path.read_text())
    assert back["predictions"]["baseline"]["bayesian"]["metrics"]["rmse"] == rec2["metrics"]["rmse"]
    if not torch.cuda.is_available():  # main.py:225-226
        with pytest.raises(RuntimeError):
            pipeline.train_and_evaluate_all(torch.from_numpy(z["features"]), torch.from_numpy(z["temporal"]), y)


# ----------------------------------------------------------------------------- TFT host side (row f4)
def test_tft_module_parameters_follow_the_reference_layout():
    from oracle import tft_oracle as tft
    from mcmc_amd import _lib
    from mcmc_amd.temporal_fusion import TemporalFusionTransformer, TFTModel
    net = TemporalFusionTransformer(11)
    assert [(n, tuple(p.shape)) for n, p in net.named_parameters()] == tft.param_spec(11)
    flat = net.flat_parameters()
    assert flat.numel() == tft.n_params(11) == _lib.load().mcmc_tft_param_count(11)
    net2 = TemporalFusionTransformer(11)
    net2.load_flat(flat)
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), net2.parameters()))
    m = TFTModel(num_features=11, device="cpu")
    with pytest.raises(RuntimeError):   # no CPU path
        m.predict(torch.zeros(40, 11))
    with pytest.raises(ValueError):
        m.predict(torch.zeros(40, 5))


def test_one_cycle_schedule_equals_torch_stepwise():
    """train.one_cycle_schedule (scheduler stepped alone, cached) == the (lr, beta1) sequence a training loop sees when
    it calls optimizer.step(); scheduler.step() after every batch (bayesian_ensemble.py:137-150,180-181), and the
    oracle's closed form."""
    from mcmc_amd.train import one_cycle_schedule
    epochs, spe = 7, 12
    lrs, b1s = one_cycle_schedule(epochs, spe)
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([dummy], lr=0.001, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.01, epochs=epochs, steps_per_epoch=spe, pct_start=0.3, anneal_strategy="cos")
    for i in range(epochs * spe):
        g = opt.param_groups[0]
        assert float(g["lr"]) == lrs[i] and float(g["betas"][0]) == b1s[i], i
        if i + 1 < epochs * spe:
            opt.step()
            sched.step()
    assert one_cycle_schedule(epochs, spe) == (lrs, b1s)  # cached copy, fresh lists
    assert lrs[0] == pytest.approx(0.01 / 25) and max(lrs) == pytest.approx(0.01, rel=1e-3) and b1s[0] == pytest.approx(0.95)


def test_dropin_utils_exports_what_main_imports(capsys):
    """main.py:12 does `from utils import evaluate_predictions`; bayesian_ensemble.py:6 `from utils import peak_weighted_loss`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dropin_utils", os.path.join(ROOT, "dropin", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.peak_weighted_loss)
    out = mod.evaluate_predictions(np.array([100.0, 200.0]), np.array([110.0, 190.0]), "Bayesian")
    assert abs(out["rmse"] - 10.0) < 1e-9 and abs(out["mape"] - 7.5) < 1e-9
    printed = capsys.readouterr().out
    assert "Bayesian Performance:" in printed and "RMSE: 10.00" in printed and "MAPE: 7.50%" in printed
