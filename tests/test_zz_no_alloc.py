"""The allocation-free entry point (mcmc_loss_fused_static: the loss's partial sums in a file-scope device array instead of a
workspace argument) against mcmc_loss_fused, through the C ABI: two calls in a row, the second on a smaller input, bit for bit.
Collected last on purpose (the round-end run uses pytest -x)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_static_scratch_loss_equals_workspace_loss():
    from mcmc_amd import _lib
    from mcmc_amd.engine import Engine, _p, _stream
    dev = torch.device("cuda", 0)
    eng = Engine(dev)
    lib, h = eng.lib, eng.ctx.handle
    ws = torch.empty(int(lib.mcmc_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
    g = torch.Generator().manual_seed(5)
    for n in (1_000_003, 4_099, 13):  # 1024 workgroups, then fewer slots of the same scratch, then one ragged workgroup
        t = torch.randn(n, generator=g).to(dev)
        p = (t.cpu() + 0.3 * torch.randn(n, generator=g)).to(dev)
        want_l, got_l = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        want_g, got_g = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        _lib.check(lib.mcmc_loss_fused(h, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(want_l), _p(want_g), _p(ws), _stream()), "mcmc_loss_fused")
        _lib.check(lib.mcmc_loss_fused_static(h, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(got_l), _p(got_g), _stream()), "mcmc_loss_fused_static")
        assert torch.equal(got_l, want_l) and torch.equal(got_g, want_g), n
        assert float(got_l) > 0 and torch.isfinite(got_g).all()
        # forward value alone (grad NULL)
        only_l = torch.zeros(1, device=dev)
        _lib.check(lib.mcmc_loss_fused_static(h, _p(p), _p(t), n, 2.0, 0.5, 1.0, _p(only_l), None, _stream()), "mcmc_loss_fused_static")
        assert torch.equal(only_l, want_l)
    # against the fp64 closed form of the oracle too (1e-5 bar of the loss tests)
    from oracle import bnn_oracle as orc
    L, G = orc.peak_weighted_loss_closed_form(p.cpu().numpy(), t.cpu().numpy())
    assert abs(float(got_l) - L) <= 2e-6 * max(1.0, abs(L)) and float(np.abs(got_g.cpu().numpy() - G).max()) <= 1e-5
    # sizes beyond the stated maximum are refused with a clear error, nothing launched
    with pytest.raises(ValueError, match="MCMC_LOSS_STATIC_MAX_N"):
        _lib.check(lib.mcmc_loss_fused_static(h, _p(p), _p(t), 1 << 31, 2.0, 0.5, 1.0, _p(got_l), None, _stream()), "mcmc_loss_fused_static")
