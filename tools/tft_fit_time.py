"""Dev tool: wall time per training step inside TFTModel.train (host loop + kernels), 365-day series."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd.temporal_fusion import TFTModel  # noqa: E402

g = torch.Generator().manual_seed(0)
X = torch.randn(365, 11, generator=g)
y = torch.randn(365, generator=g)
m = TFTModel(num_features=11, seq_length=30, batch_size=32, device="cuda:0", seed=1)
import builtins
_print = builtins.print
builtins.print = lambda *a, **k: None
m.train(X, y, epochs=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.train(X, y, epochs=40)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
builtins.print = _print
steps = m.epochs_trained * 11
print(f"{m.epochs_trained} epochs x 11 steps: {dt:.3f} s = {dt / steps * 1e6:.1f} us per step")
