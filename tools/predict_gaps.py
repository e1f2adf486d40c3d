"""Dev tool: is a config-3 predict bound by anything but its kernels?  Prints the wall time of one mc_predict, when the host
returned (everything is enqueued asynchronously), the per-phase event sums, and the GPU idle time between kernels from the
torch profiler's device timeline.  Round 2: host returns after 2-3 ms, idle 0.0 ms of 2,286 ms."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_members
from mcmc_amd import mc_predict, pack_members
from mcmc_amd.data import make_series_batch
dev = torch.device("cuda", 0)
X, _ = make_series_batch(10000, seed=2025, device=dev)
mu, rho = pack_members(make_members(8, 11, 1, dev), dev)
kw = dict(dtype="bf16_fast", noise="philox", q_levels=(0.025, 0.975), nbins=128)
mc_predict(X, mu, rho, 1024, seed=0, **kw); torch.cuda.synchronize()
for it in range(2):
    timers = {}
    t0 = time.perf_counter()
    mc_predict(X, mu, rho, 1024, seed=1 + it, timers=timers, **kw)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ph = {k: sum(a.elapsed_time(b) for a, b, _ in v) for k, v in timers.items()}
    print(f"wall {wall*1e3:.1f} ms, host returned after {t_host*1e3:.1f} ms, phases sum {sum(ph.values()):.1f} ms", {k: round(v, 2) for k, v in ph.items()})
# trace gaps: kernel-level timeline via torch profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    mc_predict(X, mu, rho, 1024, seed=9, **kw); torch.cuda.synchronize()
evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
tot = sum(e.time_range.end - e.time_range.start for e in evs)
span = evs[-1].time_range.end - evs[0].time_range.start
print(f"GPU kernels busy {tot/1e3:.1f} ms over a span of {span/1e3:.1f} ms; idle {(span-tot)/1e3:.1f} ms")
gaps = sorted(((evs[i+1].time_range.start - evs[i].time_range.end, evs[i].name[:40], evs[i+1].name[:40]) for i in range(len(evs)-1)), reverse=True)[:8]
for g in gaps: print(f"  gap {g[0]/1e3:.2f} ms after {g[1]} before {g[2]}")
