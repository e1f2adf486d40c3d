"""Seeded, batched synthetic series for the multi-store / multi-product workloads.

The reference makes ONE unseeded 365-day series (/root/reference/data.py:37-194).  Configs 3-5
need 10k-100k reproducible series, so this generator draws every series with the reference's
recipe - trend + yearly + weekly demand (data.py:37-52), 24 promotions with a 3-5 day decaying
lift (:54-70), 30 weather events of 1-3 days (:72-86), 5 % noise (:139), three all-zero
placeholder indicators (:127-129), six calendar sin/cos columns (:155-167) - vectorised over
series on the target device, then standardises columns like main.py:111-130 (StandardScaler:
zero-variance columns map to 0).

Every random draw comes from Philox4x32-10 evaluated in integer torch ops (counter = draw index, stream id;
key = seed), so the CPU and a HIP device produce the SAME draws for a seed: the promotion / weather days and
durations - hence the indicator columns - are identical on both, and the standardised X and y agree to float
rounding (tests/test_gpu_parity.py::test_device_generator_equals_cpu).
"""
from __future__ import annotations

import datetime as _dt
import math
from typing import Tuple

import torch


def calendar_features(periods: int = 365, start: str = "2023-01-01") -> torch.Tensor:
    """[periods, 6]: sin/cos of day-of-year/365, ISO week/52, month/12 (data.py:155-167)."""
    d0 = _dt.datetime.strptime(start, "%Y-%m-%d").date()
    doy, week, month = [], [], []
    for i in range(periods):
        d = d0 + _dt.timedelta(days=i)
        doy.append(d.timetuple().tm_yday)
        week.append(d.isocalendar()[1])
        month.append(d.month)
    doy, week, month = (torch.tensor(v, dtype=torch.float32) for v in (doy, week, month))
    two_pi = 2 * math.pi
    return torch.stack([torch.sin(two_pi * doy / 365), torch.cos(two_pi * doy / 365),
                        torch.sin(two_pi * week / 52), torch.cos(two_pi * week / 52),
                        torch.sin(two_pi * month / 12), torch.cos(two_pi * month / 12)], dim=1)


_M0, _M1, _W0, _W1, _MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF


def _mulhilo(a: int, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(high, low) 32 bits of a * b for a 32-bit constant and 32-bit values held in int64 (no 64-bit overflow:
    the product is formed from the 16-bit halves of b)."""
    p, q = a * (b & 0xFFFF), a * (b >> 16)
    return (q + (p >> 16)) >> 16, (p + ((q & 0xFFFF) << 16)) & _MASK


def philox_u32(seed: int, stream: int, n: int, device) -> torch.Tensor:
    """n 32-bit draws (int64 tensor) of Philox4x32-10 (Salmon et al., SC'11): counter (i, stream, 0, 0), key = seed's
    two halves, the four words of block i are draws 4i .. 4i+3.  Integer torch ops only: identical on CPU and GPU."""
    blocks = (n + 3) // 4
    c0 = torch.arange(blocks, dtype=torch.int64, device=device)
    c1 = torch.full_like(c0, stream & _MASK)
    c2, c3 = torch.zeros_like(c0), torch.zeros_like(c0)
    k0, k1 = seed & _MASK, (seed >> 32) & _MASK
    for _ in range(10):
        h0, l0 = _mulhilo(_M0, c0)
        h1, l1 = _mulhilo(_M1, c2)
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return torch.stack([c0, c1, c2, c3], dim=1).reshape(-1)[:n]


class _Draws:
    """Numbered Philox streams of one seed: every call takes the next stream id."""

    def __init__(self, seed: int, device):
        self.seed, self.device, self.stream = seed, device, 0

    def u32(self, shape) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= d
        self.stream += 1
        return philox_u32(self.seed, self.stream, n, self.device).reshape(shape)

    def uniform(self, shape) -> torch.Tensor:
        """(0, 1) in float64: (x + 0.5) / 2^32, exact on every device."""
        return (self.u32(shape).to(torch.float64) + 0.5) * (1.0 / 4294967296.0)

    def randint(self, low: int, high: int, shape) -> torch.Tensor:
        return low + self.u32(shape) % (high - low)

    def normal(self, shape) -> torch.Tensor:
        """Box-Muller in float64, returned as float32."""
        u1, u2 = self.uniform(shape), self.uniform(shape)
        return (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)).to(torch.float32)


def _random_dates(n_series: int, periods: int, k: int, draws: _Draws) -> torch.Tensor:
    """k distinct days per series (the reference's randperm(periods)[:k], data.py:57,74)."""
    return draws.uniform((n_series, periods)).argsort(dim=1, stable=True)[:, :k]


def make_series_batch(n_series: int, periods: int = 365, seed: int = 0, device="cpu",
                      standardize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """X [n_series*periods, 11] fp32, y [n_series*periods] fp32 (series-major rows)."""
    device = torch.device(device)
    draws = _Draws(seed, device)
    t = torch.arange(periods, dtype=torch.float32, device=device)
    base = 1000 + 0.5 * t + 200 * torch.sin(2 * math.pi * t / 365) + 50 * torch.sin(2 * math.pi * t / 7)
    base = base.expand(n_series, periods)

    promo_ind = torch.zeros(n_series, periods, device=device)
    promo_eff = torch.zeros(n_series, periods, device=device)
    dates = _random_dates(n_series, periods, 24, draws)
    dur = draws.randint(3, 6, tuple(dates.shape))
    promo_ind.scatter_(1, dates, 1.0)
    pattern = torch.tensor([0.5, 1.0, 0.7, 0.3, 0.1], device=device)
    for j in range(5):
        day = dates + j
        amount = torch.where((j < dur) & (day < periods), pattern[j], torch.zeros((), device=device))
        promo_eff.scatter_add_(1, day.clamp(max=periods - 1), amount.expand_as(day).contiguous())

    weather_ind = torch.zeros(n_series, periods, device=device)
    weather_eff = torch.zeros(n_series, periods, device=device)
    wdates = _random_dates(n_series, periods, 30, draws).sort(dim=1).values
    wdur = draws.randint(1, 4, tuple(wdates.shape))
    weff = draws.normal(tuple(wdates.shape)) * 0.2
    weather_ind.scatter_(1, wdates, 1.0)
    for e in range(wdates.shape[1]):  # later events overwrite earlier ones (data.py:83), in date order
        for j in range(3):
            day = wdates[:, e] + j
            ok = (j < wdur[:, e]) & (day < periods)
            idx = day.clamp(max=periods - 1).unsqueeze(1)
            cur = weather_eff.gather(1, idx).squeeze(1)
            weather_eff.scatter_(1, idx, torch.where(ok, weff[:, e], cur).unsqueeze(1))

    demand = base + 300 * promo_eff + base * weather_eff
    noise = draws.normal((n_series, periods)) * (base.std(dim=1, keepdim=True) * 0.05)
    y = torch.clamp(demand + noise, min=0).reshape(-1)

    zeros = torch.zeros(n_series, periods, device=device)
    cal = calendar_features(periods).to(device).unsqueeze(0).expand(n_series, periods, 6)
    X = torch.cat([torch.stack([promo_ind, weather_ind, zeros, zeros, zeros], dim=2), cal], dim=2).reshape(-1, 11)
    if standardize:
        Xd, yd = X.double(), y.double()  # float64 statistics: the same on every device to float32 rounding
        mean = Xd.mean(dim=0, keepdim=True)
        std = Xd.std(dim=0, unbiased=False, keepdim=True)
        X = ((Xd - mean) / torch.where(std > 0, std, torch.ones_like(std))).float()
        y = ((yd - yd.mean()) / yd.std(unbiased=False)).float()
    return X.contiguous().float(), y.contiguous().float()
