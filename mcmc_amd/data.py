"""Seeded, batched synthetic series for the multi-store / multi-product workloads.

The reference makes ONE unseeded 365-day series (/root/reference/data.py:37-194).  Configs 3-5
need 10k-100k reproducible series, so this generator draws every series with the reference's
recipe - trend + yearly + weekly demand (data.py:37-52), 24 promotions with a 3-5 day decaying
lift (:54-70), 30 weather events of 1-3 days (:72-86), 5 % noise (:139), three all-zero
placeholder indicators (:127-129), six calendar sin/cos columns (:155-167) - vectorised over
series on the target device, then standardises columns like main.py:111-130 (StandardScaler:
zero-variance columns map to 0).
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


def _random_dates(n_series: int, periods: int, k: int, gen: torch.Generator, device) -> torch.Tensor:
    """k distinct days per series (the reference's randperm(periods)[:k], data.py:57,74)."""
    return torch.rand(n_series, periods, generator=gen, device=device).argsort(dim=1)[:, :k]


def make_series_batch(n_series: int, periods: int = 365, seed: int = 0, device="cpu",
                      standardize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """X [n_series*periods, 11] fp32, y [n_series*periods] fp32 (series-major rows)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    t = torch.arange(periods, dtype=torch.float32, device=device)
    base = 1000 + 0.5 * t + 200 * torch.sin(2 * math.pi * t / 365) + 50 * torch.sin(2 * math.pi * t / 7)
    base = base.expand(n_series, periods)

    promo_ind = torch.zeros(n_series, periods, device=device)
    promo_eff = torch.zeros(n_series, periods, device=device)
    dates = _random_dates(n_series, periods, 24, gen, device)
    dur = torch.randint(3, 6, dates.shape, generator=gen, device=device)
    promo_ind.scatter_(1, dates, 1.0)
    pattern = torch.tensor([0.5, 1.0, 0.7, 0.3, 0.1], device=device)
    for j in range(5):
        day = dates + j
        amount = torch.where((j < dur) & (day < periods), pattern[j], torch.zeros((), device=device))
        promo_eff.scatter_add_(1, day.clamp(max=periods - 1), amount.expand_as(day).contiguous())

    weather_ind = torch.zeros(n_series, periods, device=device)
    weather_eff = torch.zeros(n_series, periods, device=device)
    wdates = _random_dates(n_series, periods, 30, gen, device).sort(dim=1).values
    wdur = torch.randint(1, 4, wdates.shape, generator=gen, device=device)
    weff = torch.randn(wdates.shape, generator=gen, device=device) * 0.2
    weather_ind.scatter_(1, wdates, 1.0)
    for e in range(wdates.shape[1]):  # later events overwrite earlier ones (data.py:83), in date order
        for j in range(3):
            day = wdates[:, e] + j
            ok = (j < wdur[:, e]) & (day < periods)
            idx = day.clamp(max=periods - 1).unsqueeze(1)
            cur = weather_eff.gather(1, idx).squeeze(1)
            weather_eff.scatter_(1, idx, torch.where(ok, weff[:, e], cur).unsqueeze(1))

    demand = base + 300 * promo_eff + base * weather_eff
    noise = torch.randn(n_series, periods, generator=gen, device=device) * (base.std(dim=1, keepdim=True) * 0.05)
    y = torch.clamp(demand + noise, min=0).reshape(-1)

    zeros = torch.zeros(n_series, periods, device=device)
    cal = calendar_features(periods).to(device).unsqueeze(0).expand(n_series, periods, 6)
    X = torch.cat([torch.stack([promo_ind, weather_ind, zeros, zeros, zeros], dim=2), cal], dim=2).reshape(-1, 11)
    if standardize:
        mean = X.mean(dim=0, keepdim=True)
        std = X.std(dim=0, unbiased=False, keepdim=True)
        X = (X - mean) / torch.where(std > 0, std, torch.ones_like(std))
        y = (y - y.mean()) / y.std(unbiased=False)
    return X.contiguous().float(), y.contiguous().float()
