"""The glue main.py puts around the Bayesian path, restated for the HIP classes.

Mirrors /root/reference/main.py: feature-subset enumeration (:38-57), column selection + concat
with the calendar features (:71-96), the two StandardScalers (:111-130), train -> predict
(:148-150,169-170), inverse scaling of the mean and scaling of the std (:174-177), metrics
(utils.py:30-40) and the JSON record (:183-196, :241-245, :282-283).  The Prophet row of the
record is not produced here (CPU library, out of scope); a record written by this module carries
the ``bayesian`` entry with the reference's keys (plus opt-in ``lower`` / ``upper`` interval
bounds) and, with ``include_tft=True``, the ``tft`` entry (main.py:141-146,158-165,188-191).

Scaling runs as torch ops on whatever device the tensors live on; the arithmetic follows
sklearn's StandardScaler on float32 input: statistics in float64 (population variance, scale of
a constant column := 1), each in-place step rounded back to float32.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from itertools import combinations
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def prepare_feature_combinations(n_features: int) -> List[Tuple[int, ...]]:
    """[()] (baseline = all indicator columns) + every non-empty subset, in main.py:49-54 order."""
    combos: List[Tuple[int, ...]] = [tuple()]
    for r in range(1, n_features + 1):
        combos.extend(combinations(range(n_features), r))
    return combos


def combo_name(combo: Sequence[int]) -> str:
    """main.py:63."""
    return "_".join(map(str, combo)) if combo else "baseline"


def select_features(features: torch.Tensor, temporal: torch.Tensor, combo: Sequence[int]) -> torch.Tensor:
    """X = [selected indicator columns | calendar columns], named ('time','features') (main.py:71-96)."""
    f = features.rename(None)
    if combo:
        f = f.index_select(1, torch.tensor(list(combo), device=f.device))
    return torch.cat([f, temporal.rename(None)], dim=1).refine_names("time", "features")


@dataclass
class StandardScaler:
    """sklearn.preprocessing.StandardScaler (with_mean, with_std) for float32 tensors on any device."""
    mean_: Optional[torch.Tensor] = None   # float64 [C]
    scale_: Optional[torch.Tensor] = None  # float64 [C]
    var_: Optional[torch.Tensor] = None

    def fit(self, X: torch.Tensor) -> "StandardScaler":
        x = X.rename(None)
        if x.dim() == 1:
            x = x.reshape(-1, 1)
        x64 = x.double()
        self.mean_ = x64.mean(dim=0)
        self.var_ = ((x64 - self.mean_) ** 2).mean(dim=0)  # population variance, like sklearn
        scale = self.var_.sqrt()
        # sklearn _handle_zeros_in_scale: constant columns (scale below 10 eps) are left unscaled
        eps = torch.finfo(torch.float64).eps
        self.scale_ = torch.where(scale < 10 * eps, torch.ones_like(scale), scale)
        return self

    def transform(self, X: torch.Tensor) -> torch.Tensor:
        x = X.rename(None)
        one_d = x.dim() == 1
        if one_d:
            x = x.reshape(-1, 1)
        out = (x.double() - self.mean_).to(x.dtype)           # X -= mean_   (rounded to the array dtype)
        out = (out.double() / self.scale_).to(x.dtype)        # X /= scale_
        return out.squeeze(-1) if one_d else out

    def fit_transform(self, X: torch.Tensor) -> torch.Tensor:
        return self.fit(X).transform(X)

    def inverse_transform(self, X: torch.Tensor) -> torch.Tensor:
        x = X.rename(None)
        one_d = x.dim() == 1
        if one_d:
            x = x.reshape(-1, 1)
        scale, mean = self.scale_.to(x.device), self.mean_.to(x.device)
        out = (x.double() * scale).to(x.dtype)                # X *= scale_
        out = (out.double() + mean).to(x.dtype)               # X += mean_
        return out.squeeze(-1) if one_d else out


def scale_inputs(X: torch.Tensor, y: torch.Tensor):
    """main.py:111-130: (X_scaled named ('time','features'), y_scaled named ('time',), scaler_X, scaler_y)."""
    sx, sy = StandardScaler(), StandardScaler()
    Xs = sx.fit_transform(X.rename(None)).refine_names("time", "features")
    ys = sy.fit_transform(y.rename(None)).refine_names("time")
    return Xs, ys, sx, sy


def descale_prediction(mean: torch.Tensor, std: torch.Tensor, scaler_y: StandardScaler):
    """main.py:174-177: mean -> inverse_transform, std -> std * scale_[0], on the CPU.
    sklearn hands a float32 *torch* tensor back as float64 (measured with the reference's call),
    so the de-scaled mean is float64; the std stays float32 (tensor * numpy scalar)."""
    m = scaler_y.inverse_transform(mean.cpu().rename(None).double().reshape(-1, 1)).flatten()
    s = (std.cpu().rename(None).double() * scaler_y.scale_[0].cpu()).to(torch.float32)
    return m, s


def evaluate_predictions(y_true, y_pred) -> Dict[str, float]:
    """utils.py:30-40: rmse and sklearn's mean_absolute_percentage_error * 100."""
    t = torch.as_tensor(y_true).double().reshape(-1)
    p = torch.as_tensor(y_pred).double().reshape(-1)
    rmse = float(((t - p) ** 2).mean().sqrt())
    eps = torch.finfo(torch.float64).eps
    mape = float(((p - t).abs() / t.abs().clamp(min=eps)).mean() * 100)
    return {"rmse": rmse, "mape": mape}


def bayesian_record(mean: torch.Tensor, std: torch.Tensor, y: torch.Tensor, scaler_y: StandardScaler,
                    lower: Optional[torch.Tensor] = None, upper: Optional[torch.Tensor] = None) -> dict:
    """The ``bayesian`` entry of results['predictions'][combo] (main.py:192-196); ``lower``/``upper``
    (scaled-space interval bounds) are de-scaled like the mean and added only when given."""
    m, s = descale_prediction(mean, std, scaler_y)
    rec = {"predictions": m.tolist(), "uncertainty": s.tolist(),
           "metrics": evaluate_predictions(y.rename(None).cpu(), m)}
    if lower is not None and upper is not None:
        rec["lower"] = scaler_y.inverse_transform(lower.cpu().rename(None).double().reshape(-1, 1)).flatten().tolist()
        rec["upper"] = scaler_y.inverse_transform(upper.cpu().rename(None).double().reshape(-1, 1)).flatten().tolist()
    return rec


def process_bayesian_task(features: torch.Tensor, temporal: torch.Tensor, target: torch.Tensor, combo: Sequence[int], device,
                          *, epochs: int = 400, train_samples: int = 10, predict_samples: int = 100, interval: Optional[float] = None,
                          n_models: int = 5, compute_dtype: str = "fp32", seed: Optional[int] = None, log=print,
                          include_tft: bool = False, tft_epochs: Optional[int] = None):
    """The GPU legs of main.py:process_gpu_task (:60-208) for one feature subset: the Bayesian
    ensemble and, with ``include_tft``, the TFT.  Returns (combo_name, {"bayesian": record[, "tft": record]})."""
    from .bayesian_ensemble import GPUBayesianEnsemble
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.set_device(device)  # main.py:66-67
    name = combo_name(combo)
    log(f"Training models for combination: {name} on {device}")
    X = select_features(features, temporal, combo).rename(None).to(device)
    y = target.rename(None).to(device)
    Xs, ys, _, sy = scale_inputs(X, y)
    out = {}
    if include_tft:  # main.py:141-146,158-165: TFT first, like the reference
        from .temporal_fusion import TFTModel
        log(f"Training TFT for {name}")
        tft_model = TFTModel(num_features=Xs.size("features"), seq_length=30, batch_size=32, device=device, seed=seed)
        tft_model.train(Xs, ys, epochs=epochs if tft_epochs is None else tft_epochs)
        tft_preds = sy.inverse_transform(torch.from_numpy(tft_model.predict(Xs)).reshape(-1, 1)).flatten()
        out["tft"] = {"predictions": tft_preds.tolist(), "metrics": evaluate_predictions(y.cpu(), tft_preds)}
        log(f"TFT MAPE: {out['tft']['metrics']['mape']:.2f}%")
    log(f"Training Bayesian for {name}")
    model = GPUBayesianEnsemble(input_dim=Xs.size("features"), n_models=n_models, device=device, batch_size=32,
                                compute_dtype=compute_dtype, seed=seed)
    model.train(Xs, ys, epochs=epochs, num_samples=train_samples)
    if interval is None:
        mean, std = model.predict(Xs, num_samples=predict_samples)
        rec = bayesian_record(mean, std, y, sy)
    else:
        mean, std, lo, hi = model.predict_interval(Xs, num_samples=predict_samples, level=interval)
        rec = bayesian_record(mean, std, y, sy, lo, hi)
    log(f"Bayesian MAPE: {rec['metrics']['mape']:.2f}%")
    out["bayesian"] = rec
    return name, out


def train_and_evaluate_all(features: torch.Tensor, temporal: torch.Tensor, target: torch.Tensor, feature_dates: Optional[dict] = None,
                           *, combos: Optional[Sequence[Sequence[int]]] = None, devices: Optional[Sequence] = None, **task_kw) -> dict:
    """main.py:220-253: every feature subset, task i on device i % num_gpus, serial like the reference."""
    if devices is None:
        if not torch.cuda.is_available():
            raise RuntimeError("CUDA is not available")  # main.py:225-226 (a HIP device here)
        devices = [torch.device("cuda", i) for i in range(torch.cuda.device_count())]
    if combos is None:
        combos = prepare_feature_combinations(features.rename(None).size(1))
    results = {"feature_dates": feature_dates or {}, "actual": target.cpu().rename(None).tolist(), "predictions": {}}
    for i, combo in enumerate(combos):
        name, rec = process_bayesian_task(features, temporal, target, combo, devices[i % len(devices)], **task_kw)
        results["predictions"][name] = rec
    return results


def save_results(results: dict, path: str = "results/model_results.json"):
    """main.py:270,282-283."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump(results, f)
