"""ctypes binding of libmcmc_hip.so (include/mcmc_hip.h).

The library is the product: there is no CPU fallback.  A missing or unloadable library, or a
missing symbol, raises ``MCMCLibraryError`` at first use.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCMC_LIB_PATH") or os.path.join(_HERE, "lib", "libmcmc_hip.so")  # env: dev builds of the library

DT_F32, DT_BF16, DT_BF16_FAST = 0, 1, 2
NOISE_NONE, NOISE_INJECTED, NOISE_PHILOX = 0, 1, 2
ABI_VERSION = 21


class MCMCLibraryError(RuntimeError):
    """libmcmc_hip.so missing / wrong ABI (build it: ``make`` or ``__graft_entry__.build()``)."""


class MCMCError(RuntimeError):
    """A libmcmc_hip call returned a negative code."""


_ctx_p = C.c_void_p
_stream = C.c_void_p
_ptr = C.c_void_p
_i64, _i32, _u64, _f32 = C.c_int64, C.c_int, C.c_uint64, C.c_float



class TrainBuffers(C.Structure):
    """mcmc_train_buffers (include/mcmc_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("mu", "rho", "sigma", "eps", "W", "WT", "dW", "gmu", "grho", "adam_m", "adam_v", "norm_scratch",
                                          "Y", "delta", "loss", "gnorm", "act")] + [("act_floats", C.c_int64)]


class TrainHyper(C.Structure):
    """mcmc_train_hyper (include/mcmc_hip.h)."""
    _fields_ = [(n, C.c_float) for n in ("p_drop", "alpha", "beta", "beta2", "adam_eps", "weight_decay", "max_norm")]


# name -> (restype, argtypes); must list every symbol include/mcmc_hip.h declares
SIGNATURES = {
    "mcmc_last_error": (C.c_char_p, []),
    "mcmc_abi_version": (_i32, []),
    "mcmc_ctx_create": (_i32, [_i32, C.POINTER(_ctx_p)]),
    "mcmc_ctx_destroy": (_i32, [_ctx_p]),
    "mcmc_param_count": (_i64, [_i32]),
    "mcmc_xpack_bytes": (_i64, [_i64, _i32, _i32]),
    "mcmc_wpack_bytes": (_i64, [_i32, _i32]),
    "mcmc_pack_x": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _ptr, _stream]),
    "mcmc_sample_weights": (_i32, [_ctx_p, _ptr, _ptr, _i32, _i32, _i32, _i64, _i32, _i32, _ptr, _u64, _i32, _ptr, _stream]),
    "mcmc_sample_weights_seedptr": (_i32, [_ctx_p, _ptr, _ptr, _i32, _i32, _i32, _i64, _i32, _ptr, _i32, _ptr, _stream]),
    "mcmc_philox_eps": (_i32, [_ctx_p, _u64, _i64, _i32, _i32, _ptr, _stream]),
    "mcmc_forward": (_i32, [_ctx_p, _i32, _ptr, _i64, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _f32, _ptr, _i64, _stream]),
    "mcmc_forward_dropout": (_i32, [_ctx_p, _i32, _ptr, _i64, _i32, _ptr, _i32, _i32, _i64, _i32, _u64, _f32, _i64, _ptr, _i64, _stream]),
    "mcmc_reduce_update": (_i32, [_ctx_p, _ptr, _i64, _i32, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _stream]),
    "mcmc_reduce_merge": (_i32, [_ctx_p, _i32, _i64, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _stream]),
    "mcmc_reduce_set_edges": (_i32, [_ctx_p, _i64, _i64, _ptr, _ptr, _i32, _f32, _ptr, _ptr, _stream]),
    "mcmc_reduce_finalize": (_i32, [_ctx_p, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, C.POINTER(C.c_double), _i32,
                                    _ptr, _ptr, _ptr, _stream]),
    "mcmc_loss_stats": (_i32, [_ctx_p, _ptr, _ptr, _i64, _ptr, _stream]),
    "mcmc_loss_forward": (_i32, [_ctx_p, _ptr, _ptr, _i64, _i64, _ptr, _f32, _f32, _ptr, _stream]),
    "mcmc_loss_backward": (_i32, [_ctx_p, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _f32, _f32, _f32, _ptr, _stream]),
    "mcmc_loss_workspace_bytes": (_i64, []),
    "mcmc_loss_fused": (_i32, [_ctx_p, _ptr, _ptr, _i64, _f32, _f32, _f32, _ptr, _ptr, _ptr, _stream]),
    "mcmc_loss_fused_static": (_i32, [_ctx_p, _ptr, _ptr, _i64, _f32, _f32, _f32, _ptr, _ptr, _stream]),
    "mcmc_train_sigma": (_i32, [_ctx_p, _ptr, _ptr, _i64, _stream]),
    "mcmc_train_sample": (_i32, [_ctx_p, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr, _u64, _ptr, _ptr, _ptr, _stream]),
    "mcmc_train_forward": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _stream]),
    "mcmc_train_loss": (_i32, [_ctx_p, _ptr, _ptr, _i32, _i32, _i64, _f32, _f32, _ptr, _ptr, _stream]),
    "mcmc_train_backward": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr,
                                   _stream]),
    "mcmc_train_norm_blocks": (_i32, []),
    "mcmc_train_update": (_i32, [_ctx_p, _ptr, _ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32,
                                 _f32, _f32, _f32, _f32, _i64, _ptr, _stream]),
    "mcmc_train_grad": (_i32, [_ctx_p, _i32, _i32, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _stream]),
    "mcmc_train_row_blocks": (_i32, [_i64]),
    "mcmc_train_act_bytes_bf16": (_i64, [_i64]),
    "mcmc_train_forward_bf16": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _i32, _ptr, _ptr, _stream]),
    "mcmc_train_backward_bf16": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _i32, _ptr, _ptr,
                                        _stream]),
    "mcmc_train_backward_blocks": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _i32,
                                          _ptr, _stream]),
    "mcmc_train_apply": (_i32, [_ctx_p, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _f32, _f32,
                                _f32, _i64, _ptr, _stream]),
    "mcmc_train_step": (_i32, [_ctx_p, C.POINTER(TrainBuffers), C.POINTER(TrainHyper), _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _i32, _i32,
                               _ptr, _u64, _i32, _ptr, _ptr, _ptr, _u64, _ptr, _f32, _f32, _i64, _ptr, _stream]),
    "mcmc_train_act_floats": (_i64, [_i64]),
    "mcmc_train_forward_save": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr, _stream]),
    "mcmc_train_backward_saved": (_i32, [_ctx_p, _ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _i32,
                                         _ptr, _ptr, _stream]),
    "mcmc_train_early_stop": (_i32, [_ctx_p, _ptr, _ptr, _ptr, _ptr, _i32, _i64, _stream]),
    "mcmc_tft_param_count": (_i64, [_i32]),
    "mcmc_tft_transpose": (_i32, [_ctx_p, _ptr, _ptr, _i32, _stream]),
    "mcmc_tft_forward": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _stream]),
    "mcmc_tft_act_floats": (_i64, []),
    "mcmc_tft_forward_save": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr,
                                     _stream]),
    "mcmc_tft_backward_saved": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr,
                                       _ptr, _stream]),
    "mcmc_tft_backward": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr,
                                 _stream]),
    "mcmc_tft_backward_loss": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _i32, _i32, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _u64, _f32, _ptr, _ptr,
                                      _f32, _f32, _ptr, _ptr, _ptr, _ptr, _stream]),
    "mcmc_flat_norm_blocks": (_i32, []),
    "mcmc_train_loss_acc": (_i32, [_ctx_p, _ptr, _ptr, _i32, _i32, _i64, _f32, _f32, _ptr, _ptr, _ptr, _stream]),
    "mcmc_tft_adamw": (_i32, [_ctx_p, _ptr, _ptr, _i32, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _ptr,
                              _stream]),
    "mcmc_flat_adamw": (_i32, [_ctx_p, _ptr, _i64, _ptr, _i32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _f32, _f32, _f32, _i64, _ptr,
                               _stream]),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the shared library once and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MCMCLibraryError(f"{LIB_PATH} not found - build it with `make` (hipcc, gfx950); there is no CPU fallback")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise MCMCLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MCMCLibraryError(f"{LIB_PATH} lacks symbol {name}") from e
            fn.restype, fn.argtypes = res, args
        if lib.mcmc_abi_version() != ABI_VERSION:
            raise MCMCLibraryError(f"ABI mismatch: library {lib.mcmc_abi_version()}, binding {ABI_VERSION}")
        _lib = lib
        return lib


def last_error() -> str:
    return load().mcmc_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = last_error()
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise MCMCError(f"{what} failed ({rc}): {msg}")
    return rc


class Context:
    """One per device (include/mcmc_hip.h: mcmc_ctx_create). Not thread-safe."""

    _cache = {}

    def __init__(self, device_index: int):
        self.lib = load()
        self.device_index = device_index
        h = _ctx_p()
        check(self.lib.mcmc_ctx_create(device_index, C.byref(h)), "mcmc_ctx_create")
        self.handle = h

    @classmethod
    def for_device(cls, device_index: int) -> "Context":
        ctx = cls._cache.get(device_index)
        if ctx is None:
            ctx = cls._cache[device_index] = cls(device_index)
        return ctx

    def close(self):
        if self.handle:
            self.lib.mcmc_ctx_destroy(self.handle)
            self.handle = None
