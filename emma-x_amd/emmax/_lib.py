"""
_lib.py -- ctypes binding of libemmax_hip.so (the C ABI declared in include/emmax.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is raised.  The
library is built in-tree by `__graft_entry__.build()` / `make -C emma-x_amd/csrc`.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libemmax_hip.so")

_c_i32p = C.POINTER(C.c_int32)
_c_i64p = C.POINTER(C.c_int64)
_c_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


class EmmaxError(RuntimeError):
    pass


class TowerConfigC(C.Structure):
    _fields_ = [
        ("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32), ("mlp_hidden", C.c_int32),
        ("has_cls", C.c_int32), ("n_reg", C.c_int32), ("layerscale", C.c_int32),
        ("patch", C.c_int32), ("image_size", C.c_int32), ("take_index", C.c_int32),
        ("ln_eps", C.c_float), ("mean", C.c_float * 3), ("std", C.c_float * 3),
    ]


class ConfigC(C.Structure):
    _fields_ = [
        ("tower", TowerConfigC * 2),
        ("hidden", C.c_int32), ("inter", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
        ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("vocab", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float),
        ("bos_id", C.c_int32), ("eos_id", C.c_int32), ("pad_id", C.c_int32), ("decode_fp8", C.c_int32),
    ]


ABI_VERSION = 5   # EMMAX_ABI_VERSION of include/emmax.h this binding was written against

# name -> (restype, argtypes): exactly the entry points of include/emmax.h
SIGNATURES = {
    "emmax_version": (C.c_char_p, []),
    "emmax_last_error": (C.c_char_p, []),
    "emmax_abi_version": (C.c_int, []),
    "emmax_config_size": (C.c_int, []),
    "emmax_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
    "emmax_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "emmax_model_create": (C.c_int, [C.POINTER(ConfigC), C.POINTER(_vp)]),
    "emmax_model_destroy": (None, [_vp]),
    "emmax_model_bind_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int, _c_i64p, C.c_int]),
    "emmax_model_arena_bytes": (C.c_int64, [_vp]),
    "emmax_model_max_decode_batch": (C.c_int, [_vp]),
    "emmax_model_finalize": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "emmax_model_aux_bytes": (C.c_int64, [_vp]),
    "emmax_model_build_aux": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "emmax_session_bytes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _c_i64p, _c_i64p]),
    "emmax_session_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp, C.c_int64, C.POINTER(_vp)]),
    "emmax_session_bytes_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _c_i64p, _c_i64p]),
    "emmax_session_create_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp, C.c_int64, C.POINTER(_vp)]),
    "emmax_session_stage_rows": (C.c_int, [_vp]),
    "emmax_session_destroy": (None, [_vp]),
    "emmax_vision_encode": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "emmax_vision_encode_pixels": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "emmax_vision_features": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "emmax_prefill": (C.c_int, [_vp, _vp, _c_i32p, C.c_int, C.c_int, _vp, _vp]),
    "emmax_prefill_text": (C.c_int, [_vp, _vp, _c_i32p, C.c_int, C.c_int, _vp]),
    "emmax_prefill_logits": (C.c_int, [_vp, _vp, _vp]),
    "emmax_last_logits": (C.c_int, [_vp, _vp, _vp]),
    "emmax_decode_step": (C.c_int, [_vp, _vp]),
    "emmax_set_current_tokens": (C.c_int, [_vp, _vp, _vp]),
    "emmax_generate": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "emmax_session_graph_active": (C.c_int, [_vp]),
    "emmax_profile_decode_stage": (C.c_int, [_vp, C.c_int, C.c_int, _c_f32p, _vp]),
    "emmax_session_set_stop": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "emmax_slots_open": (C.c_int, [_vp, C.c_int, _vp]),
    "emmax_slot_prefill": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp]),
    "emmax_slots_prefill": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "emmax_slots_prefill_staged": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "emmax_slots_commit": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "emmax_slots_step": (C.c_int, [_vp, C.c_int, _vp]),
    "emmax_slots_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "emmax_slot_output": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp]),
    "emmax_slot_release": (C.c_int, [_vp, C.c_int, _vp]),
    "emmax_op_gemm": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp,
                                C.c_int, C.c_int, _vp]),
    "emmax_op_gemm_splitk": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp,
                                       C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp]),
    "emmax_gemm_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_char_p, C.c_int]),
    "emmax_op_gemm_ln": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_float, C.c_int, _vp, _vp, _vp, _vp]),
    "emmax_op_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "emmax_op_rmsnorm": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "emmax_op_attention": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_float, C.c_int, _vp]),
    "emmax_op_decode_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.POINTER(C.c_int), _vp]),
    "emmax_op_decode_attention_direct": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "emmax_op_decode_attention_kv8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_float, _vp]),
    "emmax_op_gemv": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_op_resize_bicubic_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "emmax_op_quant_fm8": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp]),
    "emmax_op_gemm_small_fp8": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_op_quant_rm8": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp]),
    "emmax_op_gemv_fp8": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_op_repack_fm": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "emmax_op_gemm_small": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_op_repack_km": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_op_gemm_small_km": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "emmax_session_exact": (C.c_int, [_vp]),
    "emmax_op_x_gemm": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int64, _vp]),
    "emmax_op_x_rownorm": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp]),
    "emmax_op_x_attention": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp]),
    "emmax_op_x_join": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "emmax_op_x_decode_attention": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libemmax_hip.so and attach the signatures; raises EmmaxError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise EmmaxError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C emma-x_amd/csrc`. There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.emmax_abi_version() != ABI_VERSION:
        raise EmmaxError(f"ABI mismatch: library reports {lib.emmax_abi_version()}, host expects {ABI_VERSION}")
    if lib.emmax_config_size() != C.sizeof(ConfigC):
        raise EmmaxError(f"emmax_config is {lib.emmax_config_size()} bytes in the library, {C.sizeof(ConfigC)} in this binding")
    _lib = lib
    return lib


def tuning_set(name: str, value: int) -> None:
    """Set one of the library's tuning switches (include/emmax.h: emmax_tuning_set); the environment is only read once, at start-up."""
    check(load().emmax_tuning_set(name.encode(), int(value)), f"emmax_tuning_set({name})")


def tuning_get(name: str) -> int:
    v = C.c_int(0)
    check(load().emmax_tuning_get(name.encode(), C.byref(v)), f"emmax_tuning_get({name})")
    return int(v.value)


def gemm_plan(M: int, N: int, K: int, act: int = 0, out_f32: bool = False, ln: bool = False, residual: int = 0, norm: bool = False,
              ws_bytes: int = 64 << 20) -> str:
    """The GEMM launch plan for a problem, as text (include/emmax.h: emmax_gemm_plan; host only -- runs without a GPU)."""
    buf = C.create_string_buffer(256)
    check(load().emmax_gemm_plan(M, N, K, act, int(out_f32), int(ln), int(residual), int(norm), ws_bytes, buf, 256), "emmax_gemm_plan")
    return buf.value.decode()


class tuning:
    """`with tuning(graph=1, ks=0): ...` -- switch for the duration of a block (tests / A-B measurements), restoring the old values."""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = tuning_get(k)
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tuning_set(k, v)
        return False


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().emmax_last_error().decode("utf-8", "replace")
        raise EmmaxError(f"{what or 'emmax call'} failed with status {status}: {msg}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor as an int for c_void_p arguments (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
