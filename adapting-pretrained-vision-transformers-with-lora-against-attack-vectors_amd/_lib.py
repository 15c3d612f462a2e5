"""ctypes binding of libvitlora_hip.so (C ABI: include/vitlora.h).

The library is the product path.  There is no CPU fallback: if the shared object is
missing or does not export every symbol of the header, importing callers fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITLORA_LIB: another build of the same library (kernel A/B runs on one GPU box); never a fallback
LIB_PATH = os.environ.get("VITLORA_LIB") or os.path.join(_HERE, "libvitlora_hip.so")

VL_T = {"q": 1, "k": 2, "v": 4, "o": 8, "fc1": 16, "fc2": 32}
VL_PREC = {"f16": 0, "fp16": 0, "f32": 1, "fp32": 1, "bf16": 2}


class VLConfig(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("hidden", C.c_int32),
        ("layers", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32),
        ("num_labels", C.c_int32), ("ln_eps", C.c_float),
        ("lora_r", C.c_int32), ("lora_alpha", C.c_float), ("lora_dropout", C.c_float),
        ("lora_targets", C.c_uint32), ("lora_merged", C.c_int32),
        ("precision", C.c_int32),          # VL_PREC_F16 = 0 (fp16 operands, fp32 accumulate), VL_PREC_F32 = 1 (parity mode), VL_PREC_BF16 = 2 (bf16 operands)
        ("reserved", C.c_int32 * 3),
    ]


# name -> (restype, argtypes); mirrors include/vitlora.h one to one
SIGNATURES = {
    "vl_version": (C.c_char_p, []),
    "vl_last_error": (C.c_char_p, []),
    "vl_create": (C.c_int, [C.POINTER(VLConfig), C.POINTER(C.c_void_p)]),
    "vl_destroy": (C.c_int, [C.c_void_p]),
    "vl_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vl_param_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vl_param_flat": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vl_lora_commit": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vl_params_changed": (C.c_int, [C.c_void_p]),
    "vl_debug_counter": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "vl_merge_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "vl_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vl_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_loss_ce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_set_dlogits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_set_normalization": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vl_channel_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_void_p]),
    "vl_backward_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_backward_lora": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_pgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_void_p]),
    "vl_pgd_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_int64, C.c_void_p]),
    "vl_pgd_attack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "vl_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int64, C.c_void_p]),
    "vl_quantize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vl_patch_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_patch_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_patch_apply_persp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_patch_grad_persp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_clamp": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p]),
    "vl_set_dropout_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "vl_dropout_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_bench_gemm": (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_float)]),
    "vl_check_gemm": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_float)]),
    "vl_debug_set_gemm_pp": (C.c_int, [C.c_int]),
    "vl_debug_set_gemm_stream": (C.c_int, [C.c_int]),
    "vl_debug_set_cus": (C.c_int, [C.c_void_p, C.c_int]),
    "vl_check_errors": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vl_swin_check_errors": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vl_debug_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vl_profile_begin": (C.c_int, []),
    "vl_profile_report": (C.c_int, [C.c_char_p, C.c_size_t]),
    "vl_debug_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
}

class VLSwinConfig(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("patch_size", C.c_int32), ("embed_dim", C.c_int32), ("depths", C.c_int32 * 4),
                ("heads", C.c_int32 * 4), ("window", C.c_int32), ("num_labels", C.c_int32), ("ln_eps", C.c_float),
                ("lora_r", C.c_int32), ("lora_alpha", C.c_float), ("lora_targets", C.c_uint32), ("reserved", C.c_int32 * 4)]


SIGNATURES.update({
    "vl_swin_create": (C.c_int, [C.POINTER(VLSwinConfig), C.POINTER(C.c_void_p)]),
    "vl_swin_destroy": (C.c_int, [C.c_void_p]),
    "vl_swin_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vl_swin_param_flat": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vl_swin_param_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vl_swin_set_normalization": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vl_swin_plan": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "vl_swin_set_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "vl_swin_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vl_swin_loss_ce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_swin_backward_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vl_swin_pgd_attack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64,
                                     C.c_void_p, C.c_void_p]),
})

_lib = None


class VitLoraError(RuntimeError):
    code = 0


class NonFiniteGradient(VitLoraError):
    """VL_ERR_NONFINITE: the fp16 backward left its range (or produced NaN) in an earlier call."""
    code = -5


def load():
    """Load the shared library (once).  Raises if it is not built: `python __graft_entry__.py`
    or `<package>/csrc/build.sh` builds it with hipcc for gfx950."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitLoraError(
            f"{LIB_PATH} is missing: the HIP extension is not built. There is no CPU fallback; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vl_last_error().decode("utf-8", "replace")
        exc = NonFiniteGradient if rc == -5 else VitLoraError
        e = exc(f"{what or 'vitlora call'} failed ({rc}): {msg}")
        e.code = rc
        raise e
