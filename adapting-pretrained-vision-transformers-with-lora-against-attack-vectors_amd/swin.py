"""Swin Transformer (Swin-T) + LoRA on the HIP library: host-side binding of the vl_swin_* C ABI (include/vitlora.h).

BASELINE config 4 ("Swin-T + LoRA r=16, PGD-40, windowed-attention HIP kernel path").  The reference only lists the model
(README.md:53); the architecture and the state-dict keys are HF's `SwinForImageClassification` (transformers 4.55.2 names,
5.x names are mapped).  fp32 on the exact-f32 MFMA; adapters are applied in eval mode (the attack path)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import VL_T, VLSwinConfig, check
from .engine import _view_f32


# transformers 5.x module names -> the 4.55.2 names of the reference's pin
_HF5_TO_455 = (("attention.q_proj", "attention.self.query"), ("attention.k_proj", "attention.self.key"),
               ("attention.v_proj", "attention.self.value"), ("attention.o_proj", "attention.output.dense"),
               ("attention.relative_position_bias.relative_position_bias_table", "attention.self.relative_position_bias_table"),
               ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense"))


def canonical_swin_key(k: str) -> str:
    for a, b in _HF5_TO_455:
        k = k.replace(a, b)
    return k


@dataclass
class SwinArch:
    """Defaults = HF SwinConfig() = swin-tiny-patch4-window7-224."""
    image_size: int = 224
    patch_size: int = 4
    embed_dim: int = 96
    depths: Tuple[int, int, int, int] = (2, 2, 6, 2)
    heads: Tuple[int, int, int, int] = (3, 6, 12, 24)
    window: int = 7
    num_labels: int = 21
    ln_eps: float = 1e-5


class SwinEngine:
    def __init__(self, arch: Optional[SwinArch] = None, lora_r: int = 0, lora_alpha: float = 16.0, lora_targets=(), device="cuda:0",
                 precision: str = "f32"):
        """precision: "f32" (every operand fp32, exact-f32 MFMA) or "f16" (h16 operands with fp32 accumulation, h16 residual stream
        and gradient stream, windowed attention on the 16x16x32 MFMA, 16-bit patch embedding and patch merging; LayerNorm statistics,
        the mean-pool head and the classifier stay fp32; `pgd_attack` runs batches of >= 32 images as two half-batch chains)."""
        if precision not in ("f32", "f16"):
            raise ValueError("precision must be 'f32' or 'f16'")
        self.precision = precision
        if not torch.cuda.is_available():
            raise _lib.VitLoraError("no GPU visible: the Swin path runs on MI355X only (no CPU fallback)")
        self.lib = _lib.load()
        self.arch = arch or SwinArch()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        cfg = VLSwinConfig()
        a = self.arch
        cfg.image_size, cfg.patch_size, cfg.embed_dim, cfg.window = a.image_size, a.patch_size, a.embed_dim, a.window
        cfg.depths = (C.c_int32 * 4)(*a.depths)
        cfg.heads = (C.c_int32 * 4)(*a.heads)
        cfg.num_labels, cfg.ln_eps = a.num_labels, a.ln_eps
        tb = 0
        for t in lora_targets:
            tb |= VL_T[t]
        cfg.lora_r, cfg.lora_alpha, cfg.lora_targets = (int(lora_r) if tb else 0), float(lora_alpha), tb
        self.lora_r, self.lora_targets = cfg.lora_r, tuple(lora_targets)
        cfg.reserved = (C.c_int32 * 4)(1 if precision == "f16" else 0, 0, 0, 0)
        h = C.c_void_p()
        check(self.lib.vl_swin_create(C.byref(cfg), C.byref(h)), "vl_swin_create")
        self.h = h
        self._ws = None
        self._plan = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.vl_swin_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            if k.endswith("relative_position_index"):
                continue                          # a buffer of HF's module, recomputed on the device
            t = self._f32(v)
            check(self.lib.vl_swin_load_tensor(self.h, canonical_swin_key(k).encode(), C.c_void_p(t.data_ptr()), t.numel(), self._stream()),
                  f"vl_swin_load_tensor({k})")
        torch.cuda.current_stream(self.device).synchronize()

    def param(self, stage: int, block: int, target: str, which: str) -> torch.Tensor:
        p, n = C.c_void_p(), C.c_int64()
        check(self.lib.vl_swin_param_tensor(self.h, stage, block, VL_T[target], 0 if which == "A" else 1, C.byref(p), C.byref(n)))
        c = self.arch.embed_dim << stage
        o, k = (4 * c, c) if target == "fc1" else (c, 4 * c) if target == "fc2" else (c, c)
        return _view_f32(p.value, (self.lora_r, k) if which == "A" else (o, self.lora_r), self.device)

    def plan(self, max_batch: int):
        if self._ws is not None and self._plan >= max_batch:
            return
        n = C.c_size_t()
        check(self.lib.vl_swin_plan(self.h, max_batch, C.byref(n)), "vl_swin_plan")
        self._ws = None
        self._ws = torch.empty(n.value + 256, dtype=torch.uint8, device=self.device)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        check(self.lib.vl_swin_set_workspace(self.h, C.c_void_p(base), n.value), "vl_swin_set_workspace")
        self._plan = max_batch

    def forward(self, x: torch.Tensor, normalise: bool = False) -> torch.Tensor:
        x = self._f32(x)
        self.plan(x.shape[0])
        logits = torch.empty(x.shape[0], self.arch.num_labels, dtype=torch.float32, device=self.device)
        check(self.lib.vl_swin_forward(self.h, C.c_void_p(x.data_ptr()), x.shape[0], int(normalise), C.c_void_p(logits.data_ptr()),
                                       self._stream()), "vl_swin_forward")
        self._keep = x
        return logits

    def loss_ce(self, labels: torch.Tensor) -> torch.Tensor:
        labels = labels.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        check(self.lib.vl_swin_loss_ce(self.h, C.c_void_p(labels.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()), "vl_swin_loss_ce")
        self._keep_labels = labels
        return out[0]

    def backward_input(self, shape) -> torch.Tensor:
        gx = torch.empty(shape, dtype=torch.float32, device=self.device)
        check(self.lib.vl_swin_backward_input(self.h, C.c_void_p(gx.data_ptr()), self._stream()), "vl_swin_backward_input")
        return gx

    def pgd_attack(self, x0, labels, eps, alpha, steps, random_start=True, seed=0) -> torch.Tensor:
        x0 = self._f32(x0)
        labels = labels.to(device=self.device, dtype=torch.int64).contiguous()
        self.plan(x0.shape[0])
        adv = torch.empty_like(x0)
        check(self.lib.vl_swin_pgd_attack(self.h, C.c_void_p(x0.data_ptr()), C.c_void_p(labels.data_ptr()), x0.shape[0], float(eps),
                                          float(alpha), int(steps), int(bool(random_start)), int(seed), C.c_void_p(adv.data_ptr()),
                                          self._stream()), "vl_swin_pgd_attack")
        return adv

    def check(self):
        """Synchronise the current stream and raise what the kernels flagged: VitLoraError (bad label) or
        _lib.NonFiniteGradient (fp16 mode: a gradient left the fp16 range -- redo the batch with precision="f32")."""
        check(self.lib.vl_swin_check_errors(self.h, self._stream()), "vl_swin_check_errors")


__all__ = ["SwinArch", "SwinEngine", "canonical_swin_key"]
