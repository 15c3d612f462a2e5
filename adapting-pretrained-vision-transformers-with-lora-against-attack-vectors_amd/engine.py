"""Host-side engine: one `vl_model` handle on one GPU, driven through the C ABI.

PyTorch is used only as plumbing here -- device allocation (`torch.empty`), the current
HIP stream and zero-copy tensor views of library-owned buffers; every number is
produced by the HIP kernels behind `include/vitlora.h`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import VL_PREC, VL_T, VLConfig, check

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # get_normalization, Utils.py:92-93
IMAGENET_STD = (0.229, 0.224, 0.225)

# HF-4.55.2 module paths of the six Linear layers of one encoder layer (what peft matches on)
LINEAR_MODULES = (
    ("q", "attention.attention.query"), ("k", "attention.attention.key"),
    ("v", "attention.attention.value"), ("o", "attention.output.dense"),
    ("fc1", "intermediate.dense"), ("fc2", "output.dense"),
)
# transformers 5.x names -> the 4.55.2 names of the reference's checkpoints
_HF5_TO_455 = (
    ("vit.layers.", "vit.encoder.layer."), ("attention.q_proj", "attention.attention.query"),
    ("attention.k_proj", "attention.attention.key"), ("attention.v_proj", "attention.attention.value"),
    ("attention.o_proj", "attention.output.dense"), ("mlp.fc1", "intermediate.dense"),
    ("mlp.fc2", "output.dense"),
)


def canonical_key(k: str) -> str:
    if k.startswith("vit.layers."):
        for a, b in _HF5_TO_455:
            k = k.replace(a, b)
    return k


def resolve_targets(target_modules: Iterable[str]) -> Tuple[str, ...]:
    """peft's matching rule (module name == target or endswith '.'+target) applied to
    the HF-4.55.2 ViT module names; ["query","key","value","output.dense"]
    (train_loras.py:81) -> q, k, v, o AND fc2."""
    out = []
    for short, path in LINEAR_MODULES:
        full = "vit.encoder.layer.0." + path
        if any(full == t or full.endswith("." + t) for t in target_modules):
            out.append(short)
    return tuple(out)


@dataclass
class ArchConfig:
    """ViT architecture (defaults = google/vit-base-patch16-224, Utils.py:84-90)."""
    image_size: int = 224
    patch_size: int = 16
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    num_labels: int = 21
    ln_eps: float = 1e-12

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


@dataclass
class LoraSpec:
    r: int = 0
    alpha: float = 16.0
    dropout: float = 0.0
    targets: Tuple[str, ...] = ()
    merged: bool = False


class _DevView:
    """Zero-copy torch view of a device buffer owned by the library."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def _view_f32(ptr: int, shape, device) -> torch.Tensor:
    return torch.as_tensor(_DevView(ptr, shape), device=device)


class Engine:
    """precision: "f16" (default; fp16 operands and 16-bit residual streams, fp32 accumulation -- the MFMA-rate path),
    "bf16" (the same kernels instantiated on bf16: fp32's exponent range, so no gradient ever leaves it; 8 mantissa bits --
    BASELINE config 5 / north_star name this type) or "f32" (every operand and activation fp32: the parity mode held to 1e-3
    against the reference's CPU path)."""

    def __init__(self, arch: ArchConfig, lora: Optional[LoraSpec] = None, device="cuda:0", precision: str = "f16"):
        if not torch.cuda.is_available():
            raise _lib.VitLoraError("no GPU visible: the vitlora engine runs on MI355X only (no CPU fallback)")
        self.lib = _lib.load()
        self.arch = arch
        self.lora = lora or LoraSpec()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        cfg = VLConfig()
        cfg.image_size, cfg.patch_size, cfg.hidden = arch.image_size, arch.patch_size, arch.hidden
        cfg.layers, cfg.heads, cfg.mlp, cfg.num_labels = arch.layers, arch.heads, arch.mlp, arch.num_labels
        cfg.ln_eps = arch.ln_eps
        tb = 0
        for t in self.lora.targets:
            tb |= VL_T[t]
        cfg.lora_r = self.lora.r if tb else 0
        cfg.lora_alpha, cfg.lora_dropout = float(self.lora.alpha), float(self.lora.dropout)
        cfg.lora_targets, cfg.lora_merged = tb, int(self.lora.merged)
        if precision not in VL_PREC:
            raise ValueError(f"precision must be one of {sorted(VL_PREC)}")
        cfg.precision = VL_PREC[precision]
        self.precision = {0: "f16", 1: "f32", 2: "bf16"}[cfg.precision]
        h = C.c_void_p()
        check(self.lib.vl_create(C.byref(cfg), C.byref(h)), "vl_create")
        self.h = h
        self._ws = None
        self._plan = (0, False)
        p, n = C.c_void_p(), C.c_int64()
        check(self.lib.vl_param_flat(self.h, C.byref(p), C.byref(n)), "vl_param_flat")
        self.flat = _view_f32(p.value, (n.value,), self.device)     # LoRA A/B + classifier, fp32 master
        self._loaded = set()

    # -- plumbing ---------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.vl_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    # -- weights ----------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """HF state dict (4.55.2 keys of the reference's .pth, whitebox_attacks.py:94, or 5.x keys)."""
        expected = set(expected_keys(self.arch))
        seen = set()
        for k, v in sd.items():
            ck = canonical_key(k)
            if ck not in expected:
                if strict:
                    raise KeyError(f"unexpected key in state_dict: {k}")
                continue
            t = self._f32(v)
            check(self.lib.vl_load_tensor(self.h, ck.encode(), C.c_void_p(t.data_ptr()), t.numel(), self._stream()),
                  f"vl_load_tensor({ck})")
            seen.add(ck)
        torch.cuda.current_stream(self.device).synchronize()   # staging tensors may be freed now
        missing = expected - seen - self._loaded
        self._loaded |= seen
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {sorted(missing)[:5]} ...")
        return sorted(missing)

    def param(self, layer: int, target: str, which: str) -> torch.Tensor:
        """View of lora_A [r,in] / lora_B [out,r] (layer >= 0) or the classifier (layer = -1,
        which = 'weight' | 'bias') inside the flat master buffer."""
        p, n = C.c_void_p(), C.c_int64()
        if layer < 0:
            check(self.lib.vl_param_tensor(self.h, -1, 0, 0 if which == "weight" else 1, C.byref(p), C.byref(n)))
            shape = (self.arch.num_labels, self.arch.hidden) if which == "weight" else (self.arch.num_labels,)
        else:
            check(self.lib.vl_param_tensor(self.h, layer, VL_T[target], 0 if which == "A" else 1, C.byref(p), C.byref(n)))
            o, k = module_shape(self.arch, target)
            shape = (self.lora.r, k) if which == "A" else (o, self.lora.r)
        return _view_f32(p.value, shape, self.device)

    def commit(self):
        """Explicit vl_lora_commit.  Not needed for correctness: the library commits by itself before a forward /
        attack whenever the parameters changed (param(), adam_step() and mark_dirty() tell it so)."""
        check(self.lib.vl_lora_commit(self.h, self._stream()), "vl_lora_commit")

    def mark_dirty(self):
        """The flat parameters were written through a view kept from earlier (self.flat, a broadcast, ...)."""
        check(self.lib.vl_params_changed(self.h), "vl_params_changed")

    def set_dead_rows(self, on: bool):
        """Eval-mode forwards compute the last encoder layer on the CLS rows only (default, exact: nothing else reaches the
        classifier); off = every row of every layer (tests that read the last layer's saved activations)."""
        self.set_option("dead_rows", on)

    def set_option(self, name: str, value) -> None:
        """Diagnostic switches of the handle ("dead_rows", "fuse_pgd": include/vitlora.h)."""
        check(self.lib.vl_debug_set_option(self.h, name.encode(), int(value)), "vl_debug_set_option")

    def check(self):
        """Synchronise the current stream and raise what the kernels flagged: VitLoraError (bad label) or
        _lib.NonFiniteGradient (fp16 mode: a gradient left the fp16 range -- skip that step or redo the batch in f32)."""
        check(self.lib.vl_check_errors(self.h, self._stream()), "vl_check_errors")

    def counter(self, what: str) -> int:
        v = C.c_int64()
        check(self.lib.vl_debug_counter(self.h, what.encode(), C.byref(v)), "vl_debug_counter")
        return int(v.value)

    def set_normalization(self, mean, std):
        key = (tuple(float(v) for v in mean), tuple(float(v) for v in std))
        if getattr(self, "_norm_key", None) == key:
            return                               # unchanged: keep the cached PGD graph
        self._norm_key = key
        m = (C.c_float * 3)(*[float(v) for v in mean])
        s = (C.c_float * 3)(*[float(v) for v in std])
        check(self.lib.vl_set_normalization(self.h, m, s), "vl_set_normalization")

    # -- workspace --------------------------------------------------------------------------
    def plan(self, max_batch: int, train: bool = False):
        if self._ws is not None and self._plan[0] >= max_batch and (self._plan[1] or not train):
            return
        max_batch = max(max_batch, self._plan[0])
        train = train or self._plan[1]
        n = C.c_size_t()
        check(self.lib.vl_plan(self.h, max_batch, int(train), C.byref(n)), "vl_plan")
        self._ws = None
        self._ws = torch.empty(n.value + 256, dtype=torch.uint8, device=self.device)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        check(self.lib.vl_set_workspace(self.h, C.c_void_p(base), n.value), "vl_set_workspace")
        self._plan = (max_batch, train)

    def workspace_bytes(self, max_batch: int, train: bool = False) -> int:
        n = C.c_size_t()
        check(self.lib.vl_plan(self.h, max_batch, int(train), C.byref(n)), "vl_plan")
        if self._ws is not None:   # re-arm the live plan record
            check(self.lib.vl_plan(self.h, self._plan[0], int(self._plan[1]), C.byref(C.c_size_t())))
        return n.value

    # -- compute ----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, normalise: bool = False, train: bool = False) -> torch.Tensor:
        x = self._check_images(x)
        B = x.shape[0]
        self.plan(B, train)
        logits = torch.empty(B, self.arch.num_labels, dtype=torch.float32, device=self.device)
        check(self.lib.vl_forward(self.h, C.c_void_p(x.data_ptr()), B, int(normalise), int(train),
                                  C.c_void_p(logits.data_ptr()), self._stream()), "vl_forward")
        self._keep = x
        return logits

    def loss_ce(self, labels: torch.Tensor) -> torch.Tensor:
        labels = labels.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        check(self.lib.vl_loss_ce(self.h, C.c_void_p(labels.data_ptr()), C.c_void_p(out.data_ptr()), self._stream()),
              "vl_loss_ce")
        self._keep_labels = labels
        return out[0]

    def set_dlogits(self, dlogits: torch.Tensor):
        d = self._f32(dlogits)
        check(self.lib.vl_set_dlogits(self.h, C.c_void_p(d.data_ptr()), self._stream()), "vl_set_dlogits")

    def backward(self, want_input: bool, want_params: bool, batch_shape=None):
        gx = gp = None
        if want_input:
            gx = torch.empty(batch_shape, dtype=torch.float32, device=self.device)
        if want_params:
            gp = torch.empty_like(self.flat)
        check(self.lib.vl_backward(self.h, C.c_void_p(gx.data_ptr() if gx is not None else 0),
                                   C.c_void_p(gp.data_ptr() if gp is not None else 0), self._stream()), "vl_backward")
        return gx, gp

    def pgd_step(self, adv, x0, grad, eps, alpha, lo=0.0, hi=1.0):
        check(self.lib.vl_pgd_step(C.c_void_p(adv.data_ptr()), C.c_void_p(x0.data_ptr()), C.c_void_p(grad.data_ptr()),
                                   float(eps), float(alpha), float(lo), float(hi), adv.numel(), self._stream()),
              "vl_pgd_step")

    def pgd_init(self, adv, x0, eps, seed, lo=0.0, hi=1.0):
        check(self.lib.vl_pgd_init(C.c_void_p(adv.data_ptr()), C.c_void_p(x0.data_ptr()), float(eps), float(lo),
                                   float(hi), int(seed), adv.numel(), self._stream()), "vl_pgd_init")

    def pgd_attack(self, x0: torch.Tensor, labels: torch.Tensor, eps, alpha, steps, random_start=True, seed=0,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x0 = self._check_images(x0)
        labels = labels.to(device=self.device, dtype=torch.int64).contiguous()
        B = x0.shape[0]
        self.plan(B, False)
        adv = out if out is not None else torch.empty_like(x0)
        check(self.lib.vl_pgd_attack(self.h, C.c_void_p(x0.data_ptr()), C.c_void_p(labels.data_ptr()), B, float(eps),
                                     float(alpha), int(steps), int(bool(random_start)), int(seed),
                                     C.c_void_p(adv.data_ptr()), self._stream()), "vl_pgd_attack")
        self._keep, self._keep_labels = x0, labels
        return adv

    def channel_affine(self, x: torch.Tensor, scale, shift, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self._check_images(x)
        out = torch.empty_like(x) if out is None else out
        s = (C.c_float * 3)(*[float(v) for v in scale])
        t = (C.c_float * 3)(*[float(v) for v in shift])
        check(self.lib.vl_channel_affine(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), s, t, x.shape[0],
                                         x.shape[2] * x.shape[3], self._stream()), "vl_channel_affine")
        return out

    def adam_step(self, param, grad, m1, m2, lr, b1, b2, eps, t):
        check(self.lib.vl_adam_step(C.c_void_p(param.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(m1.data_ptr()),
                                    C.c_void_p(m2.data_ptr()), float(lr), float(b1), float(b2), float(eps), int(t),
                                    param.numel(), self._stream()), "vl_adam_step")

    # -- adversarial patch (patch_attack.py) ---------------------------------------------------------
    def patch_apply(self, images: torch.Tensor, patch: torch.Tensor, inv_affine: torch.Tensor, patch_type: int,
                    out: Optional[torch.Tensor] = None, persp: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`persp` [B, 8]: torchvision perspective coefficients per image (distortion_scale_max > 0); None = no distortion."""
        images = self._check_images(images)
        patch = self._f32(patch)
        mats = self._f32(inv_affine)
        out = torch.empty_like(images) if out is None else out
        if persp is None:
            check(self.lib.vl_patch_apply(C.c_void_p(images.data_ptr()), C.c_void_p(patch.data_ptr()), C.c_void_p(mats.data_ptr()),
                                          images.shape[0], images.shape[2], patch.shape[-1], int(patch_type),
                                          C.c_void_p(out.data_ptr()), self._stream()), "vl_patch_apply")
        else:
            q = self._persp(persp, images.shape[0])
            check(self.lib.vl_patch_apply_persp(C.c_void_p(images.data_ptr()), C.c_void_p(patch.data_ptr()),
                                                C.c_void_p(mats.data_ptr()), C.c_void_p(q.data_ptr()), images.shape[0],
                                                images.shape[2], patch.shape[-1], int(patch_type), C.c_void_p(out.data_ptr()),
                                                self._stream()), "vl_patch_apply_persp")
        return out

    def patch_grad(self, grad_out: torch.Tensor, inv_affine: torch.Tensor, patch_size: int, patch_type: int,
                   persp: Optional[torch.Tensor] = None) -> torch.Tensor:
        g = self._f32(grad_out)
        mats = self._f32(inv_affine)
        dp = torch.empty(3, patch_size, patch_size, dtype=torch.float32, device=self.device)
        if persp is None:
            check(self.lib.vl_patch_grad(C.c_void_p(g.data_ptr()), C.c_void_p(mats.data_ptr()), g.shape[0], g.shape[2],
                                         int(patch_size), int(patch_type), C.c_void_p(dp.data_ptr()), self._stream()), "vl_patch_grad")
        else:
            q = self._persp(persp, g.shape[0])
            check(self.lib.vl_patch_grad_persp(C.c_void_p(g.data_ptr()), C.c_void_p(mats.data_ptr()), C.c_void_p(q.data_ptr()),
                                               g.shape[0], g.shape[2], int(patch_size), int(patch_type), C.c_void_p(dp.data_ptr()),
                                               self._stream()), "vl_patch_grad_persp")
        return dp

    def _persp(self, persp: torch.Tensor, batch: int) -> torch.Tensor:
        q = self._f32(persp)
        if tuple(q.shape) != (batch, 8):
            raise ValueError(f"persp must be [{batch}, 8] (one set of perspective coefficients per image), got {tuple(q.shape)}")
        return q

    def clamp_(self, x: torch.Tensor, lo: float, hi: float):
        check(self.lib.vl_clamp(C.c_void_p(x.data_ptr()), float(lo), float(hi), x.numel(), self._stream()), "vl_clamp")
        return x

    def set_dropout_seed(self, seed: int):
        check(self.lib.vl_set_dropout_seed(self.h, int(seed)), "vl_set_dropout_seed")

    def dropout_mask(self, layer: int, proj: str, batch: int) -> torch.Tensor:
        """Keep-mask (0 or 1/(1-p)) the last train-mode forward used for `proj` in
        {"qkv","o","fc1","fc2"} of `layer`: [batch, tokens, in]."""
        pi = {"qkv": 0, "o": 1, "fc1": 2, "fc2": 3}[proj]
        cols = self.arch.mlp if proj == "fc2" else self.arch.hidden
        out = torch.empty(batch, self.arch.tokens, cols, dtype=torch.float32, device=self.device)
        check(self.lib.vl_dropout_mask(self.h, layer, pi, C.c_void_p(out.data_ptr()), self._stream()), "vl_dropout_mask")
        return out

    def quantize_u8(self, images: torch.Tensor) -> torch.Tensor:
        images = self._f32(images)
        B, Cn, H, W = images.shape
        out = torch.empty(B, H, W, Cn, dtype=torch.uint8, device=self.device)
        check(self.lib.vl_quantize_u8(C.c_void_p(images.data_ptr()), C.c_void_p(out.data_ptr()), B, Cn, H, W,
                                      self._stream()), "vl_quantize_u8")
        return out

    def debug_tensor(self, what: str, layer: int) -> torch.Tensor:
        p, n, dt = C.c_void_p(), C.c_int64(), C.c_int()
        check(self.lib.vl_debug_tensor(self.h, what.encode(), layer, C.byref(p), C.byref(n), C.byref(dt)))
        if dt.value == 0:
            return _view_f32(p.value, (n.value,), self.device)
        v = torch.as_tensor(_DevView(p.value, (n.value,), "<i2"), device=self.device)
        return v.view(torch.bfloat16 if dt.value == 2 else torch.float16)

    def _check_images(self, x: torch.Tensor) -> torch.Tensor:
        S = self.arch.image_size
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != S or x.shape[3] != S:
            raise ValueError(f"expected images of shape [B,3,{S},{S}], got {tuple(x.shape)}")
        return self._f32(x)


def module_shape(arch: ArchConfig, target: str) -> Tuple[int, int]:
    if target == "fc1":
        return arch.mlp, arch.hidden
    if target == "fc2":
        return arch.hidden, arch.mlp
    return arch.hidden, arch.hidden


def expected_keys(arch: ArchConfig):
    keys = ["vit.embeddings.cls_token", "vit.embeddings.position_embeddings",
            "vit.embeddings.patch_embeddings.projection.weight", "vit.embeddings.patch_embeddings.projection.bias",
            "vit.layernorm.weight", "vit.layernorm.bias", "classifier.weight", "classifier.bias"]
    for i in range(arch.layers):
        p = f"vit.encoder.layer.{i}."
        for _, path in LINEAR_MODULES:
            keys += [p + path + ".weight", p + path + ".bias"]
        for ln in ("layernorm_before", "layernorm_after"):
            keys += [p + ln + ".weight", p + ln + ".bias"]
    return keys
