"""peft-shaped LoRA surface used by the reference (train_loras.py:79-95, 343, 419;
eval_compose.py:99-110), implemented on the HIP engine.

    LoraConfig(task_type, inference_mode, r, lora_alpha, lora_dropout, target_modules)
    get_peft_model(model, config) -> PeftModel        .base_model(pixel_values=x).logits
    PeftModel.print_trainable_parameters()            .save_pretrained(dir)
    PeftModel.from_pretrained(base_model, dir)        .merge_and_unload()
    setup_peft_lora(model, rank, alpha, dropout, target_modules)

peft itself is not installable here ("parity unpinned"): the adapter file layout follows
peft 0.15's published format (adapter_config.json + adapter_model.safetensors, keys
`base_model.model.<module>.lora_A.weight` / `.lora_B.weight` and, for task_type SEQ_CLS,
`base_model.model.classifier.{weight,bias}`), the parameter counts are checked against the
known answers printed in infLora.ipynb:163,919.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass, field
from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import torch

from .engine import LINEAR_MODULES, ArchConfig, LoraSpec, module_shape, resolve_targets
from .model import ViTForImageClassification


class TaskType:
    SEQ_CLS = "SEQ_CLS"
    FEATURE_EXTRACTION = "FEATURE_EXTRACTION"


@dataclass
class LoraConfig:
    task_type: Optional[str] = None
    inference_mode: bool = False
    r: int = 8
    lora_alpha: float = 8
    lora_dropout: float = 0.0
    target_modules: Optional[Union[List[str], str]] = None
    bias: str = "none"
    modules_to_save: Optional[List[str]] = None
    peft_type: str = "LORA"
    base_model_name_or_path: Optional[str] = "google/vit-base-patch16-224"

    def targets(self):
        tm = ["query", "value"] if self.target_modules is None else self.target_modules
        if isinstance(tm, str):
            tm = [tm]
        return resolve_targets(tm)

    def classifier_trainable(self) -> bool:
        # peft adds the head to modules_to_save for SEQ_CLS; the notebook does it by hand (infLora.ipynb:178)
        return self.task_type == TaskType.SEQ_CLS or bool(self.modules_to_save and "classifier" in self.modules_to_save)


def _module_path(layer: int, target: str) -> str:
    return f"vit.encoder.layer.{layer}." + dict(LINEAR_MODULES)[target]


class PeftModel(torch.nn.Module):
    """Wraps a vitlora ViT whose engine carries the adapters."""

    def __init__(self, vit: ViTForImageClassification, config: LoraConfig):
        super().__init__()
        self.peft_config = {"default": config}
        self.config = config
        self._vit = vit
        # `.base_model(pixel_values=x).logits` (train_loras.py:310) and `.base_model.model` as in peft
        self.base_model = _BaseModelProxy(vit)

    # -- construction ---------------------------------------------------------------------------
    @staticmethod
    def _attach(base: ViTForImageClassification, config: LoraConfig, init_B_zero=True, seed: Optional[int] = None):
        spec = LoraSpec(r=int(config.r), alpha=float(config.lora_alpha), dropout=float(config.lora_dropout),
                        targets=config.targets(), merged=False)
        vit = ViTForImageClassification(base.arch, spec, device=base._device, precision=base.precision)
        vit.load_state_dict(base.state_dict(), strict=False)
        eng = vit._engine()
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        for i in range(base.arch.layers):
            for t in spec.targets:
                o, k = module_shape(base.arch, t)
                # peft: lora_A kaiming_uniform_(a=sqrt(5)) -> U(-1/sqrt(in), 1/sqrt(in)); lora_B zeros
                bound = 1.0 / math.sqrt(k)
                A = (torch.rand(spec.r, k, generator=g) * 2 - 1) * bound
                eng.param(i, t, "A").copy_(A)
                if init_B_zero:
                    eng.param(i, t, "B").zero_()
        vit.mark_dirty()
        vit.train(base.training)
        return PeftModel(vit, config)

    @classmethod
    def from_pretrained(cls, model: ViTForImageClassification, model_id: str, is_trainable: bool = False, **kw):
        """PeftModel.from_pretrained(base, dir) (train_loras.py:419, eval_compose.py:99,108)."""
        from safetensors.torch import load_file
        base = model._vit if isinstance(model, PeftModel) else model
        with open(os.path.join(model_id, "adapter_config.json")) as f:
            raw = json.load(f)
        fields = {k: raw[k] for k in ("task_type", "inference_mode", "r", "lora_alpha", "lora_dropout", "target_modules",
                                      "bias", "modules_to_save", "base_model_name_or_path") if k in raw}
        cfg = LoraConfig(**fields)
        pm = cls._attach(base, cfg)
        sd = load_file(os.path.join(model_id, "adapter_model.safetensors"))
        pm.load_adapter_state_dict(sd)
        pm.train(is_trainable)
        return pm

    # -- the surface the reference uses ---------------------------------------------------------
    def forward(self, pixel_values=None, **kw):
        return self._vit(pixel_values=pixel_values, **kw)

    def parameters(self, recurse: bool = True):
        return self._vit.parameters()

    def named_parameters(self, *a, **k):
        return self._vit.named_parameters(*a, **k)

    def train(self, mode: bool = True):
        super().train(mode)
        self._vit.train(mode)
        return self

    def to(self, *a, **k):
        self._vit.to(*a, **k)
        return self

    def trainable_parameter_counts(self):
        a = self._vit.arch
        D, M, P, C = a.hidden, a.mlp, a.patch_size, a.num_labels
        per_layer = 4 * (D * D + D) + (M * D + M) + (D * M + D) + 4 * D
        base = D + a.tokens * D + D * 3 * P * P + D + a.layers * per_layer + 2 * D
        head = D * C + C
        lp = sum(self.config.r * sum(module_shape(a, t)) for t in self.config.targets()) * a.layers
        if self.config.classifier_trainable():
            return lp + head, base + head + lp + head      # peft counts the saved copy of the head too
        return lp, base + head + lp

    def print_trainable_parameters(self):
        tr, tot = self.trainable_parameter_counts()
        print(f"trainable params: {tr:,} || all params: {tot:,} || trainable%: {100 * tr / tot:.4f}")

    def adapter_state_dict(self):
        eng = self._vit._engine()
        sd = {}
        for i in range(self._vit.arch.layers):
            for t in self.config.targets():
                base = "base_model.model." + _module_path(i, t)
                sd[base + ".lora_A.weight"] = eng.param(i, t, "A").detach().cpu().clone().contiguous()
                sd[base + ".lora_B.weight"] = eng.param(i, t, "B").detach().cpu().clone().contiguous()
        if self.config.classifier_trainable():
            sd["base_model.model.classifier.weight"] = eng.param(-1, "", "weight").detach().cpu().clone()
            sd["base_model.model.classifier.bias"] = eng.param(-1, "", "bias").detach().cpu().clone()
        return sd

    def load_adapter_state_dict(self, sd):
        eng = self._vit._engine()
        for k, v in sd.items():
            k2 = k.replace(".default", "")
            if k2.startswith("base_model.model.classifier"):
                which = "weight" if k2.endswith("weight") else "bias"
                eng.param(-1, "", which).copy_(v.float())
                continue
            for short, path in LINEAR_MODULES:
                for which, tag in (("A", ".lora_A.weight"), ("B", ".lora_B.weight")):
                    if k2.endswith(path + tag) and ".layer." in k2:
                        layer = int(k2.split(".layer.")[1].split(".")[0])
                        # "output.dense" is a suffix of "attention.output.dense": take the longest match
                        if short == "fc2" and k2.endswith("attention.output.dense" + tag):
                            continue
                        eng.param(layer, short, which).copy_(v.float())
        self._vit.mark_dirty()

    def save_pretrained(self, save_directory: str, **kw):
        """peft_model.save_pretrained(dir) (train_loras.py:343,354)."""
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = asdict(self.config)
        cfg["target_modules"] = list(self.config.target_modules or ["query", "value"])
        with open(os.path.join(save_directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file(self.adapter_state_dict(), os.path.join(save_directory, "adapter_model.safetensors"))

    def merge_and_unload(self) -> ViTForImageClassification:
        """W <- W + (alpha/r) B A for every adapted module, adapters removed (eval_compose.py:110).
        The merge runs on the device (vl_lora_commit of a lora_merged engine) and the merged fp32
        weights are read back into a plain model so that further adapters can be stacked."""
        import ctypes as C
        from ._lib import VL_T, check
        vit, eng = self._vit, self._vit._engine()
        sd = vit.state_dict()
        for i in range(vit.arch.layers):
            for t in self.config.targets():
                key = _module_path(i, t) + ".weight"
                W = sd[key].to(device=eng.device, dtype=torch.float32).contiguous()
                check(eng.lib.vl_merge_weight(eng.h, i, VL_T[t], C.c_void_p(W.data_ptr()), C.c_void_p(W.data_ptr()),
                                              eng._stream()), "vl_merge_weight")
                sd[key] = W.cpu()
        out = ViTForImageClassification(vit.arch, LoraSpec(), device=vit._device, precision=vit.precision)
        out.load_state_dict(sd, strict=False)
        out.train(self.training)
        return out


class _BaseModelProxy(torch.nn.Module):
    def __init__(self, vit):
        super().__init__()
        self.model = vit

    def forward(self, pixel_values=None, **kw):
        return self.model(pixel_values=pixel_values, **kw)


def get_peft_model(model: ViTForImageClassification, peft_config: LoraConfig, seed: Optional[int] = None) -> PeftModel:
    """peft.get_peft_model (train_loras.py:92): freeze the backbone, attach rank-r adapters
    (A kaiming-uniform, B = 0) to every matched Linear, keep the classifier trainable for SEQ_CLS."""
    return PeftModel._attach(model, peft_config, init_B_zero=True, seed=seed)


def setup_peft_lora(model, rank=16, alpha=16, dropout=0.1, target_modules=None, seed=None):
    """train_loras.py:79-95, same defaults.  `seed` (not in the reference) makes the kaiming-uniform A reproducible without
    touching the global RNG; None = torch's global generator, as peft does."""
    if target_modules is None:
        target_modules = ["query", "key", "value", "output.dense"]
    cfg = LoraConfig(task_type=TaskType.SEQ_CLS, inference_mode=False, r=rank, lora_alpha=alpha, lora_dropout=dropout,
                     target_modules=target_modules)
    pm = get_peft_model(model, cfg, seed=seed)
    pm.print_trainable_parameters()
    return pm
