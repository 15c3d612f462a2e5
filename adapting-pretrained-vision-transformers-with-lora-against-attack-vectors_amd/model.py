"""Model facade: the reference's model callables on top of the HIP engine.

    create_vit_model(num_classes)      Utils.py:84-90   (HF ViTForImageClassification, ViT-B/16)
    get_normalization(model_name)      Utils.py:92-93
    get_model_output / LogitsModel     whitebox_attacks.py:13-19, 41-48

`ViTForImageClassification` here is an nn.Module-shaped object: `model(x)` returns an
object with `.logits`, `model.eval()/.train()/.to()/.parameters()/.state_dict()/
.load_state_dict()` behave as the reference's scripts expect, and the call is an autograd
node: `loss.backward()` reaches the input (whitebox_attacks.py:24-31) and the LoRA /
classifier parameters (train_loras.py:310-314) through the HIP backward kernels.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Iterable, Optional

import torch

from .engine import IMAGENET_MEAN, IMAGENET_STD, ArchConfig, Engine, LoraSpec, canonical_key, expected_keys


def get_normalization(model_name=None):
    """ImageNet mean / std regardless of the argument (Utils.py:92-93)."""
    return list(IMAGENET_MEAN), list(IMAGENET_STD)


def get_model_output(outputs):
    """whitebox_attacks.py:13-19: unwrap .logits / ['logits'] / passthrough."""
    if hasattr(outputs, "logits"):
        return outputs.logits
    if isinstance(outputs, dict) and "logits" in outputs:
        return outputs["logits"]
    return outputs


class _ViTFunction(torch.autograd.Function):
    """logits = ViT(x); backward feeds dLoss/dlogits to vl_backward."""

    @staticmethod
    def forward(ctx, x, flat, model, normalise):
        eng = model._engine()
        train = bool(model.training and model.lora_spec.r > 0)
        logits = eng.forward(x, normalise=normalise, train=train)
        ctx.model, ctx.train, ctx.shape = model, train, tuple(x.shape)
        ctx.need_x = x.requires_grad
        ctx.token = model._fwd_token = object()
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if model._fwd_token is not ctx.token:
            raise RuntimeError("backward through a stale forward: the engine keeps the activations of the LAST call only")
        eng = model._engine()
        eng.set_dlogits(dlogits)
        want_p = ctx.train and ctx.needs_input_grad[1]
        gx, gp = eng.backward(ctx.need_x and ctx.needs_input_grad[0], want_p, ctx.shape)
        return gx, gp, None, None


class ViTForImageClassification(torch.nn.Module):
    """Drop-in for the object `create_vit_model` returns (Utils.py:84-90)."""

    def __init__(self, arch: Optional[ArchConfig] = None, lora: Optional[LoraSpec] = None, device=None,
                 precision: str = "f16"):
        super().__init__()
        self.precision = precision
        self.arch = arch or ArchConfig()
        self.lora_spec = lora or LoraSpec()
        self._device = torch.device(device) if device is not None else None
        self._eng: Optional[Engine] = None
        self._host_sd: Dict[str, torch.Tensor] = {}      # weights given before the engine exists
        self._flat_param: Optional[torch.nn.Parameter] = None
        self._fwd_token = None
        self.config = SimpleNamespace(num_labels=self.arch.num_labels, hidden_size=self.arch.hidden,
                                      num_hidden_layers=self.arch.layers, image_size=self.arch.image_size,
                                      patch_size=self.arch.patch_size)

    # ---- device / engine ------------------------------------------------------------------
    def _engine(self) -> Engine:
        if self._eng is None:
            dev = self._device or torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
            self._eng = Engine(self.arch, self.lora_spec, device=dev, precision=self.precision)
            # eval-mode forwards (and the backward-to-pixels after them) of 2 .. 191 images run as two half-batch chains on two
            # streams -- bit-identical, faster at the reference's batch sizes (evaluate loops, the patch EoT step); train-mode
            # steps are not split (include/vitlora.h: "api_chains")
            self._eng.set_option("api_chains", 1)
            if self._host_sd:
                self._eng.load_state_dict(self._host_sd, strict=False)
            self._flat_param = torch.nn.Parameter(self._eng.flat, requires_grad=True)
            self._seen_version = self._flat_param._version
            # read-only views of the classifier inside the flat buffer (taken once: asking the library for a pointer marks
            # the handle dirty, state_dict() must not force a re-pack of the adapters)
            self._cls_views = (self._eng.param(-1, "", "weight"), self._eng.param(-1, "", "bias"))
        return self._eng

    def sync_params(self):
        """The facade heals itself for in-place torch operations ON THE FLAT PARAMETER ITSELF (an in-place optimizer step of
        the reference's unmodified torch.optim.Adam, parameter.copy_/mul_/add_ under no_grad, a view taken from the Parameter):
        those bump the Parameter's version counter; the library is told so and re-derives the fp16 adapter operands before it
        runs.  NOT seen: writes through `parameter.data` (its own version counter) and through tensors obtained from
        `engine.param()` / `engine.flat` -- after those call `mark_dirty()` (train_loras.py does after its broadcast of
        `.data`).  (vl_adam_step and vl_param_* mark the handle themselves.)"""
        if self._eng is not None and self._flat_param is not None and self._flat_param._version != self._seen_version:
            self._eng.mark_dirty()
            self._seen_version = self._flat_param._version

    def to(self, device=None, *args, **kwargs):
        if device is not None and not isinstance(device, torch.dtype):
            d = torch.device(device)
            if d.type != "cuda":
                if self._eng is not None:
                    raise RuntimeError("the vitlora engine lives on the GPU; it cannot be moved to " + str(d))
                return self               # harmless before the engine exists (scripts call .to(device) early)
            if self._eng is not None and self._eng.device != d and d.index is not None:
                raise RuntimeError("engine already created on " + str(self._eng.device))
            self._device = d
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    # ---- parameters -----------------------------------------------------------------------
    def trainable_flat(self) -> torch.nn.Parameter:
        """The single flat fp32 Parameter holding every LoRA A/B and the classifier."""
        self._engine()
        return self._flat_param

    def parameters(self, recurse: bool = True):
        yield self.trainable_flat()

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        yield (prefix + "trainable_flat", self.trainable_flat())

    def commit(self):
        """Re-derive the fp16 GEMM operands now.  Optional: the library tracks parameter changes itself and
        commits before the next forward / attack (vl_params_changed, include/vitlora.h)."""
        self._engine().commit()

    def mark_dirty(self):
        """Parameters were written through the flat Parameter (optimizer, broadcast, in-place copy)."""
        self._engine().mark_dirty()

    # ---- state dict -----------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {canonical_key(k): v for k, v in state_dict.items()}
        exp = set(expected_keys(self.arch))
        unexpected = [k for k in sd if k not in exp]
        missing = [k for k in exp if k not in sd and k not in self._host_sd]
        if strict and (unexpected or missing):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:3]}... unexpected {unexpected[:3]}...")
        sd = {k: v.detach().float() for k, v in sd.items() if k in exp}
        self._host_sd.update({k: v.cpu() for k, v in sd.items()})
        if self._eng is not None:
            self._eng.load_state_dict(sd, strict=False)
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def state_dict(self, *args, **kwargs):
        sd = dict(self._host_sd)
        if self._eng is not None:
            sd["classifier.weight"] = self._cls_views[0].detach().cpu().clone()
            sd["classifier.bias"] = self._cls_views[1].detach().cpu().clone()
        return sd

    # ---- forward --------------------------------------------------------------------------
    def forward(self, pixel_values=None, normalise: bool = False, **kwargs):
        x = pixel_values
        if x is None:
            raise ValueError("pixel_values is required")
        eng = self._engine()
        self.sync_params()
        x = x.to(device=eng.device, dtype=torch.float32)
        logits = _ViTFunction.apply(x, self._flat_param, self, bool(normalise))
        return SimpleNamespace(logits=logits)


class LogitsModel(torch.nn.Module):
    """whitebox_attacks.py:41-48."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, x):
        return get_model_output(self.model(x))


def create_vit_model(num_classes: int, pretrained: bool = True, arch: Optional[ArchConfig] = None, device=None,
                     precision: str = "f16"):
    """ViT-B/16 classifier with `num_classes` labels (Utils.py:84-90).  The reference fetches
    'google/vit-base-patch16-224' from the hub; here the architecture is built locally and the
    weights come from `load_state_dict` (the reference's very next step, whitebox_attacks.py:94)."""
    a = arch or ArchConfig()
    a = ArchConfig(image_size=a.image_size, patch_size=a.patch_size, hidden=a.hidden, layers=a.layers,
                   heads=a.heads, mlp=a.mlp, num_labels=int(num_classes), ln_eps=a.ln_eps)
    return ViTForImageClassification(a, device=device, precision=precision)
