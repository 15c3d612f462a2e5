"""MI355X-native ViT + LoRA + FGSM/PGD hot path (gfx950 HIP kernels behind a C ABI).

Drop-in surface for the reference's hot path (whitebox_attacks.py / train_loras.py):

    create_vit_model, get_normalization            (Utils.py:84-93)
    batched_fgsm_attack, LogitsModel, get_model_output, FGSM, PGD
                                                   (whitebox_attacks.py:13-48,110-113)
    LoraConfig, TaskType, get_peft_model, PeftModel, setup_peft_lora
                                                   (train_loras.py:79-95,419)
    Adam                                           (train_loras.py:284)
    save_images                                    (Utils.py:106-113)

Everything numeric runs in libvitlora_hip.so (include/vitlora.h); there is no CPU path.
"""
from ._lib import LIB_PATH, NonFiniteGradient, VitLoraError, check  # noqa: F401
from .engine import (IMAGENET_MEAN, IMAGENET_STD, ArchConfig, Engine, LoraSpec,  # noqa: F401
                     canonical_key, expected_keys, resolve_targets)

__all__ = ["ArchConfig", "Engine", "LoraSpec", "VitLoraError", "NonFiniteGradient", "LIB_PATH", "IMAGENET_MEAN", "IMAGENET_STD",
           "resolve_targets", "canonical_key", "expected_keys"]


def __getattr__(name):
    # the facade modules import torch.nn etc.; load them lazily
    import importlib
    for mod in ("model", "attacks", "peft_compat", "optim", "io", "patch", "swin"):
        try:
            m = importlib.import_module(f".{mod}", __name__)
        except ModuleNotFoundError:
            continue
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
