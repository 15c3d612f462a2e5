#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ -- run ONLY in the build container.

It imports the reference's own ``whitebox_attacks.batched_fgsm_attack`` /
``get_model_output`` / ``LogitsModel`` from /root/reference (read-only) and HF
``ViTForImageClassification`` (the class ``Utils.create_vit_model`` instantiates,
Utils.py:84-90; built here from a local ``ViTConfig`` because the hub name is a
network fetch).  ``torchvision`` and ``torchattacks`` are only used inside the
reference's ``main()``; they are not installed, so empty placeholder modules are
registered so that ``import whitebox_attacks`` reaches the function definitions
(SURVEY.md section 8c).  Nothing from the reference is copied: the outputs written here are
data (inputs are regenerated from seeds; expected logits, input gradients and
FGSM images are stored).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import vit_lora_oracle as O  # noqa: E402

REF = "/root/reference"


def import_reference():
    import transformers  # noqa: F401  (before the stubs, it probes torchvision)
    from transformers import ViTForImageClassification  # noqa: F401
    for name in ("torchvision", "torchvision.transforms", "torchattacks"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = None
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchattacks"].FGSM = object
    sys.modules["torchattacks"].PGD = object
    sys.path.insert(0, REF)
    import whitebox_attacks as ref
    return ref


def hf_model(cfg: O.OracleConfig, w):
    from transformers import ViTConfig, ViTForImageClassification
    hc = ViTConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                   num_attention_heads=cfg.heads, intermediate_size=cfg.mlp,
                   image_size=cfg.image_size, patch_size=cfg.patch_size,
                   num_labels=cfg.num_labels)
    hc._attn_implementation = "eager"
    m = ViTForImageClassification(hc)
    # container transformers is 5.x: rename the 4.55.2 keys the reference's .pth uses
    ren = [("encoder.layer.", "layers."), ("attention.attention.query", "attention.q_proj"),
           ("attention.attention.key", "attention.k_proj"),
           ("attention.attention.value", "attention.v_proj"),
           ("attention.output.dense", "attention.o_proj"),
           ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2")]
    sd = {}
    for k, v in w.items():
        for a, b in ren:
            k = k.replace(a, b)
        sd[k] = v.clone()
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    m.eval()
    return m


CASES = {
    # name: (config, batch, weight seed, eps)
    "tiny17": (O.OracleConfig(image_size=64, hidden=128, layers=2, heads=2, mlp=256, num_labels=10), 4, 11, 8 / 255),
    "tiny197": (O.OracleConfig(image_size=224, hidden=128, layers=2, heads=2, mlp=256, num_labels=10), 3, 12, 8 / 255),
    "vitb": (O.OracleConfig(num_labels=10), 2, 13, 8 / 255),
}


def case_inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    y = torch.randint(0, cfg.num_labels, (B,), generator=g)
    return x, y


def main():
    ref = import_reference()
    torch.set_num_threads(8)
    mean = torch.tensor(O.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(1, 3, 1, 1)
    for name, (cfg, B, seed, eps) in CASES.items():
        # std=0.05 on the tiny nets so that softmax/GELU are exercised away from 0
        w = O.init_weights(cfg, seed=seed, std=0.02 if name == "vitb" else 0.05)
        x, y = case_inputs(cfg, B, seed)
        m = hf_model(cfg, w)
        # reference FGSM (whitebox_attacks.py:22-38), unmodified
        adv = ref.batched_fgsm_attack(m, x, y, eps, mean, std)
        m.zero_grad(set_to_none=True)
        # reference model boundary: LogitsModel / get_model_output (whitebox_attacks.py:13-19,41-48)
        with torch.no_grad():
            logits = ref.LogitsModel(m)((x - mean) / std)
        xg = x.clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(ref.get_model_output(m((xg - mean) / std)), y)
        (grad,) = torch.autograd.grad(loss, xg)
        wsum = float(sum(v.double().sum() for v in w.values()))
        out = os.path.join(HERE, f"fgsm_{name}.npz")
        np.savez_compressed(
            out, logits=logits.numpy(), loss=np.float32(loss.item()),
            grad=grad.numpy().astype(np.float32),
            # adv is x +- eps clamped: store the decision (sign) compactly + the exact tensor hash
            adv_minus_x_sign=torch.sign(adv - x).to(torch.int8).numpy(),
            adv_sum=np.float64(adv.double().sum().item()),
            adv_absmax=np.float32((adv - x).abs().max().item()),
            weight_sum=np.float64(wsum), eps=np.float32(eps),
            meta=np.array([cfg.image_size, cfg.patch_size, cfg.hidden, cfg.layers, cfg.heads,
                           cfg.mlp, cfg.num_labels, B, seed], dtype=np.int64))
        print(name, "loss", loss.item(), "->", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
