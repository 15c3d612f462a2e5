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

Vectors written:
  G1-G3  fgsm_<case>.npz      logits / loss / input gradient / FGSM decisions (imported batched_fgsm_attack + HF ViT)
  G4     pgd_<case>.npz       PGD-{1,3,7} trajectories: every iteration's ascent step IS the imported reference
                              batched_fgsm_attack(model, adv, y, alpha, mean, std) on the HF model; only the eps-ball
                              projection around x0 (two clamps) is written here.  random_start = a seeded noise tensor.
  G5     lora_<case>.npz      logits and dLoss/d(A, B, classifier) of plain-torch low-rank branches
                              y = W x + b + (alpha/r) B (A x) wrapped around the HF model's own nn.Linear modules
                              (peft is not installable: this pins the oracle's LoRA arithmetic to HF's linears + autograd).
  G8     save_images.npz      bytes written by the imported Utils.save_images (PNG round trip through PIL).
  fmnist_t10k_labels_512.bin  idx header + first 512 labels of the reference's FashionMNIST t10k label file
                              (BASELINE config 1: real labels, synthesised pixels).

Usage:  python tests/golden/make_golden.py [g123] [g4] [g5] [g8]     (default: all)
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


class LoraLinear(torch.nn.Module):
    """y = base(x) + (alpha / r) * B (A x): the formula of peft's lora.Linear.forward with dropout 0
    (train_loras.py:79-95 asks peft for exactly this), written with plain torch around HF's own nn.Linear."""

    def __init__(self, base, A, B, scaling):
        super().__init__()
        self.base = base
        self.A = torch.nn.Parameter(A.clone())
        self.B = torch.nn.Parameter(B.clone())
        self.scaling = scaling

    def forward(self, x):
        return self.base(x) + self.scaling * torch.nn.functional.linear(torch.nn.functional.linear(x, self.A), self.B)


HF_ATTR = {"q": ("attention", "q_proj"), "k": ("attention", "k_proj"), "v": ("attention", "v_proj"),
           "o": ("attention", "o_proj"), "fc1": ("mlp", "fc1"), "fc2": ("mlp", "fc2")}


def wrap_lora(m, lora):
    wrapped = {}
    for (i, t), (A, B) in lora.ab.items():
        parent = getattr(m.vit.layers[i], HF_ATTR[t][0])
        w = LoraLinear(getattr(parent, HF_ATTR[t][1]), A, B, lora.scaling)
        setattr(parent, HF_ATTR[t][1], w)
        wrapped[(i, t)] = w
    return wrapped


PGD_CASES = {"tiny17": (1, 3, 7, 20), "vitb": (1, 3, 7, 20)}      # 20 = the headline attack length (BASELINE config 2)
LORA_CASES = {"tiny17": (4, 8, 16), "tiny197": (8,), "vitb": (4, 8, 16)}
LORA_TARGETS = ("q", "k", "v", "o", "fc2")          # ["query","key","value","output.dense"], train_loras.py:81


def make_pgd(ref, mean, std):
    eps, alpha = 8 / 255, 2 / 255
    for name, steps_list in PGD_CASES.items():
        cfg, B, seed, _ = CASES[name]
        w = O.init_weights(cfg, seed=seed, std=0.02 if name == "vitb" else 0.05)
        x, y = case_inputs(cfg, B, seed)
        m = hf_model(cfg, w)
        noise = torch.rand(x.shape, generator=torch.Generator().manual_seed(seed + 2000)) * 2 - 1
        out = {}
        for start in ("x0", "noise"):
            adv = x.clone() if start == "x0" else torch.clamp(x + eps * noise, 0, 1)
            for k in range(1, max(steps_list) + 1):
                # ascent step = the reference's own function on the HF model (alpha in the place of epsilon)
                stepped = ref.batched_fgsm_attack(m, adv, y, alpha, mean, std)
                m.zero_grad(set_to_none=True)
                adv = torch.clamp(x + torch.clamp(stepped - x, -eps, eps), 0, 1).detach()
                if k in steps_list:
                    # delta is a multiple of alpha unless clipped: int8 code + exact float where it is not
                    out[f"delta_{start}_{k}"] = (adv - x).numpy().astype(np.float32)
        path = os.path.join(HERE, f"pgd_{name}.npz")
        np.savez_compressed(path, eps=np.float32(eps), alpha=np.float32(alpha), noise_seed=np.int64(seed + 2000),
                            steps=np.array(steps_list, dtype=np.int64), **out)
        print("G4", name, "->", path, os.path.getsize(path) // 1024, "KiB")


def make_lora(ref, mean, std):
    for name, ranks in LORA_CASES.items():
        cfg, B, seed, _ = CASES[name]
        w = O.init_weights(cfg, seed=seed, std=0.02 if name == "vitb" else 0.05)
        x, y = case_inputs(cfg, B, seed)
        out = {}
        for r in ranks:
            lora = O.init_lora(cfg, r=r, targets=LORA_TARGETS, seed=seed + 100 + r, b_std=0.02 if name == "vitb" else 0.05)
            m = hf_model(cfg, w)
            for p_ in m.parameters():
                p_.requires_grad_(False)
            m.classifier.weight.requires_grad_(True)
            m.classifier.bias.requires_grad_(True)
            wrapped = wrap_lora(m, lora)
            logits = ref.get_model_output(m((x - mean) / std))
            loss = torch.nn.functional.cross_entropy(logits, y)
            loss.backward()
            out[f"r{r}_logits"] = logits.detach().numpy()
            out[f"r{r}_loss"] = np.float32(loss.item())
            out[f"r{r}_dcls_w"] = m.classifier.weight.grad.numpy()
            out[f"r{r}_dcls_b"] = m.classifier.bias.grad.numpy()
            keep_layers = sorted({0, cfg.layers // 2, cfg.layers - 1})
            norms = []
            for (i, t), mod in sorted(wrapped.items()):
                norms.append([float(mod.A.grad.double().norm()), float(mod.B.grad.double().norm())])
                if i in keep_layers and (name != "vitb" or t in ("q", "o", "fc2")):
                    out[f"r{r}_dA_{i}_{t}"] = mod.A.grad.numpy()
                    out[f"r{r}_dB_{i}_{t}"] = mod.B.grad.numpy()
            out[f"r{r}_grad_norms"] = np.array(norms, dtype=np.float64)     # every (layer, target), sorted order
        path = os.path.join(HERE, f"lora_{name}.npz")
        np.savez_compressed(path, ranks=np.array(ranks, dtype=np.int64), **out)
        print("G5", name, "->", path, os.path.getsize(path) // 1024, "KiB")


def make_save_images():
    import tempfile
    from PIL import Image
    sys.path.insert(0, REF)
    import Utils
    g = torch.Generator().manual_seed(77)
    img = torch.rand(3, 3, 20, 24, generator=g) * 1.3 - 0.15            # values below 0 and above 1 too
    img[0, :, 0, :8] = torch.tensor([0.0, 1.0, 0.5, 0.999, 254.5 / 255, 0.0039215, 0.00392157, 1.0 / 255])
    names = [f"g8_{i}.png" for i in range(3)]
    with tempfile.TemporaryDirectory() as d:
        Utils.save_images(img, names, d)                                # Utils.py:106-113, unmodified
        got = np.stack([np.asarray(Image.open(os.path.join(d, n))) for n in names])
    path = os.path.join(HERE, "save_images.npz")
    np.savez_compressed(path, images=img.numpy(), bytes_hwc=got)
    print("G8 ->", path, got.shape, got.dtype)


def make_fmnist_labels():
    src = os.path.join(REF, "fashion_data", "FashionMNIST", "raw", "t10k-labels-idx1-ubyte")
    raw = open(src, "rb").read()[:8 + 512]
    with open(os.path.join(HERE, "fmnist_t10k_labels_512.bin"), "wb") as f:
        f.write(raw)
    print("fmnist labels:", list(raw[8:24]), "...")


def main():
    ref = import_reference()
    torch.set_num_threads(8)
    what = set(sys.argv[1:]) or {"g123", "g4", "g5", "g8"}
    mean = torch.tensor(O.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(1, 3, 1, 1)
    if "g4" in what:
        make_pgd(ref, mean, std)
    if "g5" in what:
        make_lora(ref, mean, std)
    if "g8" in what:
        make_save_images()
        make_fmnist_labels()
    if "g123" not in what:
        return
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
