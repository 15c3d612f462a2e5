"""GPU tests of the drop-in surface: the reference's model / attack / peft callables on the HIP
engine, checked against the oracle and against the golden vectors made by the reference's own
`batched_fgsm_attack` + HF ViT (tests/golden/)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import O, PKG, make_case, pkg, rel_l2
from test_oracle_golden import load_case

pytestmark = pytest.mark.gpu


PRECS = ["f16", "f32"]
TOL_ACT = {"f16": 4e-3, "f32": 1e-4}       # north_star: 1e-2 (16-bit) / 1e-3 (fp32)
TOL_GRAD = {"f16": 6e-3, "f32": 1e-4}


def _model(cfg, w, num_labels=None, precision="f16"):
    P = pkg()
    arch = P.ArchConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, hidden=cfg.hidden, layers=cfg.layers,
                        heads=cfg.heads, mlp=cfg.mlp, num_labels=cfg.num_labels, ln_eps=cfg.ln_eps)
    m = P.create_vit_model(cfg.num_labels, arch=arch, precision=precision)
    m.load_state_dict(w)
    return m.eval()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["tiny17", "tiny197", "vitb"])
def test_golden_reference_fgsm_and_logits(name, prec):
    """HIP path vs what the REFERENCE produced (HF ViT logits, loss, input gradient, FGSM decisions: G1-G3)."""
    P = pkg()
    cfg, w, x, y, z = load_case(name)
    model = _model(cfg, w, precision=prec)
    mean, std = P.get_normalization("google_vit")
    mt, st = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)
    logits = P.LogitsModel(model)(((x - mt) / st).cuda()).cpu()
    assert rel_l2(logits, torch.from_numpy(z["logits"])) < TOL_ACT[prec]
    # the reference's own input gradient (perturbed.grad of whitebox_attacks.py:30-32)
    eng = model._engine()
    eng.forward(x.cuda(), normalise=True)
    loss = eng.loss_ce(y.cuda()).item()
    gx, _ = eng.backward(True, False, tuple(x.shape))
    g_ref = torch.from_numpy(z["grad"])
    assert abs(loss - float(z["loss"])) < TOL_ACT[prec] * float(z["loss"])
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD[prec], rel_l2(gx.cpu(), g_ref)
    adv = P.batched_fgsm_attack(model, x.cuda(), y.cuda(), float(z["eps"]), mt.cuda(), st.cuda()).cpu()
    assert abs((adv - x).abs().max().item() - float(z["adv_absmax"])) < 1e-6
    ref_sign = torch.from_numpy(z["adv_minus_x_sign"])
    sign = torch.sign(adv - x).to(torch.int8)
    # decisions may differ only where the reference gradient is inside the rounding noise
    big = (g_ref.abs() > 0.1 * g_ref.abs().mean()) & (ref_sign != 0)
    agree_big = (sign[big] == ref_sign[big]).float().mean().item()
    agree_all = (sign == ref_sign).float().mean().item()
    assert agree_big > 0.9999 and agree_all > (0.998 if prec == "f16" else 0.9999), (agree_big, agree_all)


def test_reference_style_autograd_fgsm():
    """The reference's FGSM body (whitebox_attacks.py:24-36) written with torch ops against OUR model
    object: requires_grad_ -> normalise -> model -> F.cross_entropy -> backward -> .grad."""
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=3, r=0)
    model = _model(cfg, w)
    mean, std = P.get_normalization(None)
    mt, st = torch.tensor(mean).view(1, 3, 1, 1).cuda(), torch.tensor(std).view(1, 3, 1, 1).cuda()
    perturbed = x.cuda().clone().detach().requires_grad_(True)
    outputs = model((perturbed - mt) / st)
    loss = F.cross_entropy(P.get_model_output(outputs), y.cuda())
    loss.backward()
    l_ref, g_ref, _ = O.loss_and_input_grad(w, cfg, x, y)
    assert abs(loss.item() - l_ref.item()) < TOL_ACT["f16"] * abs(l_ref.item())
    assert rel_l2(perturbed.grad.cpu(), g_ref) < TOL_GRAD["f16"]


def test_pgd_class_canonical_and_torchattacks_compat():
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=4, r=0)
    model = _model(cfg, w)
    eps, alpha, steps = 8 / 255, 2 / 255, 4
    mean, std = P.get_normalization(None)
    # compat: the reference's call pattern (set_normalization_used on [0,1] images)
    atk = P.PGD(P.LogitsModel(model), eps=eps, alpha=alpha, steps=steps, random_start=False)
    atk.set_normalization_used(mean=mean, std=std)
    adv = atk(x.cuda(), y.cuda()).cpu()
    ref = O.pgd_torchattacks_compat(w, cfg, x, y, eps, alpha, steps)
    st = torch.tensor(std).view(1, 3, 1, 1)
    d, dr = (adv - x) * st, (ref - x) * st            # perturbation in the attack's own space
    assert d.abs().max().item() <= eps + 1e-5
    agree = (torch.sign(d) == torch.sign(dr)).float().mean().item()
    assert agree > 0.93, agree      # the attack runs in x*std+mean space on a model fed raw pixels: flatter gradients, more near-zero entries
    # no normalisation registered: model consumes the adversarial image directly
    atk2 = P.PGD(P.LogitsModel(model), eps=eps, alpha=alpha, steps=steps, random_start=False)
    adv2 = atk2(x.cuda(), y.cuda()).cpu()
    adv_o = x.clone()
    for _ in range(steps):
        _, g, _ = O.loss_and_input_grad(w, cfg, adv_o, y, normalised=True)
        adv_o = O.pgd_step(adv_o, x, g, eps, alpha)
    assert (torch.sign(adv2 - x) == torch.sign(adv_o - x)).float().mean().item() > 0.97
    # FGSM class == batched_fgsm_attack when the same normalisation is registered
    f = P.FGSM(P.LogitsModel(model), eps=eps)
    a1 = f(O.normalise(x).cuda(), y.cuda()).cpu()      # no set_normalization_used: model fed as is
    a_ref = torch.clamp(O.normalise(x) + eps * torch.sign(O.loss_and_input_grad(w, cfg, O.normalise(x), y, normalised=True)[1]), 0, 1)
    assert (torch.sign(a1 - O.normalise(x)) == torch.sign(a_ref - O.normalise(x))).float().mean().item() > 0.99


def test_peft_roundtrip_and_merge(tmp_path):
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=3)
    base = _model(cfg, w)
    pm = P.setup_peft_lora(base, rank=lora.r, alpha=lora.alpha, dropout=0.0)
    eng = pm._vit._engine()
    # peft init: B = 0 -> the adapted model equals the base model
    xn = O.normalise(x).cuda()
    pm.eval()
    l0 = pm.base_model(pixel_values=xn).logits.detach().cpu()
    assert rel_l2(l0, base(xn).logits.detach().cpu()) < 2e-3
    for (i, t), (A, B) in lora.ab.items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(B)
    l1 = pm.base_model(pixel_values=xn).logits.detach().cpu()
    assert rel_l2(l1, O.vit_forward(w, cfg, O.normalise(x), lora)) < TOL_ACT["f16"]
    d = str(tmp_path / "adapter")
    pm.save_pretrained(d)
    assert sorted(os.listdir(d)) == ["adapter_config.json", "adapter_model.safetensors"]
    from safetensors.torch import load_file
    sd = load_file(os.path.join(d, "adapter_model.safetensors"))
    assert "base_model.model.vit.encoder.layer.0.attention.attention.query.lora_A.weight" in sd
    assert "base_model.model.vit.encoder.layer.1.output.dense.lora_B.weight" in sd
    assert "base_model.model.classifier.weight" in sd
    assert len(sd) == cfg.layers * 5 * 2 + 2
    pm2 = P.PeftModel.from_pretrained(_model(cfg, w), d)
    l2 = pm2.base_model(pixel_values=xn).logits.detach().cpu()
    assert torch.allclose(l1, l2, rtol=0, atol=1e-5)
    merged = pm2.merge_and_unload()
    l3 = merged(xn).logits.detach().cpu()
    assert rel_l2(l3, l1) < TOL_ACT["f16"]


def test_lora_training_steps_match_oracle():
    """train_loras.py:303-315 with the same objects: base_model(pixel_values=...).logits,
    CrossEntropyLoss, backward, Adam.step -- three steps, compared with torch autograd + oracle Adam."""
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=4)
    base = _model(cfg, w)
    pm = P.setup_peft_lora(base, rank=lora.r, alpha=lora.alpha, dropout=0.0)
    eng = pm._vit._engine()
    for (i, t), (A, B) in lora.ab.items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(B)
    pm.train()
    crit = torch.nn.CrossEntropyLoss()
    lr = 1e-3
    opt = P.Adam(pm.parameters(), lr=lr, model=pm)
    xn = O.normalise(x)
    # oracle state
    ol = O.OracleLora(r=lora.r, alpha=lora.alpha, targets=lora.targets, ab={k: (a.clone(), b.clone()) for k, (a, b) in lora.ab.items()})
    ow = dict(w)
    mom = {}
    losses_hip, losses_ref = [], []
    for step in range(1, 4):
        opt.zero_grad()
        logits = pm.base_model(pixel_values=xn.cuda()).logits
        loss = crit(logits, y.cuda())
        loss.backward()
        opt.step()
        losses_hip.append(loss.item())
        l_ref, _, grads = O.lora_train_grads(ow, cfg, xn, y, ol)
        losses_ref.append(l_ref.item())
        for key, g in grads.items():
            if key[0] == "cls":
                name = "classifier." + key[1]
                p = ow[name]
            else:
                A, B = ol.ab[key[1:]]
                p = A if key[0] == "A" else B
            m_, v_ = mom.get(key, (torch.zeros_like(p), torch.zeros_like(p)))
            p2, m_, v_ = O.adam_step(p, g, m_, v_, step, lr=lr)
            mom[key] = (m_, v_)
            if key[0] == "cls":
                ow[name] = p2
            else:
                A, B = ol.ab[key[1:]]
                ol.ab[key[1:]] = (p2, B) if key[0] == "A" else (A, p2)
    assert all(abs(a - b) < 5e-3 * abs(b) for a, b in zip(losses_hip, losses_ref)), (losses_hip, losses_ref)
    assert losses_hip[-1] < losses_hip[0]
    # parameters after 3 Adam steps (Adam normalises the step: compare the direction of the update)
    for (i, t) in list(lora.ab.keys())[:6]:
        for which, idx in (("A", 0), ("B", 1)):
            got = eng.param(i, t, which).cpu() - lora.ab[(i, t)][idx]
            want = ol.ab[(i, t)][idx] - lora.ab[(i, t)][idx]
            cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-30))
            assert cos > 0.97, (i, t, which, cos)


def test_flagged_fp16_step_leaves_the_parameters_untouched_and_is_consumed_in_its_own_step():
    """Round-3 ADVICE: flag code 2 (a backward kernel clamped a gradient to +-65504) leaves FINITE but wrong LoRA gradients;
    the train loop therefore reads the flag with engine.check() after loss.backward() -- in the step it belongs to -- and
    drops the step the way train_loras.py does (NaN gradient -> the fused Adam skips every element; with ranks the NaN
    rides the one all-reduce).  Parameters and moments must be unchanged, the next evaluate() must not inherit a flag, and
    a clean step afterwards must train."""
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=4, layers=4)
    gain = 512.0
    w2 = {k: (v * gain if (k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight")) else v) for k, v in w.items()}
    pm = P.setup_peft_lora(_model(cfg, w2), rank=lora.r, alpha=lora.alpha, dropout=0.0)
    eng = pm._vit._engine()
    for (i, t), (A, B) in lora.ab.items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(B)
    pm.train()
    crit = torch.nn.CrossEntropyLoss()
    opt = P.Adam(pm.parameters(), lr=1e-2, model=pm)
    xn = O.normalise(x).cuda()
    before = pm._vit.trainable_flat().detach().clone()
    opt.zero_grad()
    loss = crit(pm.base_model(pixel_values=xn).logits, y.cuda())
    loss.backward()
    with pytest.raises(P.NonFiniteGradient):
        eng.check()                                   # this step's flag, read in this step
    flat = opt.params[0]
    flat.grad = torch.full_like(flat.data, float("nan"))
    opt.step()
    with pytest.raises(P.NonFiniteGradient):
        eng.check()                                   # Adam's own report (every element skipped)
    assert torch.equal(pm._vit.trainable_flat().detach(), before)
    assert float(opt.m1.abs().max()) == 0.0 and float(opt.m2.abs().max()) == 0.0
    # nothing is left for the next forward (evaluate() after the last step of an epoch) to trip over
    pm.eval()
    with torch.no_grad():
        pm.base_model(pixel_values=xn).logits
    eng.check()
    # the handle still trains: unit-gain weights through the same objects
    pm2 = P.setup_peft_lora(_model(cfg, w), rank=lora.r, alpha=lora.alpha, dropout=0.0)
    pm2.train()
    opt2 = P.Adam(pm2.parameters(), lr=1e-2, model=pm2)
    b2 = pm2._vit.trainable_flat().detach().clone()
    crit(pm2.base_model(pixel_values=xn).logits, y.cuda()).backward()
    pm2._vit._engine().check()
    opt2.step()
    assert opt2.params[0].grad is None                # the gradient buffer is consumed by step() (rewritten by the next backward)
    assert not torch.equal(pm2._vit.trainable_flat().detach(), b2)


def test_unmodified_torch_optimizer_is_seen_by_the_next_forward():
    """train_loras.py:284 builds torch.optim.Adam(peft_model.parameters()) -- torch writes the flat Parameter in place, the
    library never sees that call.  The facade compares the Parameter's version counter before every forward / attack and
    tells the library, so the fp16 adapter operands are never stale (round-2 ADVICE); same for copy_ and a kept view."""
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=4)
    base = _model(cfg, w)
    pm = P.setup_peft_lora(base, rank=lora.r, alpha=lora.alpha, dropout=0.0)
    eng = pm._vit._engine()
    for (i, t), (A, B) in lora.ab.items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(B)
    pm.train()
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(pm.parameters(), lr=1e-2)              # the reference's optimizer, unmodified
    xn = O.normalise(x).cuda()
    logits0 = pm.base_model(pixel_values=xn).logits
    loss = crit(logits0, y.cuda())
    loss.backward()
    commits = eng.counter("commits")
    opt.step()                                                     # in-place torch write: bumps the version, nothing else
    logits1 = pm.base_model(pixel_values=xn).logits
    assert eng.counter("commits") == commits + 1                   # the forward re-derived the operands
    assert float((logits1 - logits0).abs().max()) > 1e-4           # ... and they are the new adapters
    # reference value: the same parameters through a freshly committed engine
    flat = pm._vit.trainable_flat().detach().clone()
    eng.commit()
    logits2 = pm.base_model(pixel_values=xn).logits
    assert torch.equal(logits1.detach(), logits2.detach())
    # a write through a kept view + an attack: FGSM must see it too
    pm.eval()
    adv0 = P.batched_fgsm_attack(pm, x.cuda(), y.cuda(), 8 / 255, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]).clone()
    with torch.no_grad():
        pm._vit.trainable_flat().mul_(0.5)
    adv1 = P.batched_fgsm_attack(pm, x.cuda(), y.cuda(), 8 / 255, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert float((adv1 != adv0).float().mean()) > 0.01
    assert torch.isfinite(flat).all()


def test_sequential_adapter_merge_matches_summed_update(tmp_path):
    """eval_compose.py:102-114: adapters merged one after the other == W + s*B1*A1 + s*B2*A2 (oracle arithmetic)."""
    P = pkg()
    cfg, w, lora1, x, y = make_case(batch=3, seed=5)
    lora2 = O.init_lora(cfg, r=lora1.r, targets=lora1.targets, seed=77, b_std=0.05)
    dirs = []
    for k, lora in enumerate((lora1, lora2)):
        pm = P.setup_peft_lora(_model(cfg, w), rank=lora.r, alpha=lora.alpha, dropout=0.0)
        eng = pm._vit._engine()
        for (i, t), (A, B) in lora.ab.items():
            eng.param(i, t, "A").copy_(A)
            eng.param(i, t, "B").copy_(B)
        d = str(tmp_path / f"adapter{k}")
        pm.save_pretrained(d)
        dirs.append(d)
    import eval_compose
    merged = eval_compose.merge_lora_adapters(_model(cfg, w), dirs)
    xn = O.normalise(x)
    got = merged(xn.cuda()).logits.detach().cpu()
    w2 = {k: v.clone() for k, v in w.items()}
    for lora in (lora1, lora2):
        for (i, t), (A, B) in lora.ab.items():
            key = f"vit.encoder.layer.{i}." + dict(O.LINEAR_MODULES)[t] + ".weight"
            w2[key] = w2[key] + lora.scaling * (B @ A)
    ref = O.vit_forward(w2, cfg, xn, None)
    assert rel_l2(got, ref) < TOL_ACT["f16"]
    # and the merge changed the model (both adapters matter)
    assert rel_l2(ref, O.vit_forward(w, cfg, xn, None)) > 5e-2
    # accuracy / weighted-F1 helper against hand-computed values
    acc, f1 = eval_compose.accuracy_and_weighted_f1([0, 0, 1, 1, 2], [0, 1, 1, 1, 0], 3)
    assert abs(acc - 0.6) < 1e-6 and abs(f1 - (0.5 * 2 / 5 + 0.8 * 2 / 5 + 0.0)) < 1e-6
