"""The bf16 operand mode (VL_PREC_BF16, round 4): the SAME kernels as the fp16 path instantiated on __bf16
(csrc/common.h; the library compiles every 16-bit source twice), behind the same C ABI (precision = "bf16").

north_star names "1e-2 bf16"; BASELINE config 5 names bf16.  What bf16 buys is RANGE -- fp32's exponent, so no gradient
ever leaves it: no VL_ERR_NONFINITE, no fp32 redo of a batch, no skipped optimizer step, no per-image scale needed (it is
kept: harmless).  What it costs is mantissa: 8 bits against fp16's 11.  Measured on the MI355X (tools/parity_report.py,
gpurun_out/r4_parity.log) against the reference's own outputs / the fp32 oracle:

                       logits      dL/dx       LoRA dA / dB
    tiny (2 layers)    5e-3..7e-3  8.0e-3      1.2e-2
    ViT-B/16 (golden)  7.9e-3      1.54e-2     2.5e-2 (12 layers, r = 8)
    fp16 mode, ViT-B   1.1e-3      1.8e-3      3.6e-3

so bf16 meets north_star's 1e-2 on logits and on 2-layer gradients and does NOT meet it on ViT-B gradients: a finding, not a
tuning gap -- rounding the frozen weights to bf16 alone costs 6.6e-3 of the ViT-B input gradient, all 16-bit storage sites
1.1e-2 with fp32 residual streams and 1.4e-2 with the 16-bit streams of this round (tools/error_budget_streams.py vitb bf16,
the CPU oracle with bf16 round trips -- no kernel involved).  The tolerances below are those measurements with ~30 % head room;
the kernel LOGIC is pinned much tighter against the oracle that rounds to bf16 where the kernels do (TOL_SIM)."""
import numpy as np
import pytest
import torch

from helpers import O, make_case, make_engine, pkg, rel_l2
from test_oracle_golden import GOLD, load_case

pytestmark = pytest.mark.gpu

TOL_LOGITS = 1.5e-2          # measured 5e-3 .. 1.2e-2
TOL_GRAD = 2.2e-2            # measured 8e-3 (2 layers) .. 1.54e-2 (ViT-B)
TOL_LORA = {2: 2e-2, 12: 3.5e-2}      # by depth: measured 1.2e-2 / 2.5e-2
TOL_SIM = 8e-3               # against the bf16-simulating oracle: accumulation order and which way a tie rounds


@pytest.fixture
def bf16_sim():
    keep = O.SIM_DTYPE
    O.SIM_DTYPE = torch.bfloat16
    yield
    O.SIM_DTYPE = keep


@pytest.mark.parametrize("image_size,batch", [(64, 4), (224, 3)])
@pytest.mark.parametrize("with_lora", [False, True])
def test_bf16_forward_loss_and_input_gradient(image_size, batch, with_lora, bf16_sim):
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8 if with_lora else 0)
    eng = make_engine(cfg, w, lora, precision="bf16")
    assert eng.precision == "bf16"
    logits = eng.forward(x.cuda(), normalise=True)
    loss = eng.loss_ce(y.cuda())
    gx, _ = eng.backward(True, False, tuple(x.shape))
    eng.check()
    l_ref, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
    assert rel_l2(logits.cpu(), lg_ref) < TOL_LOGITS
    assert abs(loss.item() - l_ref.item()) < TOL_LOGITS * max(1.0, abs(l_ref.item()))
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD, rel_l2(gx.cpu(), g_ref)
    big = g_ref.abs() > 0.05 * g_ref.abs().mean()
    assert (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item() > 0.998
    # kernel logic: the oracle with bf16 round trips at the kernels' storage sites
    tr = {}
    lg_sim = O.vit_forward(w, cfg, O.normalise(x), lora, sim16=True, trace=tr)
    assert rel_l2(logits.cpu(), lg_sim) < TOL_SIM, rel_l2(logits.cpu(), lg_sim)
    # saved activations are bf16 tensors of the right values
    eng.set_dead_rows(False)
    eng.forward(x.cuda(), normalise=True)
    for name, i in (("xs", 0), ("xs", 2 * cfg.layers), ("qkv", 0), ("ctx", cfg.layers - 1)):
        got = eng.debug_tensor(name, i)
        assert got.dtype == torch.bfloat16
        key = f"{name}{i}"
        ref = tr[key]
        assert rel_l2(got.float().cpu().view(ref.shape), ref) < 1.2e-2, key


def test_bf16_matches_the_reference_outputs_on_vit_b():
    """G1-G3 at ViT-B/16: logits, loss and dCE/dx computed by HF ViT + the reference's batched_fgsm_attack (tests/golden)."""
    cfg, w, x, y, z = load_case("vitb")
    eng = make_engine(cfg, w, None, precision="bf16")
    logits = eng.forward(x.cuda(), normalise=True).cpu()
    loss = eng.loss_ce(y.cuda()).item()
    gx, _ = eng.backward(True, False, tuple(x.shape))
    eng.check()
    g_ref = torch.from_numpy(z["grad"])
    e_l, e_g = rel_l2(logits, torch.from_numpy(z["logits"])), rel_l2(gx.cpu(), g_ref)
    print(f"bf16 vs the reference on ViT-B: logits {e_l:.2e}, input gradient {e_g:.2e} (north_star 1e-2: met / NOT met)")
    assert e_l < TOL_LOGITS and abs(loss - float(z["loss"])) < TOL_LOGITS * float(z["loss"])
    assert e_g < TOL_GRAD, e_g
    big = g_ref.abs() > 0.1 * g_ref.abs().mean()
    assert (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item() > 0.9995
    assert (torch.sign(gx.cpu()) == torch.sign(g_ref)).float().mean().item() > 0.99          # FGSM decisions


@pytest.mark.parametrize("name,r,depth", [("tiny17", 4, 2), ("tiny197", 8, 2), ("vitb", 8, 12)])
def test_bf16_lora_gradients_against_the_golden_vectors(name, r, depth):
    """G5: d(A), d(B), d(classifier) against plain-torch low-rank branches around the HF model's own linears."""
    import os
    from test_hip_configs import TARGETS, flat_slices
    cfg, w, x, y, _ = load_case(name)
    z = np.load(os.path.join(GOLD, f"lora_{name}.npz"))
    seed = int(np.load(os.path.join(GOLD, f"fgsm_{name}.npz"))["meta"][8])
    lora = O.init_lora(cfg, r=r, targets=TARGETS, seed=seed + 100 + r, b_std=0.02 if name == "vitb" else 0.05)
    eng = make_engine(cfg, w, lora, precision="bf16")
    sl = flat_slices(eng, cfg, lora)
    logits = eng.forward(x.cuda(), normalise=True, train=True).cpu()
    eng.loss_ce(y.cuda())
    _, gp = eng.backward(False, True)
    eng.check()
    assert rel_l2(logits, torch.from_numpy(z[f"r{r}_logits"])) < TOL_LOGITS
    worst = 0.0
    for key in z.files:
        if key.startswith(f"r{r}_dA_") or key.startswith(f"r{r}_dB_"):
            _, which, i, t = key.split("_", 3)
            off, n, shape = sl[(which[1], int(i), t)]
            worst = max(worst, rel_l2(gp[off:off + n].view(shape).cpu(), torch.from_numpy(z[key])))
    print(f"bf16 LoRA gradients, {name} r={r}: worst relative error {worst:.2e}")
    assert 0.0 < worst < TOL_LORA[depth], worst


def test_bf16_has_no_gradient_range_cliff():
    """LayerNorm gains of 512 over 4 layers: the fp16 path MUST flag this (tests/test_hip_engine.py) and the CLIs redo the batch
    in fp32; bf16 carries fp32's exponent -- no flag, a finite non-zero gradient, a finite attack.  (No accuracy claim at this
    gain: the network is chaotic there -- the bf16 and fp32 gradients are uncorrelated, and two fp32 programs agree to a few
    per cent only; the test is about RANGE.)"""
    P = pkg()
    cfg, w, lora, x, y = make_case(image_size=64, batch=4, r=8, layers=4)
    gain = 512.0
    w2 = {k: (v * gain if (k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight") or k == "classifier.weight")
              else v) for k, v in w.items()}
    e16 = make_engine(cfg, w2, lora, precision="f16")
    e16.forward(x.cuda(), normalise=True)
    e16.loss_ce(y.cuda())
    e16.backward(True, False, tuple(x.shape))
    with pytest.raises(P.NonFiniteGradient):
        e16.check()
    eb = make_engine(cfg, w2, lora, precision="bf16")
    eb.forward(x.cuda(), normalise=True)
    eb.loss_ce(y.cuda())
    gb, _ = eb.backward(True, False, tuple(x.shape))
    eb.check()                                           # nothing flagged
    adv = eb.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=1)
    eb.check()
    assert torch.isfinite(gb).all() and torch.isfinite(adv).all() and float(gb.abs().max()) > 0


@pytest.mark.parametrize("name", ["tiny17", "vitb"])
def test_bf16_pgd_against_the_reference_driven_trajectories(name):
    """G4 in bf16: pixels identical to the trajectory whose every ascent step is the reference's batched_fgsm_attack.  A
    near-zero gradient entry flips its sign more often at 8 mantissa bits, and every later step sees it (measured on ViT-B:
    99.5 / 98.5 / 96.7 / 94.3 % after 1 / 3 / 7 / 20 steps; fp16 mode 99.9 / 99.8 / 99.5 / 98.5 %)."""
    import os
    cfg, w, x, y, _ = load_case(name)
    z = np.load(os.path.join(GOLD, f"pgd_{name}.npz"))
    eng = make_engine(cfg, w, None, precision="bf16")
    eps, alpha = float(z["eps"]), float(z["alpha"])
    for k in [int(v) for v in z["steps"]]:
        adv = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, k, random_start=False).cpu()
        same = ((adv - (x + torch.from_numpy(z[f"delta_x0_{k}"]))).abs() < 1e-6).float().mean().item()
        print(f"G4 {name} bf16 k={k}: {same:.5f} of the pixels identical")
        assert same > {1: 0.99, 3: 0.975, 7: 0.95, 20: 0.92}[k], (k, same)
        assert (adv - x).abs().max().item() <= eps + 1e-6
    eng.check()
    # seeded determinism, one captured graph, shard slice bit-equal (images are independent in this mode too)
    a = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 5, random_start=True, seed=9).clone()
    assert torch.equal(a, eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 5, random_start=True, seed=9))
    full = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 5, random_start=False).clone()
    h = x.shape[0] // 2          # half the batch: the 1/B of the mean loss changes by a power of two, which the per-image scale absorbs
    part = eng.pgd_attack(x[h:2 * h].cuda().contiguous(), y[h:2 * h].cuda().contiguous(), eps, alpha, 5, random_start=False)
    assert torch.equal(part, full[h:2 * h])


def test_bf16_train_steps_through_the_facade_and_the_cli(tmp_path):
    """train_loras.py:303-315 with precision = bf16: the loss falls over Adam steps, no optimizer step is dropped, and
    whitebox_attacks.py --precision bf16 never takes the fp32 redo path."""
    import train_loras
    import whitebox_attacks
    P = pkg()
    cfg, w, lora, x, y = make_case(batch=8)
    arch = P.ArchConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, hidden=cfg.hidden, layers=cfg.layers, heads=cfg.heads,
                        mlp=cfg.mlp, num_labels=cfg.num_labels, ln_eps=cfg.ln_eps)
    base = P.ViTForImageClassification(arch, device="cuda:0", precision="bf16")
    base.load_state_dict(w)
    pm = P.setup_peft_lora(base, rank=8, dropout=0.0)
    pm.train()
    opt = P.Adam(pm.parameters(), lr=3e-3, model=pm)
    crit = torch.nn.CrossEntropyLoss()
    xn = O.normalise(x).cuda()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(pm.base_model(pixel_values=xn).logits, y.cuda())
        loss.backward()
        pm._vit._engine().check()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 1e-3, losses
    res = train_loras.main(["--output_dir", str(tmp_path / "loras"), "--attacks", "pgd", "--ranks", "4", "--epochs", "1", "--synthetic", "24",
                            "--arch", "tiny", "--batch_size", "8", "--pgd-inner-steps", "2", "--precision", "bf16"])
    assert res["google_vit"]["mapillary"]["pgd"][4]["train_loss"]
    (tele,) = [v for (d, r), v in train_loras.TELEMETRY.items() if d.startswith(str(tmp_path)) and r == 4]
    assert tele == {"fp16_skipped_steps": 0, "optimizer_steps": 3, "precision": "bf16"}
    whitebox_attacks.main(["--models", "google_vit", "--sources", "mapillary", "--synthetic", "16", "--arch", "tiny", "--attacks", "fgsm", "pgd",
                           "--pgd_iters", "3", "--batch_size", "8", "--output_dir", str(tmp_path / "adv"), "--precision", "bf16", "--splits", "test"])
