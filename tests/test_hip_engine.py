"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs, in both
precisions of vl_config:

  * "f32"  every operand fp32 (exact-f32 MFMA): held to TOL["f32"] = 1e-4 -- north_star allows 1e-3;
  * "f16"  fp16 operands, fp32 accumulation: held to 4e-3 on logits / activations and 6e-3 on gradients --
           north_star allows 1e-2 for the 16-bit path.  (bf16 operands sat AT 1e-2 on the ViT-B input gradient:
           tools/error_budget.py; that is why the 16-bit type is fp16.)

Every comparison is against the fp32 oracle (the reference arithmetic); the oracle's sim16 mode (fp16 round trips
where the kernels store 16-bit values) is used additionally where a tighter bound on kernel LOGIC is wanted.
"""
import pytest
import torch

from helpers import O, make_case, make_engine, rel_l2
from helpers import pkg as pkg_mod

pytestmark = pytest.mark.gpu

PRECS = ["f16", "f32"]
TOL_ACT = {"f16": 4e-3, "f32": 1e-4}       # logits, residual stream, qkv, ctx, loss
TOL_GRAD = {"f16": 6e-3, "f32": 1e-4}      # dLoss/dx, LoRA A/B gradients, classifier gradients
TOL_SIM = 2e-3                             # f16 path vs the fp16-simulating oracle (accumulation order only)


def _trace(cfg, w, lora, x_norm, sim):
    tr = {}
    logits = O.vit_forward(w, cfg, x_norm, lora, sim16=sim, trace=tr)
    tr["logits"] = logits
    return tr


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("image_size,batch", [(64, 4), (224, 3)])
@pytest.mark.parametrize("with_lora", [False, True])
def test_forward_stagewise(image_size, batch, with_lora, prec):
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8 if with_lora else 0)
    eng = make_engine(cfg, w, lora, precision=prec)
    xn = O.normalise(x)
    eng.set_dead_rows(False)            # every row of the last layer too: its saved activations are compared below
    logits = eng.forward(xn.cuda(), normalise=False)
    torch.cuda.synchronize()
    tr = _trace(cfg, w, lora, xn, sim=False)
    B, T, D = batch, cfg.tokens, cfg.hidden
    report = []
    for i in range(2 * cfg.layers + 1):
        got = eng.debug_tensor("xs", i).float().cpu().view(B, T, D)
        report.append((f"xs{i}", rel_l2(got, tr[f"xs{i}"])))
    for l in range(cfg.layers):
        got = eng.debug_tensor("qkv", l).float().cpu().view(B, T, 3 * D)
        report.append((f"qkv{l}", rel_l2(got, tr[f"qkv{l}"])))
        got = eng.debug_tensor("ctx", l).float().cpu().view(B, T, D)
        report.append((f"ctx{l}", rel_l2(got, tr[f"ctx{l}"])))
    report.append(("logits", rel_l2(logits.cpu(), tr["logits"])))
    bad = [(n, e) for n, e in report if not (e < TOL_ACT[prec])]
    assert not bad, f"stages off: {bad}\nall: {report}"
    if prec == "f16":       # kernel logic: against the oracle that rounds where the kernels round
        sim = _trace(cfg, w, lora, xn, sim=True)
        assert rel_l2(logits.cpu(), sim["logits"]) < TOL_SIM
        # default route: the last layer on the CLS rows only -- same logits, and its CLS rows of the residual stream
        eng.set_dead_rows(True)
        logits2 = eng.forward(xn.cuda(), normalise=False)
        torch.cuda.synchronize()
        # (the compact CLS rows keep an fp32 stream through the last layer, the full-row route rounds it to 16 bits twice more)
        assert rel_l2(logits2.cpu(), logits.cpu()) < 1.5e-3 and rel_l2(logits2.cpu(), tr["logits"]) < TOL_ACT[prec]
        L2 = 2 * cfg.layers
        for name, ref in (("cls_x1", tr[f"xs{L2 - 1}"][:, 0]), ("cls_x2", tr[f"xs{L2}"][:, 0])):
            got = eng.debug_tensor(name, 0).float().cpu().view(B, D)
            assert rel_l2(got, ref) < TOL_ACT[prec], (name, rel_l2(got, ref))


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("image_size,batch", [(64, 4), (224, 3)])
@pytest.mark.parametrize("with_lora", [False, True])
def test_loss_and_input_grad(image_size, batch, with_lora, prec):
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8 if with_lora else 0)
    eng = make_engine(cfg, w, lora, precision=prec)
    logits = eng.forward(x.cuda(), normalise=True)
    loss = eng.loss_ce(y.cuda())
    gx, _ = eng.backward(True, False, tuple(x.shape))
    torch.cuda.synchronize()
    l_ref, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
    assert abs(loss.item() - l_ref.item()) < TOL_ACT[prec] * max(1.0, abs(l_ref.item()))
    assert rel_l2(logits.cpu(), lg_ref) < TOL_ACT[prec]
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD[prec], rel_l2(gx.cpu(), g_ref)
    # FGSM / PGD consume sign(g): agreement over ALL pixels, and where the gradient is not in the rounding noise
    agree_all = (torch.sign(gx.cpu()) == torch.sign(g_ref)).float().mean().item()
    big = g_ref.abs() > 0.05 * g_ref.abs().mean()
    agree = (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item()
    assert agree > 0.9995 and agree_all > (0.998 if prec == "f16" else 0.99995), (agree, agree_all)


@pytest.mark.parametrize("r,targets", [(4, ("q", "v")), (8, ("q",)), (16, ("q", "v")), (16, ("o", "fc2")), (32, ("q", "o", "fc2")), (8, ("fc1", "fc2")),
                                       (32, ("q", "k", "v", "o", "fc2")), (32, ("q", "k", "v", "o", "fc1", "fc2"))])
@pytest.mark.parametrize("prec", PRECS)
def test_lora_down_fusion_variants(r, targets, prec):
    """The LoRA down-projections are computed by different kernels depending on r * modules (fused into the
    LayerNorm forward / backward rows for <= 8 / <= 16 columns, skinny GEMM otherwise): every route must agree
    with the oracle on logits and on the input gradient.  r = 32 on the reference's default target set
    (train_loras.py:441 --ranks default [8, 16, 32]; :81 target_modules) puts 96 LoRA columns into the fused qkv
    projection: TWO extra K tiles."""
    cfg, w, lora, x, y = make_case(image_size=64, batch=4, r=r, targets=targets)
    eng = make_engine(cfg, w, lora, precision=prec)
    logits = eng.forward(x.cuda(), normalise=True)
    eng.loss_ce(y.cuda())
    gx, _ = eng.backward(True, False, tuple(x.shape))
    torch.cuda.synchronize()
    _, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
    assert rel_l2(logits.cpu(), lg_ref) < TOL_ACT[prec]
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD[prec]
    # the LoRA branch must matter in this case (otherwise the test proves nothing)
    _, g_nolora, _ = O.loss_and_input_grad(w, cfg, x, y, None)
    assert rel_l2(g_ref, g_nolora) > 5e-2


@pytest.mark.parametrize("hidden,heads,mlp,layers,r", [(256, 4, 1024, 2, 8), (1024, 16, 4096, 1, 16)])
@pytest.mark.parametrize("prec", PRECS)
def test_other_widths_run_the_256_row_gemm(hidden, heads, mlp, layers, r, prec):
    """Widths other than ViT-B's: hidden 256 (K only 4-5 tiles deep: the shortest pipeline the 256-row GEMM accepts)
    and the ViT-L/16 width (hidden 1024, 16 heads, mlp 4096, LoRA r = 16: BASELINE config 5's model family)."""
    cfg, w, lora, x, y = make_case(image_size=224, hidden=hidden, heads=heads, mlp=mlp, layers=layers, batch=3, r=r,
                                   std=0.03 if hidden == 1024 else 0.05)
    eng = make_engine(cfg, w, lora, precision=prec)
    logits = eng.forward(x.cuda(), normalise=True)
    eng.loss_ce(y.cuda())
    gx, _ = eng.backward(True, False, tuple(x.shape))
    torch.cuda.synchronize()
    _, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
    assert rel_l2(logits.cpu(), lg_ref) < TOL_ACT[prec]
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD[prec]
    big = g_ref.abs() > 0.05 * g_ref.abs().mean()
    assert (torch.sign(gx.cpu())[big] == torch.sign(g_ref)[big]).float().mean().item() > 0.9995


@pytest.mark.parametrize("gain", [4.0, 32.0, 512.0])
def test_fp16_backward_out_of_range_is_flagged_never_silent(gain):
    """Round-2 ADVICE: every parity fixture has unit-gain weights, a fine-tuned checkpoint need not.  LayerNorm gains and the
    classifier scaled by `gain` amplify the backward chain per layer; the fp16 path must then EITHER stay finite and accurate
    (the per-image power-of-two gradient scale absorbs the classifier factor) OR report VL_ERR_NONFINITE at the next check --
    a clamped (f2h_sat), infinite or NaN gradient that goes unreported is the failure this test exists for."""
    P = pkg_mod()
    cfg, w, lora, x, y = make_case(image_size=64, batch=4, r=8, layers=4)
    w2 = {k: (v * gain if (k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight") or k == "classifier.weight")
              else v) for k, v in w.items()}
    eng = make_engine(cfg, w2, lora, precision="f16")
    eng.forward(x.cuda(), normalise=True)
    eng.loss_ce(y.cuda())
    gx, _ = eng.backward(True, False, tuple(x.shape))
    flagged = False
    try:
        eng.check()
    except P.NonFiniteGradient:
        flagged = True
    _, g_ref, _ = O.loss_and_input_grad(w2, cfg, x, y, lora)
    if not flagged:
        assert torch.isfinite(gx).all()
        # "accurate" for a network whose gains make it ill-conditioned: within 1.5x of what 16-bit STORAGE alone costs on this
        # network (the oracle with fp16 round trips at the kernels' storage sites; at gain 4 this 128-wide net amplifies every
        # rounding 9x -- tools/error_budget_streams.py: 1.1e-3 -> 9.8e-3 before, 2.0e-2 with the 16-bit residual stream)
        _, g_sim, _ = O.loss_and_input_grad(w2, cfg, x, y, lora, sim16=True)
        bound = max(2 * TOL_GRAD["f16"], 1.5 * rel_l2(g_sim, g_ref))
        assert rel_l2(gx.cpu(), g_ref) < bound, (gain, rel_l2(gx.cpu(), g_ref), bound)
    if gain >= 512.0:
        assert flagged, "a 512x gain per LayerNorm over 4 layers cannot fit fp16: the overflow must be reported"
        # the fp32 mode is the documented way out and has the range
        e32 = make_engine(cfg, w2, lora, precision="f32")
        e32.forward(x.cuda(), normalise=True)
        e32.loss_ce(y.cuda())
        g32, _ = e32.backward(True, False, tuple(x.shape))
        e32.check()
        # (with 512x LayerNorm gains the softmax saturates and the network is ill-conditioned: two fp32 programs agree to a few
        #  per cent only -- the point here is range, not parity)
        assert torch.isfinite(g32).all() and rel_l2(g32.cpu(), g_ref) < 0.1
    # the flag is cleared by reading it: the engine is usable again
    eng.check()


def test_ragged_batches_replanning_and_empty_input():
    """Batch sizes the tile sizes do not divide (1, 5, 7 images = 17 .. 119 token rows), a workspace that grows when a
    larger batch arrives and is reused for smaller ones, and an empty batch (refused, not a crash)."""
    cfg, w, lora, x, y = make_case(image_size=64, batch=7)
    eng = make_engine(cfg, w, lora)
    ref = O.vit_forward(w, cfg, O.normalise(x), lora)
    for b in (1, 5, 7, 2):                       # grows the plan at 5 and 7, reuses it at 2
        got = eng.forward(x[:b].cuda(), normalise=True).cpu()
        assert rel_l2(got, ref[:b]) < TOL_ACT["f16"], b
    # attack on a ragged batch stays in the eps-ball and is the slice of the larger batch
    adv7 = eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=False).clone()
    adv3 = eng.pgd_attack(x[2:5].cuda(), y[2:5].cuda(), 8 / 255, 2 / 255, 3, random_start=False)
    assert (adv7 - x.cuda()).abs().max().item() <= 8 / 255 + 1e-6
    assert (torch.sign(adv3 - x[2:5].cuda()) == torch.sign(adv7[2:5] - x[2:5].cuda())).float().mean().item() > 0.99
    with pytest.raises(Exception):
        eng.forward(x[:0].cuda(), normalise=True)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(2, 3, 32, 32).cuda(), normalise=True)
    # the engine still works after the refused calls
    assert rel_l2(eng.forward(x[:3].cuda(), normalise=True).cpu(), ref[:3]) < TOL_ACT["f16"]


def test_unnormalised_forward_equals_normalised_input():
    cfg, w, lora, x, y = make_case(batch=2)
    eng = make_engine(cfg, w, lora)
    a = eng.forward(x.cuda(), normalise=True).clone()
    b = eng.forward(O.normalise(x).cuda(), normalise=False)
    assert rel_l2(a.cpu(), b.cpu()) < 2e-3


@pytest.mark.parametrize("prec", PRECS)
def test_merged_matches_fused(prec):
    cfg, w, lora, x, y = make_case(batch=3)
    fused = make_engine(cfg, w, lora, precision=prec).forward(x.cuda(), normalise=True).cpu()
    merged = make_engine(cfg, w, lora, merged=True, precision=prec).forward(x.cuda(), normalise=True).cpu()
    ref = O.vit_forward(w, cfg, O.normalise(x), lora)
    assert rel_l2(merged, ref) < TOL_ACT[prec]
    assert rel_l2(fused, merged) < TOL_ACT[prec]


def test_pgd_step_kernel_bit_exact():
    torch.manual_seed(0)
    n = 3 * 224 * 224 * 2 + 3
    x0 = torch.rand(n)
    adv = (x0 + (torch.rand(n) - 0.5) * 0.05).clamp(0, 1)
    g = torch.randn(n)
    g[::7] = 0.0
    g[1::11] = 1e-30
    g[2::13] = -1e-38
    eps, alpha = 8 / 255, 2 / 255
    want = O.pgd_step(adv, x0, g, eps, alpha)
    cfg, w, lora, _, _ = make_case(batch=1, r=0)
    eng = make_engine(cfg, w)
    a = adv.cuda()
    eng.pgd_step(a, x0.cuda(), g.cuda(), eps, alpha)
    assert torch.equal(a.cpu(), want)
    # FGSM form: x0 == adv, alpha == eps  (whitebox_attacks.py:32-36)
    a = x0.cuda().clone()
    eng.pgd_step(a, x0.cuda(), g.cuda(), eps, eps)
    assert torch.equal(a.cpu(), torch.clamp(x0 + eps * torch.sign(g), 0, 1))


@pytest.mark.parametrize("prec", PRECS)
def test_pgd_attack_matches_oracle_trajectory(prec):
    cfg, w, lora, x, y = make_case(batch=4)
    eng = make_engine(cfg, w, lora, precision=prec)
    eps, alpha, steps = 8 / 255, 2 / 255, 5
    adv = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, steps, random_start=False).cpu()
    ref = O.pgd(w, cfg, x, y, eps, alpha, steps, lora)
    assert (adv - x).abs().max().item() <= eps + 1e-6
    assert adv.min().item() >= 0 and adv.max().item() <= 1
    # sign() is discontinuous and PGD feeds each iterate back: compare pixels, allowing the few that flipped
    same = ((adv - ref).abs() < 1e-6).float().mean().item()
    assert same > (0.985 if prec == "f16" else 0.9995), same
    # the attack must be as strong as the oracle's
    l_hip = O.loss_and_input_grad(w, cfg, adv, y, lora)[0].item()
    l_ref = O.loss_and_input_grad(w, cfg, ref, y, lora)[0].item()
    l_clean = O.loss_and_input_grad(w, cfg, x, y, lora)[0].item()
    assert l_hip > l_clean
    assert abs(l_hip - l_ref) < 0.05 * abs(l_ref - l_clean) + 1e-3, (l_hip, l_ref, l_clean)
    # graph replay == eager launches, bit for bit
    import os
    os.environ["VITLORA_NO_GRAPH"] = "1"
    try:
        eng2 = make_engine(cfg, w, lora, precision=prec)
        adv2 = eng2.pgd_attack(x.cuda(), y.cuda(), eps, alpha, steps, random_start=False).cpu()
    finally:
        os.environ.pop("VITLORA_NO_GRAPH")
    assert torch.equal(adv, adv2)


def test_pgd_random_start_is_seeded_and_bounded():
    cfg, w, lora, x, y = make_case(batch=2, r=0)
    eng = make_engine(cfg, w)
    eps = 8 / 255
    a = torch.empty_like(x).cuda()
    b = torch.empty_like(x).cuda()
    eng.pgd_init(a, x.cuda(), eps, seed=3)
    eng.pgd_init(b, x.cuda(), eps, seed=3)
    assert torch.equal(a, b)
    eng.pgd_init(b, x.cuda(), eps, seed=4)
    assert not torch.equal(a, b)
    d = (a.cpu() - x)
    assert d.abs().max().item() <= eps + 1e-7
    inside = (x > eps) & (x < 1 - eps)
    u = d[inside] / eps
    assert abs(u.mean().item()) < 0.02 and abs(u.std().item() - 3 ** -0.5) < 0.02


@pytest.mark.parametrize("r", [8, 32])
@pytest.mark.parametrize("prec", PRECS)
def test_lora_train_grads(prec, r):
    cfg, w, lora, x, y = make_case(batch=4, r=r, targets=("q", "k", "v", "o", "fc1", "fc2"))
    eng = make_engine(cfg, w, lora, precision=prec)
    xn = O.normalise(x)
    logits = eng.forward(xn.cuda(), normalise=False, train=True)
    loss = eng.loss_ce(y.cuda())
    _, gp = eng.backward(False, True)
    torch.cuda.synchronize()
    l_ref, lg_ref, grads = O.lora_train_grads(w, cfg, xn, y, lora)
    assert abs(loss.item() - l_ref.item()) < TOL_ACT[prec] * abs(l_ref.item())
    # walk the flat layout: same order as the library (layer, target) -> A, B ; classifier
    flat = eng.flat
    base = flat.data_ptr()
    worst = 0.0
    for i in range(cfg.layers):
        for t in lora.targets:
            for which in ("A", "B"):
                v = eng.param(i, t, which)
                off = (v.data_ptr() - base) // 4
                got = gp[off:off + v.numel()].view(v.shape).cpu()
                e_ref = rel_l2(got, grads[(which, i, t)])
                worst = max(worst, e_ref)
                assert e_ref < TOL_GRAD[prec], (i, t, which, e_ref)
    for which in ("weight", "bias"):
        v = eng.param(-1, "", which)
        off = (v.data_ptr() - base) // 4
        got = gp[off:off + v.numel()].view(v.shape).cpu()
        assert rel_l2(got, grads[("cls", which)]) < TOL_GRAD[prec]


@pytest.mark.parametrize("prec", ["f16", "bf16", "f32"])
def test_lora_gradients_are_bit_reproducible_and_need_no_zeroed_output(prec):
    """Two backward passes of the same train step give the same LoRA / classifier gradient BIT FOR BIT (round 5: the weight
    gradients are summed as per-chunk partial blocks in chunk order, csrc/lora_grad.hip; rounds 1-4 used float atomics and two
    identical training runs were not bit-reproducible).  224-pixel images x 8 = 1 576 token rows = 4 chunks of 512, so the
    chunk reduction is exercised; the output buffer is handed over full of NaN to show that every element is written
    (train_loras.py:310-314 -- the determinism SURVEY section 4 (iv) asks of the data-parallel gradient)."""
    cfg, w, lora, x, y = make_case(image_size=224, batch=8, r=8, targets=("q", "k", "v", "o", "fc2"))
    eng = make_engine(cfg, w, lora, precision=prec)
    xn = O.normalise(x).cuda()
    outs = []
    for rep in range(3):
        eng.forward(xn, normalise=False, train=True)
        eng.loss_ce(y.cuda())
        gp = torch.full_like(eng.flat, float("nan"))
        import ctypes as C
        rc = eng.lib.vl_backward(eng.h, C.c_void_p(0), C.c_void_p(gp.data_ptr()), eng._stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert torch.isfinite(gp).all()
        outs.append(gp.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    if prec != "bf16":
        l_ref, lg_ref, grads = O.lora_train_grads(w, cfg, O.normalise(x), y, lora)
        v = eng.param(0, "q", "A")
        off = (v.data_ptr() - eng.flat.data_ptr()) // 4
        assert rel_l2(outs[0][off:off + v.numel()].view(v.shape).cpu(), grads[("A", 0, "q")]) < TOL_GRAD[prec]


@pytest.mark.parametrize("prec", PRECS)
def test_lora_dropout_train_mode_matches_oracle_with_same_masks(prec):
    """lora_dropout > 0 (the reference trains with 0.1, train_loras.py:79): the HIP path's masks are
    read back and handed to the oracle, so the stochastic op is checked exactly."""
    cfg, w, lora, x, y = make_case(batch=4, targets=("q", "k", "v", "o", "fc1", "fc2"))
    p = 0.25
    eng = make_engine(cfg, w, lora, dropout=p, precision=prec)
    eng.set_dropout_seed(77)
    xn = O.normalise(x)
    logits = eng.forward(xn.cuda(), normalise=False, train=True)
    loss = eng.loss_ce(y.cuda())
    masks = {(l, pr): eng.dropout_mask(l, pr, 4).cpu() for l in range(cfg.layers) for pr in ("qkv", "o", "fc1", "fc2")}
    gx, gp = eng.backward(True, True, tuple(x.shape))
    torch.cuda.synchronize()
    m0 = masks[(0, "qkv")]
    keep = (m0 > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.02 and torch.allclose(m0[m0 > 0], torch.tensor(1 / (1 - p)))
    assert not torch.equal(masks[(0, "qkv")], masks[(1, "qkv")]) and not torch.equal(masks[(0, "qkv")], masks[(0, "o")])
    l_ref, lg_ref, grads = O.lora_train_grads(w, cfg, xn, y, lora, drop_masks=masks)
    assert rel_l2(logits.cpu(), lg_ref) < TOL_ACT[prec]
    # without the masks the oracle must NOT match: the masks really were applied
    assert rel_l2(logits.cpu(), O.vit_forward(w, cfg, xn, lora)) > 1.5 * rel_l2(logits.cpu(), lg_ref)
    base = eng.flat.data_ptr()
    for i in range(cfg.layers):
        for t in lora.targets:
            for which in ("A", "B"):
                v = eng.param(i, t, which)
                off = (v.data_ptr() - base) // 4
                got = gp[off:off + v.numel()].view(v.shape).cpu()
                assert rel_l2(got, grads[(which, i, t)]) < TOL_GRAD[prec], (i, t, which)
    # a new forward draws new masks; eval-mode forward ignores dropout
    eng.forward(xn.cuda(), normalise=False, train=True)
    assert not torch.equal(eng.dropout_mask(0, "qkv", 4).cpu(), m0)
    ev = eng.forward(xn.cuda(), normalise=False, train=False).cpu()
    assert rel_l2(ev, O.vit_forward(w, cfg, xn, lora)) < TOL_ACT[prec]


def test_adam_and_quantiser():
    torch.manual_seed(1)
    cfg, w, lora, x, y = make_case(batch=1, r=0)
    eng = make_engine(cfg, w)
    p0 = torch.randn(5000)
    p, m, v = p0.cuda().clone(), torch.zeros(5000).cuda(), torch.zeros(5000).cuda()
    q, qm, qv = p0.clone(), torch.zeros(5000), torch.zeros(5000)
    for t in range(1, 4):
        g = torch.randn(5000)
        eng.adam_step(p, g.cuda(), m, v, 1e-4, 0.9, 0.999, 1e-8, t)
        q, qm, qv = O.adam_step(q, g, qm, qv, t)
    assert torch.allclose(p.cpu(), q, rtol=1e-5, atol=1e-7)
    img = torch.rand(2, 3, 17, 19) * 1.2 - 0.1
    assert torch.equal(eng.quantize_u8(img.cuda()).cpu(), O.save_images_quant(img))


@pytest.mark.parametrize("prec", ["f16"])
@pytest.mark.parametrize("image_size,batch", [(64, 5), (224, 3)])
@pytest.mark.parametrize("r,targets", [(8, ("q", "k", "v", "o", "fc2")), (4, ("q", "v")), (8, ("k", "o")), (0, ())])
def test_per_image_attention_kernels_with_fused_lora_down(image_size, batch, r, targets, prec, monkeypatch):
    """The persistent per-image attention kernels (chosen by themselves for chip-filling batches, forced here) sum the
    LoRA down projections t = ctx Ad^T and u = dqkv Bd^T over heads in registers: same parity bar as the per-head
    kernels + skinny GEMMs they replace, including the LoRA gradients that read t and u."""
    monkeypatch.setenv("VITLORA_ATTN_IMG", "1")
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=r, targets=targets)
    eng = make_engine(cfg, w, lora, precision=prec)
    xn = O.normalise(x)
    train = r > 0
    logits = eng.forward(xn.cuda(), normalise=False, train=train).cpu()
    eng.loss_ce(y.cuda())
    gx, gp = eng.backward(True, train, tuple(x.shape))
    torch.cuda.synchronize()
    _, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, xn, y, lora, normalised=True)
    assert rel_l2(logits, lg_ref) < TOL_ACT[prec]
    assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD[prec]
    if train:
        _, _, grads = O.lora_train_grads(w, cfg, xn, y, lora)
        base = eng.flat.data_ptr()
        for i in range(cfg.layers):
            for t in lora.targets:
                for which in ("A", "B"):
                    v = eng.param(i, t, which)
                    off = (v.data_ptr() - base) // 4
                    got = gp[off:off + v.numel()].view(v.shape).cpu()
                    assert rel_l2(got, grads[(which, i, t)]) < TOL_GRAD[prec], (i, t, which)
    # same engine configuration through the per-head kernels: the two forms agree far below the parity bar
    monkeypatch.setenv("VITLORA_ATTN_IMG", "0")
    eng2 = make_engine(cfg, w, lora, precision=prec)
    logits2 = eng2.forward(xn.cuda(), normalise=False, train=train).cpu()
    eng2.loss_ce(y.cuda())
    gx2, _ = eng2.backward(True, False, tuple(x.shape))
    assert rel_l2(logits, logits2) < 1e-3 and rel_l2(gx.cpu(), gx2.cpu()) < 2e-3      # two fp16 kernels, each within TOL of the oracle


@pytest.mark.parametrize("shape", [(128 * 50, 3072, 768, 64), (128 * 100, 768, 3072, 64), (128 * 110, 3072, 768, 0),
                                   (128 * 3, 768, 768, 64), (128 * 37, 2304, 768, 64)])
def test_main_gemm_kernels_agree_bitwise(shape):
    """The three forms of the main GEMM on the same random operands: the persistent 256-row kernel (pp_mode 0) and the
    ping-pong kernel (pp_mode 1: epilogue of one tile under the main loop of the next) against the plain 128-row kernel.
    Same MFMA, same K order: the results must be IDENTICAL for the store / residual / gelu' epilogues; the GELU forward
    may differ by one fp16 ulp of its largest outputs (16-column and 4-column forms of the same epilogue, contracted differently).  The sizes
    cover full pair rounds, a single-tile tail round, a pair tail round, one tile per workgroup, and N = 9 tiles."""
    import ctypes as C
    import importlib
    from helpers import PKG
    lib = importlib.import_module(PKG + "._lib").load()
    M, N, K1, K2 = shape
    for epi, name in ((0, "store_h16"), (1, "resid_f32"), (2, "gelu"), (3, "gelu_bwd"), (6, "store_f32"), (10, "resid_h16")):
        for mode in (0, 1):
            d = C.c_float(-1.0)
            assert lib.vl_check_gemm(M, N, K1, K2, epi, mode, C.byref(d)) == 0, lib.vl_last_error()
            assert d.value <= (4e-3 if epi == 2 else 0.0), (name, mode, d.value)
    if K2 == 64 and K1 >= 768:
        # the LoRA down projection INSIDE the ping-pong GEMM (pp_mode 2 / 3 = 16 / 32 columns): t and the result must equal the
        # skinny-GEMM route bit for bit with the plain store; with the residual-add epilogue the bias rides in column 63 of the
        # LoRA K tile (added inside the last MFMA step instead of after it: fp32 association differs -> at most one fp16 ulp of
        # the largest outputs, |C| < 64 here)
        for epi, tol in ((0, 0.0), (10, 0.04)):
            for mode in (2, 3):
                d = C.c_float(-1.0)
                assert lib.vl_check_gemm(M, N, K1, K2, epi, mode, C.byref(d)) == 0, lib.vl_last_error()
                assert 0.0 <= d.value <= tol, (epi, mode, d.value)




def test_residual_add_placement_modes_agree():
    """The residual add of the 16-bit stream can sit in the o / fc2 GEMM epilogue (EPI_RESID_H16: x' = round16(x + acc + bias); default for
    the attention output projection = mode 1, mode 2 adds fc2) or in the LayerNorm that follows (x' = round16(x + round16(acc + bias)), `resid_epi` 0; 1 = o projection only).
    All three are held to the oracle at the usual tolerances and agree with each other to fp16 rounding."""
    cfg, w, lora, x, y = make_case(image_size=224, hidden=256, heads=4, mlp=1024, layers=2, batch=3, r=8)
    _, g_ref, lg_ref = O.loss_and_input_grad(w, cfg, x, y, lora)
    eng = make_engine(cfg, w, lora, precision="f16")
    outs = []
    for mode in (2, 1, 0):
        eng.set_option("resid_epi", mode)
        logits = eng.forward(x.cuda(), normalise=True).cpu()
        eng.loss_ce(y.cuda())
        gx, _ = eng.backward(True, False, tuple(x.shape))
        torch.cuda.synchronize()
        assert rel_l2(logits, lg_ref) < TOL_ACT["f16"], mode
        assert rel_l2(gx.cpu(), g_ref) < TOL_GRAD["f16"], mode
        outs.append((logits, gx.cpu()))
    eng.set_option("resid_epi", 1)
    for a, b in ((0, 1), (0, 2)):
        assert rel_l2(outs[a][0], outs[b][0]) < 2e-3 and rel_l2(outs[a][1], outs[b][1]) < 3e-3
    assert not torch.equal(outs[0][0], outs[2][0])       # the switch really selected another path


@pytest.mark.parametrize("image_size,batch", [(224, 3), (64, 4), (224, 64)])
def test_ring_and_two_phase_attention_backward_agree(image_size, batch, monkeypatch):
    """The single-pass ("ring") per-image attention backward against the two-phase form it replaced (`attn_ring` 0), on the
    same engine and inputs, three times over: input gradient, LoRA gradients, the fused u = dqkv Bd^T and dqkv itself agree to
    fp16 rounding, both are within the parity bar of the oracle, and the ring form is finite and repeatable (the round-3 NaN
    from uninitialised LDS rows showed up once in a few hundred runs of exactly this comparison; tools/attn_ring_check.py)."""
    monkeypatch.setenv("VITLORA_ATTN_IMG", "1")
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8, layers=2)
    eng = make_engine(cfg, w, lora)
    xn = O.normalise(x)
    _, g_ref, _ = O.loss_and_input_grad(w, cfg, xn, y, lora, normalised=True)
    first = None
    try:
        for rep in range(3):
            outs = {}
            for ring in (0, 1):
                eng.set_option("attn_ring", ring)
                eng.forward(xn.cuda(), normalise=False, train=True)
                eng.loss_ce(y.cuda())
                gx, gp = eng.backward(True, True, tuple(x.shape))
                torch.cuda.synchronize()
                outs[ring] = (gx.cpu(), gp.cpu(), eng.debug_tensor("u", 0).float().cpu(), eng.debug_tensor("dqkv", 0).float().cpu())
            assert all(torch.isfinite(t).all() for t in outs[1])
            assert rel_l2(outs[1][0], outs[0][0]) < 2e-3 and rel_l2(outs[1][1], outs[0][1]) < 3e-3
            assert rel_l2(outs[1][2], outs[0][2]) < 3e-3 and rel_l2(outs[1][3], outs[0][3]) < 3e-3
            assert rel_l2(outs[1][0], g_ref) < TOL_GRAD["f16"] and rel_l2(outs[0][0], g_ref) < TOL_GRAD["f16"]
            if first is None:
                first = outs[1]
            else:
                assert torch.equal(first[0], outs[1][0])                   # the input gradient is bit-reproducible
                assert rel_l2(outs[1][1], first[1]) < 1e-5                 # (LoRA gradients: fp32 atomics over token chunks, csrc/lora_grad.hip)
    finally:
        eng.set_option("attn_ring", 1)


@pytest.mark.parametrize("prec", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("image_size,batch,mlp,hidden,r,targets,merged,dropout", [
    (64, 4, 256, 128, 8, ("q", "k", "v", "o", "fc2"), False, 0.0),
    (224, 3, 256, 128, 8, ("q", "k", "v", "o", "fc2"), False, 0.0),
    (224, 3, 128, 128, 4, ("q", "v"), False, 0.1),
    (64, 5, 1024, 256, 16, ("q", "k", "v", "o", "fc1", "fc2"), False, 0.1),
    (64, 2, 512, 128, 8, ("q", "k", "v", "o", "fc2"), True, 0.0),
    (64, 3, 512, 128, 32, ("q", "k", "v", "o", "fc2"), False, 0.1),
    (224, 1, 384, 384, 0, (), False, 0.0),
    (224, 2, 4096, 1024, 16, ("q", "k", "v", "o", "fc2"), False, 0.0)])
def test_workspace_is_never_written_outside_its_planned_bytes(prec, image_size, batch, mlp, hidden, r, targets, merged, dropout):
    """The workspace the caller hands over (vl_set_workspace) is the ONLY scratch memory the library may touch, and nothing in it
    is read before it is written.  (a) Guard bands of 1 MiB on both sides of the planned bytes keep their pattern through
    forward / loss / backward (input and parameter gradients, eval and train plans) and a PGD attack -- for architectures
    whose MLP is narrower than the flattened patch (3 P^2 = 768 columns): round 4 found the fp32 mode's patch-gradient product
    running past the end of a buffer that was sized for the MLP alone (it corrupted whatever tensor the allocator had placed
    behind the workspace).  (b) The same calls on a workspace pre-filled with 0x00 and with 0xFF bytes (NaN in every float
    format) give bit-identical logits, loss, input gradient and attack result: an uninitialised read would show."""
    import ctypes as C
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, mlp=mlp, hidden=hidden, heads=hidden // 64, r=r, targets=targets)
    eng = make_engine(cfg, w, lora, merged=merged, dropout=dropout, precision=prec)
    guard = 1 << 20
    for train in ((False,) if (r == 0 or merged) else (False, True)):
        n = eng.workspace_bytes(batch, train)
        buf = torch.full((n + 2 * guard + 512,), 0xA5, dtype=torch.uint8, device="cuda")
        base = (buf.data_ptr() + guard + 255) // 256 * 256
        off = base - buf.data_ptr()
        eng.plan(batch, train)
        eng._ws = buf                                            # the engine keeps the tensor alive; the library gets the inner range
        assert eng.lib.vl_set_workspace(eng.h, C.c_void_p(base), n) == 0
        outs = []
        for fill in (0x00, 0xFF):
            buf[off:off + n] = fill
            if dropout > 0:
                eng.set_dropout_seed(7)
            logits = eng.forward(x.cuda(), normalise=True, train=train).clone()
            loss = eng.loss_ce(y.cuda()).clone()
            gx, gp = eng.backward(True, train, tuple(x.shape))
            res = [logits, loss, gx.clone()]
            if gp is not None:
                res.append(gp.clone())       # (LoRA gradients sum with float atomics: compared to rounding, not bit for bit)
            if not train:
                res.append(eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=1).clone())
            torch.cuda.synchronize()
            outs.append(res)
            assert bool((buf[:off] == 0xA5).all()), (prec, train, "bytes BEFORE the workspace were written")
            bad = (buf[off + n:] != 0xA5).nonzero()
            assert bad.numel() == 0, (prec, train, "bytes AFTER the workspace were written", int(bad[0]), int(bad[-1]), int(bad.numel()))
        for k, (a, b) in enumerate(zip(*outs)):
            assert not torch.isnan(b).any(), (prec, train, k, "NaN from a pre-filled workspace")
            if train and k == 3:
                assert rel_l2(a, b) < 1e-5, (prec, train, k)
            else:
                assert torch.equal(a, b), (prec, train, k, "result depends on what the workspace held before", (a != b).sum().item())


def test_plan_refuses_batches_beyond_the_32_bit_operand_offsets():
    """The 16-bit kernels address an operand with 32-bit byte offsets from its base (LDS-DMA: wave-uniform base + per-lane
    offset).  A batch whose widest activation would pass 4 GiB is refused at vl_plan, with the per-call limit in the message --
    not executed with wrapped addresses.  ViT-B's MLP activation: [round_up(197 B, 256), 3072] h16 -> 3 547 images."""
    lib_err = pkg_mod().VitLoraError
    cfg, w, _, _, _ = make_case(image_size=224, hidden=768, heads=12, mlp=3072, layers=1, batch=1, r=0)
    for prec in ("f16", "bf16"):
        eng = make_engine(cfg, w, precision=prec)
        assert eng.workspace_bytes(3547) > 0
        with pytest.raises(lib_err, match="4 GiB"):
            eng.workspace_bytes(3548)
        with pytest.raises(lib_err, match="4 GiB"):
            eng.plan(100000)
    eng = make_engine(cfg, w, precision="f32")          # the fp32 parity kernels index with 64 bits
    assert eng.workspace_bytes(3548) > 0


@pytest.mark.parametrize("attn_img,opts", [("0", {}), ("1", {}), ("1", {"dead_rows": 0}), ("0", {"fuse_pgd": 0}), ("1", {"resid_epi": 0}),
                                           ("0", {"resid_epi": 1, "attn_ring": 0})])
@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_workspace_guard_bands_hold_for_every_kernel_selection(prec, attn_img, opts, monkeypatch):
    """The guard-band / pre-fill check of the test above on the kernel variants a small batch does not select by itself: the
    per-image attention forms (chosen when the batch fills the chip; pinned here by VITLORA_ATTN_IMG), the two-phase attention
    backward, every row of the last layer instead of the CLS rows only, the PGD step as its own kernel, the residual add in the
    LayerNorm instead of the GEMM epilogue."""
    import ctypes as C
    monkeypatch.setenv("VITLORA_ATTN_IMG", attn_img)
    cfg, w, lora, x, y = make_case(image_size=224, batch=3, mlp=256, hidden=128, heads=2, r=8)
    eng = make_engine(cfg, w, lora, precision=prec)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        guard = 1 << 20
        n = eng.workspace_bytes(3, False)
        buf = torch.full((n + 2 * guard + 512,), 0xA5, dtype=torch.uint8, device="cuda")
        base = (buf.data_ptr() + guard + 255) // 256 * 256
        off = base - buf.data_ptr()
        eng.plan(3, False)
        eng._ws = buf
        assert eng.lib.vl_set_workspace(eng.h, C.c_void_p(base), n) == 0
        outs = []
        for fill in (0x00, 0xFF):
            buf[off:off + n] = fill
            logits = eng.forward(x.cuda(), normalise=True).clone()
            eng.loss_ce(y.cuda())
            gx, _ = eng.backward(True, False, tuple(x.shape))
            adv = eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=1).clone()
            torch.cuda.synchronize()
            outs.append((logits, gx.clone(), adv))
            assert bool((buf[:off] == 0xA5).all()) and bool((buf[off + n:] == 0xA5).all()), (prec, attn_img, opts)
        for k, (a, b) in enumerate(zip(*outs)):
            assert not torch.isnan(b).any() and torch.equal(a, b), (prec, attn_img, opts, k)
    finally:
        if "attn_ring" in opts:
            eng.set_option("attn_ring", 1)           # process-wide switch: back to the default


@pytest.mark.parametrize("attn_img", ["0", "1"])
@pytest.mark.parametrize("prec", ["f16", "bf16", "f32"])
def test_results_do_not_depend_on_what_the_previous_kernel_left_in_lds(prec, attn_img, monkeypatch):
    """LDS is not cleared between kernels: whatever the previous kernel on a CU left there is what an uninitialised read sees
    (round 3: padded attention rows read as 0 x NaN after a GEMM had left fp16 pairs behind).  With the "poison_lds" hook every
    launch is preceded by a kernel that fills all 160 KB of every CU's LDS with NaN patterns; logits, input gradient and a PGD
    attack must come out bit for bit as without it, the LoRA gradients (float atomics) to rounding."""
    monkeypatch.setenv("VITLORA_ATTN_IMG", attn_img)
    for image_size, batch in ((224, 3), (64, 5)):
        cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8)
        eng = make_engine(cfg, w, lora, precision=prec)
        outs = []
        n0 = eng.counter("lds_poisons")
        try:
            for poison in (0, 1):
                eng.set_option("poison_lds", poison)
                logits = eng.forward(x.cuda(), normalise=True).clone()
                eng.loss_ce(y.cuda())
                gx, _ = eng.backward(True, False, tuple(x.shape))
                gx = gx.clone()
                adv = eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=5).clone()
                eng.forward(x.cuda(), normalise=True, train=True)
                eng.loss_ce(y.cuda())
                _, gp = eng.backward(False, True)
                torch.cuda.synchronize()
                outs.append((logits, gx, adv, gp.clone()))
        finally:
            eng.set_option("poison_lds", 0)
        assert eng.counter("lds_poisons") - n0 > 100, eng.counter("lds_poisons") - n0        # the hook really ran: one per profiled launch
        for k, (a, b) in enumerate(zip(*outs)):
            assert not torch.isnan(b).any(), (prec, attn_img, image_size, k)
            if k == 3:
                assert rel_l2(a, b) < 1e-5, (prec, attn_img, image_size, k)
            else:
                assert torch.equal(a, b), (prec, attn_img, image_size, k, int((a != b).sum()))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("image_size,batch", [(64, 2), (64, 5), (224, 3), (224, 32)])
def test_two_chain_attack_equals_the_single_chain_attack(prec, image_size, batch):
    """vl_pgd_attack runs batches of 2 .. 191 images as TWO half-batch chains (own activation workspaces, one captured iteration
    each, two streams that meet at the start and the end of the attack: DESIGN.md section 3.7).  Images are independent, so the
    result is the single-chain result bit for bit -- for odd batches (3 + 2), with and without the random start, graph replay and
    eager, and whatever the option says; one capture event serves the whole attack."""
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8)
    eng = make_engine(cfg, w, lora, precision=prec)
    eps, alpha = 8 / 255, 2 / 255
    outs = {}
    for mode in (1, 0, 2):                          # 1: one chain; 0: by batch size (two here); 2: forced
        eng.set_option("pgd_chains", mode)
        c0 = eng.counter("graph_captures")
        a = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 4, random_start=True, seed=9).clone()
        b = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 4, random_start=False).clone()
        assert eng.counter("graph_captures") - c0 == 1
        eng.check()
        outs[mode] = (a, b)
        # the handle still serves the plain API afterwards (the chains leave no half-batch forward behind)
        logits = eng.forward(x.cuda(), normalise=True)
        eng.loss_ce(y.cuda())
        gx, _ = eng.backward(True, False, tuple(x.shape))
        outs[mode] += (logits.clone(), gx.clone())
    for mode in (0, 2):
        for k in range(4):
            assert torch.equal(outs[mode][k], outs[1][k]), (prec, image_size, batch, mode, k, int((outs[mode][k] != outs[1][k]).sum()))
    # eager (no graph) two-chain iterations: fork / join events around every iteration
    eng.set_option("pgd_chains", 0)
    eng.set_option("poison_lds", 1)
    try:
        e = eng.pgd_attack(x.cuda(), y.cuda(), eps, alpha, 4, random_start=True, seed=9)
        assert torch.equal(e, outs[1][0])
    finally:
        eng.set_option("poison_lds", 0)


def test_two_chain_attack_is_planned_away_when_switched_off_before_the_plan():
    """ "pgd_chains" = 1 BEFORE vl_plan: no chain workspaces are carved (the planned bytes shrink) and attacks run as one chain."""
    cfg, w, lora, x, y = make_case(image_size=64, batch=6, r=8)
    eng = make_engine(cfg, w, lora)
    with_chains = eng.workspace_bytes(6)
    ref = eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=2).clone()
    eng2 = make_engine(cfg, w, lora)
    eng2.set_option("pgd_chains", 1)
    assert eng2.workspace_bytes(6) < with_chains
    assert torch.equal(eng2.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=2), ref)
    eng2.set_option("pgd_chains", 2)               # asked for after the plan: there is no chain workspace, it stays one chain
    assert torch.equal(eng2.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 3, random_start=True, seed=2), ref)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("image_size,batch", [(64, 2), (64, 5), (224, 3), (224, 32)])
def test_two_chain_forward_backward_equals_the_single_chain_calls(prec, image_size, batch):
    """ "api_chains": vl_forward(train = 0) / vl_loss_ce / vl_backward_input with the batch split into two half-batch chains (the
    adversarial-patch EoT step runs through these calls) -- logits, loss and input gradient are those of the plain calls bit for
    bit; a train-mode forward, the fp32 mode and a later attack are unaffected."""
    cfg, w, lora, x, y = make_case(image_size=image_size, batch=batch, r=8)
    eng = make_engine(cfg, w, lora, precision=prec)
    ref = []
    for chains in (0, 1, 0):
        eng.set_option("api_chains", chains)
        logits = eng.forward(x.cuda(), normalise=True).clone()
        loss = eng.loss_ce(y.cuda()).clone()
        gx, _ = eng.backward(True, False, tuple(x.shape))
        eng.check()
        ref.append((logits, loss, gx.clone()))
    for k in range(3):
        assert torch.equal(ref[1][k], ref[0][k]) and torch.equal(ref[2][k], ref[0][k]), (prec, image_size, batch, k)
    eng.set_option("api_chains", 1)
    # a different batch size next (graph-free path: nothing cached), then a train-mode step and an attack on the same handle
    h = max(2, batch // 2)
    l2 = eng.forward(x[:h].cuda(), normalise=True).clone()
    eng.set_option("api_chains", 0)
    assert torch.equal(l2, eng.forward(x[:h].cuda(), normalise=True))
    eng.set_option("api_chains", 1)
    eng.forward(x.cuda(), normalise=True, train=True)
    eng.loss_ce(y.cuda())
    _, gp = eng.backward(False, True)
    assert torch.isfinite(gp).all()
    adv = eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 2, random_start=False)
    eng.set_option("api_chains", 0)
    assert torch.equal(adv, eng.pgd_attack(x.cuda(), y.cuda(), 8 / 255, 2 / 255, 2, random_start=False))
    with pytest.raises(Exception):
        eng.set_option("api_chains", 1)
        eng.forward(x.cuda(), normalise=True)
        eng.loss_ce(y.cuda())
        eng.backward(False, True)          # parameter gradients need a train-mode forward
    eng.set_option("api_chains", 0)
