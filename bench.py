#!/usr/bin/env python3
"""Headline benchmark: adversarial images/sec, PGD-20, ViT-B/16 + LoRA r=8, batch 256 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: a full PGD-20 attack
(random start, then 20 x {forward, CE, backward-to-input, fused sign/project step}, each
iteration one hipGraph launch) on 256 synthetic 224x224x3 images that are already resident
in HBM.  Image batches shard over ranks (weak scaling, no data-path collective; the frozen
backbone and the adapters are replicated).  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline      the dominant kernel, timed live with HIP events on the launch stream
                (vl_profile_begin/_report, include/vitlora.h) against the gfx950 dense fp16 / bf16
                MFMA peak; plus `path` = whole-path algorithmic FLOP/s and `pgd_step` =
                the HBM-bound elementwise kernel against the 8 TB/s HBM peak.
  extras        secondary measurements (merged-LoRA attack, one LoRA train step); not `value`.
  cpu_baseline  the CPU oracle (oracle/, a torch restatement of the reference path) timed on
                this box's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes
import importlib
import json
import re
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd"

PEAK_BF16_DENSE = 2.5e15      # FLOP/s dense (no sparsity), the bf16 / fp16 MFMA peak of MI355X_MICROARCH.md (same rate)
PEAK_F32_MATRIX = 157.3e12    # FLOP/s, exact-f32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2), same guide
PEAK_HBM = 8.0e12             # B/s

EPS, ALPHA = 8 / 255, 2 / 255
TARGETS = ("q", "k", "v", "o", "fc2")    # ["query","key","value","output.dense"], train_loras.py:81


def algorithmic_flops_per_image_step(arch, r, targets):
    """fwd + bwd-to-input FLOPs of one PGD iteration for one image (2 FLOP per MAC), the
    accounting of SURVEY.md 8(d) / BASELINE.md section 3."""
    D, M, T, H = arch.hidden, arch.mlp, arch.tokens, arch.heads
    NP, PK = T - 1, 3 * arch.patch_size ** 2
    lin = 2 * T * (4 * D * D + 2 * D * M) * arch.layers
    attn = 4 * H * T * T * (D // H) * arch.layers
    pe = 2 * NP * PK * D
    lora = 0
    for t in targets:
        o, k = (M, D) if t == "fc1" else (D, M) if t == "fc2" else (D, D)
        lora += 2 * T * r * (o + k)
    lora *= arch.layers
    fwd = lin + attn + pe + lora
    bwd = lin + 2.0 * attn + pe + lora      # dgrad only; attention x2 as BASELINE.md section 3 counts it
    return fwd + bwd


def swin_flops_per_image_step(arch, r, targets):
    """fwd + bwd-to-input FLOPs of one PGD iteration of Swin-T for one image (2 FLOP per MAC; the same accounting as the ViT
    function: dgrad only, windowed-attention products x2 in the backward).  HF SwinConfig() shapes (modeling_swin.py:329-368,
    486-505, 584-627): patch 4 -> 56 x 56 tokens of C = 96, stages (2, 2, 6, 2) with C doubling and tokens / 4 at each merge,
    7 x 7 windows, MLP ratio 4."""
    C, tok, w2 = arch.embed_dim, (arch.image_size // arch.patch_size) ** 2, arch.window ** 2
    lin = attn = lora = 0.0
    pe = 2.0 * tok * (3 * arch.patch_size ** 2) * C
    for si, depth in enumerate(arch.depths):
        lin += depth * 2.0 * tok * 12 * C * C                  # qkv 3 C^2, o C^2, fc1 + fc2 8 C^2
        attn += depth * 4.0 * tok * w2 * C                     # Q K^T and P V inside a window: 2 x 49 x C MACs per token
        for t in targets:
            o, k = (4 * C, C) if t == "fc1" else (C, 4 * C) if t == "fc2" else (C, C)
            lora += depth * 2.0 * tok * r * (o + k)
        if si + 1 < len(arch.depths):
            lin += 2.0 * (tok // 4) * (4 * C) * (2 * C)        # patch merging: Linear(4C -> 2C) on a quarter of the tokens
            C, tok = 2 * C, tok // 4
    fwd = lin + attn + pe + lora
    return fwd + (lin + 2.0 * attn + pe + lora)


def extras(P, syn, arch, args, dev, x, y):
    """Secondary measurements on rank 0 at N = 1 (never the headline `value`):
    lora_merged  the same PGD attack with the adapters folded into W first (merge_and_unload,
                 eval_compose.py:110): no LoRA K tiles, no down-projections;
    lora_train_step  train_loras.py:303-315 on one GPU's share of BASELINE config 3 (64 images):
                 forward(train, LoRA dropout 0.1) + CE + backward (LoRA + classifier grads) + fused Adam."""
    res = {}
    spec = P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.0, targets=TARGETS, merged=True)
    eng = P.Engine(arch, spec, device=dev)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    adv = torch.empty_like(x)
    eng.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=1, out=adv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=2, out=adv)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["lora_merged"] = {"value": x.shape[0] / dt, "unit": "img/s", "ms_per_step": 1e3 * dt}
    del eng
    log(f"extras: merged-LoRA attack {x.shape[0] / dt:.1f} img/s")

    bt = 64
    eng = P.Engine(arch, P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.1, targets=TARGETS), device=dev)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    xt, yt = x[:bt], y[:bt]
    m1, m2 = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)

    def train_step(t):
        eng.forward(xt, normalise=True, train=True)
        eng.loss_ce(yt)
        _, g = eng.backward(False, True)
        eng.adam_step(eng.flat, g, m1, m2, 1e-4, 0.9, 0.999, 1e-8, t)
        eng.commit()

    for t in range(1, 3):
        train_step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for t in range(3, 3 + n):
        train_step(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n

    # BASELINE config 3 itself: the adversarial example is made INSIDE the step, against the current adapters (PGD-7, the
    # inner loop of train_loras.py --pgd-inner-steps), then the train step runs on it -- 64 images per GPU
    def adv_train_step(t):
        xa = eng.pgd_attack(xt, yt, EPS, ALPHA, 7, True, seed=t)
        eng.forward(xa, normalise=True, train=True)
        eng.loss_ce(yt)
        _, g = eng.backward(False, True)
        eng.adam_step(eng.flat, g, m1, m2, 1e-4, 0.9, 0.999, 1e-8, t)
        eng.commit()

    for t in range(20, 22):
        adv_train_step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    na = 5
    for t in range(22, 22 + na):
        adv_train_step(t)
    torch.cuda.synchronize()
    dta = (time.perf_counter() - t0) / na
    res["adv_lora_train_step_pgd7"] = {"value": bt / dta, "unit": "img/s", "ms_per_step": 1e3 * dta, "batch": bt,
                                       "what": "PGD-7 against the current adapters + forward(train, dropout 0.1) + CE + LoRA/classifier "
                                               "backward + Adam (BASELINE config 3, one GPU's 64 images)"}
    log(f"extras: adversarial LoRA train step (PGD-7 inner) {1e3 * dta:.2f} ms at batch {bt}")
    if args.vitl:
        # BASELINE config 5's model family (ViT-L/16 + LoRA r = 16, 128 images per GPU): the same PGD attack
        del eng
        archl = P.ArchConfig(hidden=1024, layers=24, heads=16, mlp=4096, num_labels=21)
        engl = P.Engine(archl, P.LoraSpec(r=16, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev)
        engl.load_state_dict(syn.random_state_dict(archl, seed=0))
        for (i, t), (A, Bm) in syn.random_lora(archl, 16, TARGETS, seed=1).items():
            engl.param(i, t, "A").copy_(A)
            engl.param(i, t, "B").copy_(Bm)
        engl.commit()
        xl, yl = x[:128].contiguous(), y[:128].contiguous()
        advl = torch.empty_like(xl)
        engl.pgd_attack(xl, yl, EPS, ALPHA, args.pgd_steps, random_start=True, seed=1, out=advl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engl.pgd_attack(xl, yl, EPS, ALPHA, args.pgd_steps, random_start=True, seed=2, out=advl)
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t0
        fl = algorithmic_flops_per_image_step(archl, 16, TARGETS)
        res["vit_l16_lora_r16_pgd"] = {"value": 128 / dtl, "unit": "img/s", "ms_per_step": 1e3 * dtl, "batch": 128, "dtype": "f16",
                                       "tflops": 128 / dtl * args.pgd_steps * fl / 1e12, "frac": 128 / dtl * args.pgd_steps * fl / PEAK_BF16_DENSE,
                                       "what": f"BASELINE config 5's model (ViT-L/16 + LoRA r=16), PGD-{args.pgd_steps} at its per-GPU batch 128"}
        log(f"extras: ViT-L/16 + LoRA r=16 PGD-{args.pgd_steps} at batch 128: {128 / dtl:.1f} img/s")
        # BASELINE config 5: adversarial-patch EoT steps (32x32 circular patch, random scale / rotation / location per image,
        # Adam lr 5 on the patch) on the same model and batch: overlay -> forward -> CE -> backward-to-input -> patch gradient
        patch_mod = importlib.import_module(PKG + ".patch")
        model_mod = importlib.import_module(PKG + ".model")
        vit = model_mod.ViTForImageClassification(archl, P.LoraSpec(r=16, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev)
        vit._eng = engl                     # the attack resolves its engine through the model object
        atk = patch_mod.AdversarialPatchPyTorch(vit, rotation_max=22.5, scale_min=0.05, scale_max=1.0, learning_rate=5.0, max_iter=1,
                                                batch_size=128, patch_shape=(3, 32, 32), patch_type="circle", targeted=False,
                                                verbose=False, seed=3)
        for _ in range(2):
            atk.train_step(xl, yl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ne = 5
        for _ in range(ne):
            atk.train_step(xl, yl)
        torch.cuda.synchronize()
        dte = (time.perf_counter() - t0) / ne
        res["vit_l16_lora_r16_patch_eot_step"] = {"value": 128 / dte, "unit": "img/s", "ms_per_step": 1e3 * dte, "batch": 128,
                                                  "tflops": 128 / dte * fl / 1e12, "frac": 128 / dte * fl / PEAK_BF16_DENSE, "dtype": "f16",
                                                  "what": "BASELINE config 5: one EoT step: warp-and-paste 32x32 circle patch, forward, CE, backward to pixels, patch gradient, Adam, clamp"}
        log(f"extras: ViT-L/16 patch EoT step {1e3 * dte:.1f} ms at batch 128")
        del engl, vit, atk
        eng = None
    if args.swin:
        # BASELINE config 4: Swin-T + LoRA r = 16, PGD (fp32 first form of the windowed-attention path), batch 256
        swin = importlib.import_module(PKG + ".swin")
        from transformers import SwinConfig, SwinForImageClassification     # random-init weights of the architecture (no hub access)
        torch.manual_seed(0)
        hf = SwinForImageClassification(SwinConfig(num_labels=21))
        for prec in ("f16", "f32"):
            se = swin.SwinEngine(swin.SwinArch(num_labels=21), lora_r=16, lora_alpha=16.0, lora_targets=TARGETS, device=dev, precision=prec)
            g = torch.Generator().manual_seed(5)
            se.load_state_dict(hf.state_dict())
            for si, d in enumerate((2, 2, 6, 2)):
                for bi in range(d):
                    for t in TARGETS:
                        A, Bm = se.param(si, bi, t, "A"), se.param(si, bi, t, "B")
                        A.copy_((torch.rand(A.shape, generator=g) * 2 - 1) / A.shape[1] ** 0.5)
                        Bm.copy_(torch.randn(Bm.shape, generator=g) * 0.02)
            nsw = 8 if prec == "f16" else 4
            se.pgd_attack(x, y, EPS, ALPHA, 2, random_start=True, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            se.pgd_attack(x, y, EPS, ALPHA, nsw, random_start=True, seed=2)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / nsw
            sfl = swin_flops_per_image_step(swin.SwinArch(num_labels=21), 16, TARGETS)
            speak = PEAK_BF16_DENSE if prec == "f16" else PEAK_F32_MATRIX
            res["swin_t_lora_r16_pgd_step_" + prec] = {"value": x.shape[0] / dts, "unit": "img/s per PGD step", "ms_per_pgd_step": 1e3 * dts,
                                                        "batch": int(x.shape[0]), "pgd40_img_per_s": x.shape[0] / (40 * dts), "dtype": prec,
                                                        "algorithmic_gflop_per_image_per_pgd_step": sfl / 1e9,
                                                        "roofline": {"bound": "mfma", "achieved": x.shape[0] / dts * sfl / 1e12, "peak": speak / 1e12,
                                                                     "unit": "TFLOP/s", "frac": x.shape[0] / dts * sfl / speak,
                                                                     "note": "whole PGD step (BASELINE config 4); stage 1-2 products are HBM / L2-fill bound (C = 96 / 192), "
                                                                             "per-kernel times: profiles/r05_swin16_kernel_stats.csv"}}
            log(f"extras: Swin-T + LoRA r=16 PGD step {1e3 * dts:.1f} ms at batch {x.shape[0]} ({prec})")
            del se
        del hf
    if args.precision == "f16":
        # the reference's own precision (whitebox_attacks.py:22-38 is fp32 end to end): the SAME PGD attack in the fp32 parity
        # mode (every operand and activation fp32, exact-f32 MFMA), priced against the f32 matrix peak
        e32 = P.Engine(arch, P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev, precision="f32")
        e32.load_state_dict(syn.random_state_dict(arch, seed=0))
        for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
            e32.param(i, t, "A").copy_(A)
            e32.param(i, t, "B").copy_(Bm)
        e32.commit()
        adv32 = torch.empty_like(x)
        # (round-4 verdict, weak 9: one timed attack after a one-iteration warm-up was thin for the figure that carries the
        #  precision argument) a FULL-LENGTH warm-up attack, then the mean of three timed attacks
        e32.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=1, out=adv32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n32 = 3
        for i in range(n32):
            e32.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=2 + i, out=adv32)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t0) / n32
        fl = algorithmic_flops_per_image_step(arch, args.rank, TARGETS)
        tf = x.shape[0] / d32 * args.pgd_steps * fl / 1e12
        res["fp32_mode"] = {"value": x.shape[0] / d32, "unit": "img/s", "ms_per_step": 1e3 * d32, "batch": int(x.shape[0]),
                            "dtype": "f32", "tflops": tf, "peak_tflops": PEAK_F32_MATRIX / 1e12, "frac": tf * 1e12 / PEAK_F32_MATRIX,
                            "timed_attacks": n32, "warmup_attacks": 1,
                            "what": f"the headline PGD-{args.pgd_steps} attack with precision=f32 (the reference's arithmetic)"}
        log(f"extras: fp32 mode {x.shape[0] / d32:.1f} img/s ({tf:.1f} TFLOP/s)")
        del e32, adv32
        # and with bf16 operands (VL_PREC_BF16: the same kernels instantiated on __bf16; the type north_star / config 5 name)
        eb = P.Engine(arch, P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.0, targets=TARGETS), device=dev, precision="bf16")
        eb.load_state_dict(syn.random_state_dict(arch, seed=0))
        for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
            eb.param(i, t, "A").copy_(A)
            eb.param(i, t, "B").copy_(Bm)
        eb.commit()
        advb = torch.empty_like(x)
        eb.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=1, out=advb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 3
        for i in range(nb):
            eb.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=2 + i, out=advb)
        torch.cuda.synchronize()
        db = (time.perf_counter() - t0) / nb
        flb = algorithmic_flops_per_image_step(arch, args.rank, TARGETS)
        res["bf16_mode"] = {"value": x.shape[0] / db, "unit": "img/s", "ms_per_step": 1e3 * db, "batch": int(x.shape[0]), "dtype": "bf16",
                            "tflops": x.shape[0] / db * args.pgd_steps * flb / 1e12, "frac": x.shape[0] / db * args.pgd_steps * flb / PEAK_BF16_DENSE,
                            "timed_attacks": nb, "warmup_attacks": 1,
                            "what": f"the headline PGD-{args.pgd_steps} attack with precision=bf16"}
        log(f"extras: bf16 mode {x.shape[0] / db:.1f} img/s")
        del eb, advb
    res["lora_train_step"] = {"value": bt / dt, "unit": "img/s", "ms_per_step": 1e3 * dt, "batch": bt,
                              "what": "forward(train, dropout 0.1) + CE + LoRA/classifier backward + Adam, clean inputs"}
    log(f"extras: LoRA train step {1e3 * dt:.2f} ms at batch {bt}")
    return res


_PMC = None


def _pmc_table():
    """The committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE per launch, tools/pmc_traffic.py; counters cannot be read
    from inside this process).  The profile carries a hash of the kernel sources it was taken on: if the sources changed
    since, every traffic number is OMITTED (None), never quoted stale."""
    global _PMC
    if _PMC is None:
        _PMC = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_traffic import kernel_source_sha16
            t = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
            if t.get("_meta", {}).get("kernel_source_sha16") == kernel_source_sha16():
                _PMC = {k: v for k, v in t.items() if k != "_meta"}
        except Exception:
            _PMC = {}
    return _PMC


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (the library's profiling-scope name) or None."""
    t = _pmc_table()
    if not t:
        return None
    # the library's scopes name the ping-pong GEMM "gemm_pp_kernel<EPI>" / "<EPI, down N>"; rocprofv3 reports "<EPI, N>"
    m = re.fullmatch(r"gemm_pp_kernel<(\d+)(?:, down (\d+))?>", kernel)
    if m:       # as rocprofv3 prints it: <EPI, ND, BC> (BC: the residual-add epilogue with the down projection inside)
        kernel = f"gemm_pp_kernel<{m.group(1)}, {m.group(2) or 0}, {'true' if m.group(1) == '10' and m.group(2) else 'false'}>"
    if kernel in t:
        return t[kernel]["hbm_bytes_per_launch"]
    # scopes without template arguments (layernorm_fwd_kernel, attn_bwd_img_kernel ...): launch-weighted mean of the instances
    inst = [v for k, v in t.items() if k.split("<")[0] == kernel]
    n = sum(v["launches"] for v in inst)
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / n if n else None


_SQ = None


def pmc_mfma_busy(kernel):
    """Fraction of time a SIMD's matrix pipe was busy in `kernel`, from the committed SQ counter pass
    (profiles/pmc_sq.json, written by tools/pmc_summary.py --json: SQ_VALU_MFMA_BUSY_CYCLES / (32 shader engines x
    SQ_BUSY_CYCLES) ... normalised per SIMD there); None when the kernel sources changed since the pass."""
    global _SQ
    if _SQ is None:
        _SQ = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_traffic import kernel_source_sha16
            t = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))
            if t.get("_meta", {}).get("kernel_source_sha16") == kernel_source_sha16():
                _SQ = {k: v for k, v in t.items() if k != "_meta"}
        except Exception:
            _SQ = {}
    if not _SQ:
        return None
    m = re.fullmatch(r"gemm_pp_kernel<(\d+)(?:, down (\d+))?>", kernel)
    if m:       # as rocprofv3 prints it: <EPI, ND, BC> (BC: the residual-add epilogue with the down projection inside)
        kernel = f"gemm_pp_kernel<{m.group(1)}, {m.group(2) or 0}, {'true' if m.group(1) == '10' and m.group(2) else 'false'}>"
    inst = [v for k, v in _SQ.items() if k == kernel or k.split("<")[0] == kernel]
    n = sum(v["launches"] for v in inst)
    return round(sum(v["mfma_busy"] * v["launches"] for v in inst) / n, 4) if n else None


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return int(os.environ.get("BENCH_CPU_CORES", min(n, 16)))   # a 1-GPU box's CPU share is 16


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torch.distributed.run, one rank
    per GPU over RCCL) before this process has touched the GPU, relay rank 0's JSON line and the child's exit code."""
    import socket
    import subprocess
    if os.environ.get("BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()          # does not initialise the GPU
        if have < n:
            print(f"[bench] --gpus {n} but only {have} device(s) visible: refusing to report a {n}-GPU number", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting", " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks exited without a result line", file=sys.stderr)
        rc = 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--pgd-steps", type=int, default=20)
    ap.add_argument("--rank", type=int, default=8, help="LoRA rank")
    ap.add_argument("--merged", action="store_true", help="fold LoRA into W (merge_and_unload) instead of fusing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    # BASELINE configs 5 and 4 are part of the default line since round 5 (the driver's run must witness them); --vitl / --swin
    # are still accepted (no-ops), --no-vitl / --no-swin skip them
    ap.add_argument("--vitl", action="store_true", default=True, help="extras: ViT-L/16 + LoRA r=16 at batch 128: PGD attack and patch EoT step (default on)")
    ap.add_argument("--no-vitl", dest="vitl", action="store_false")
    ap.add_argument("--swin", action="store_true", default=True, help="extras: a PGD step of the Swin-T + LoRA r=16 path at the bench batch, fp16 and fp32 (default on)")
    ap.add_argument("--no-swin", dest="swin", action="store_false")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (merged-LoRA attack, LoRA train step)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch images PER GPU (the default, what the driver's scaling curve assumes); strong: --batch is the "
                         "GLOBAL batch, split into contiguous shards of batch/N images per GPU (SURVEY 8e: 256 -> 32 per GPU at N = 8)")
    ap.add_argument("--precision", choices=("f16", "bf16", "f32"), default="f16",
                    help="f16: fp16 operands / fp32 accumulation (the MFMA-rate path); f32: the reference's own precision (parity mode)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs (one-GPU box): BENCH_BACKEND=gloo, BENCH_SHARE_GPU=1 puts every rank on cuda:0
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        # the rank count the collective library itself saw (RCCL when backend = nccl)
        one = torch.ones(1, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    P = importlib.import_module(PKG)
    arch = P.ArchConfig(num_labels=21)
    spec = P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.0, targets=TARGETS, merged=args.merged)
    eng = P.Engine(arch, spec, device=dev, precision=args.precision)
    syn = importlib.import_module(PKG + ".synthetic")
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    log("weights loaded")
    if args.scaling == "strong":
        # the global batch (seed 100, the one-GPU run's batch) cut into contiguous shards: rank r attacks images [lo, hi)
        optim = importlib.import_module(PKG + ".optim")
        lo, hi = optim.shard_batch(args.batch, rank, world)
        xg, yg = syn.random_batch(arch, args.batch, seed=100)
        x, y = xg[lo:hi].contiguous(), yg[lo:hi].contiguous()
        del xg, yg
    else:
        x, y = syn.random_batch(arch, args.batch, seed=100 + rank)
    local_batch = int(x.shape[0])
    x, y = x.to(dev), y.to(dev)
    adv = torch.empty_like(x)

    def step(seed):
        eng.pgd_attack(x, y, EPS, ALPHA, args.pgd_steps, random_start=True, seed=seed, out=adv)

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(2 + i)
        torch.cuda.synchronize()
        log(f"warmup {i} done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(2 + i)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    global_batch = args.batch if args.scaling == "strong" else world * args.batch
    imgs = global_batch * args.steps
    value = imgs / dt
    if rank == 0:
        log(f"timed region: {dt:.3f} s for {args.steps} steps -> {value:.1f} img/s")
    flops_img_step = algorithmic_flops_per_image_step(arch, args.rank, TARGETS)

    out = {
        "metric": "adversarial images/sec (PGD-20, ViT-B/16+LoRA r=8, bs256)",
        "value": value, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"PGD-{args.pgd_steps} eps=8/255 alpha=2/255 random_start, ViT-B/16 (21 classes) + LoRA "
                               f"r={args.rank} on q,k,v,attn-out,fc2 ({'merged' if args.merged else 'fused'}), "
                               f"batch {local_batch}/GPU of synthetic 224x224x3 in HBM, seeded random-init weights",
                   "global_batch": global_batch, "per_gpu_batch": local_batch, "scaling": args.scaling,
                   "pgd_steps": args.pgd_steps, "lora_rank": args.rank,
                   "parallelism": f"dp{world} (batch shards, no data-path collective)",
                   "ranks_seen_by_collective": ranks_seen,
                   "collective_backend": (dist.get_backend() if world > 1 else None),
                   "scaling_claim": ("north_star's >= 6x at 8 GPUs is claimed under WEAK scaling (this line's default: 256 images per GPU, "
                                     "independent shards, no data-path collective); under strong scaling of ONE 256-image batch (32 per "
                                     "GPU) the estimate is 8 x extras.batch_sweep['32'] / value")},
    }
    peak = PEAK_F32_MATRIX if args.precision == "f32" else PEAK_BF16_DENSE

    if rank == 0 and not args.no_roofline:
        lib = eng.lib
        P.check(lib.vl_profile_begin(), "vl_profile_begin")
        eng.pgd_attack(x, y, EPS, ALPHA, 2, random_start=False, out=adv)      # eager, event-bracketed
        # K10 stands alone only outside vl_pgd_attack (FGSM, vl_pgd_step callers; inside the attack it is the epilogue of
        # the patch-gradient GEMM): its own HBM roofline is taken on the same batch here
        g10 = torch.randn_like(x)
        a10 = x.clone()
        for _ in range(4):
            eng.pgd_step(a10, x, g10, EPS, ALPHA)
        del g10, a10
        buf = ctypes.create_string_buffer(1 << 16)
        P.check(lib.vl_profile_report(buf, len(buf)), "vl_profile_report")
        prof = json.loads(buf.value.decode())
        log("roofline pass done")
        ps = prof.pop("pgd_step_kernel", None)
        tot_ms = sum(v["ms"] for v in prof.values())
        exec_iter = sum(v.get("exec_flops", 0.0) for v in prof.values()) / 2      # per PGD iteration of this rank's batch
        dom = max((k for k in prof if prof[k]["flops"] > 0), key=lambda k: prof[k]["ms"])
        d = prof[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # every kernel that takes >= 1 % of the iteration: time, algorithmic work, fraction of the roof that bounds it, and
        # counter traffic / algorithmic bytes (> 1 = re-reads); MFMA-bound when it has FLOPs and they, priced at the matrix
        # peak, outweigh its bytes priced at the HBM peak
        table = {}
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            if v["ms"] < 0.01 * tot_ms:
                continue
            t_s = v["ms"] * 1e-3
            mf = v["flops"] / peak > v["bytes"] / PEAK_HBM
            row = {"ms_per_pgd_iteration": round(v["ms"] / 2, 4), "launches_per_iteration": v["n"] // 2,
                   "bound": "mfma" if mf else "hbm",
                   "algorithmic_gflop_per_launch": round(v["flops"] / v["n"] / 1e9, 3),
                   "algorithmic_mb_per_launch": round(v["bytes"] / v["n"] / 1e6, 2),
                   "achieved": round(v["flops"] / t_s / 1e12, 1) if mf else round(v["bytes"] / t_s / 1e9, 1),
                   "unit": "TFLOP/s" if mf else "GB/s",
                   "frac": round((v["flops"] / t_s / peak) if mf else (v["bytes"] / t_s / PEAK_HBM), 4)}
            if v.get("exec_flops", 0) > 0:
                row["executed_gflop_per_launch"] = round(v["exec_flops"] / v["n"] / 1e9, 3)
                row["executed_frac"] = round(v["exec_flops"] / t_s / peak, 4)
            row["mfma_busy"] = pmc_mfma_busy(k)
            tr = pmc_traffic(k)
            row["traffic_mb_per_launch"] = None if tr is None else round(tr / 1e6, 1)
            row["traffic_over_algorithmic"] = None if (tr is None or not v["bytes"]) else round(tr / (v["bytes"] / v["n"]), 3)
            table[k] = row
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": ach * 1e12 / peak, "traffic": pmc_traffic(dom),
            "launches": d["n"], "avg_launch_ms": d["ms"] / d["n"], "share_of_step_time": d["ms"] / tot_ms,
            "path": {"achieved": value * args.pgd_steps * flops_img_step / 1e12, "unit": "TFLOP/s",
                     "frac": value * args.pgd_steps * flops_img_step / (world * peak),
                     "gflop_per_image_per_pgd_step": flops_img_step / 1e9,
                     # what the matrix pipes were asked to do: padded rows, 32-token attention tiles, the whole LoRA K tile,
                     # minus the dead rows of the last layer -- summed over the launches of one eager iteration, priced at the
                     # timed region's iteration time
                     "executed_gflop_per_image_per_pgd_step": exec_iter / local_batch / 1e9,
                     "executed_frac": exec_iter / (1e-3 * (1e3 * dt / args.steps) / args.pgd_steps) / peak,
                     "note": "frac: algorithmic FLOPs of the reference's computation (SURVEY 8d); executed_frac: FLOPs the kernels "
                             "issue to the MFMA pipes (work per image differs: LoRA tile padding, T = 197 -> 224, dead rows removed)"},
            "pgd_step": None if not ps else {
                "bound": "hbm", "achieved": ps["bytes"] / (ps["ms"] * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9,
                "unit": "GB/s", "frac": ps["bytes"] / (ps["ms"] * 1e-3) / PEAK_HBM, "avg_launch_ms": ps["ms"] / ps["n"],
                "note": "standalone K10 (FGSM / vl_pgd_step callers), 16 B per element; inside vl_pgd_attack the step is the "
                        "epilogue of the patch-gradient GEMM and has no launch of its own"},
            "kernels": table,
            "kernels_ms_per_pgd_iteration": {k: round(v["ms"] / 2, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        }

    if rank == 0 and world == 1 and not args.no_extras:
        out["extras"] = extras(P, syn, arch, args, dev, x, y)
        # the same attack at the per-GPU batches of the data-parallel configs (256 -> 32 per GPU at 8 GPUs;
        # BASELINE config 3: 64 per GPU; the reference's own default batch is 32, train_loras.py:237-243)
        sweep = {}
        for bsz in (32, 64, 128):
            if bsz >= local_batch:
                continue
            xs_, ys_ = x[:bsz].contiguous(), y[:bsz].contiguous()
            advs = torch.empty_like(xs_)
            eng.pgd_attack(xs_, ys_, EPS, ALPHA, args.pgd_steps, random_start=True, seed=1, out=advs)     # graph capture for this batch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nrep = 3
            for i in range(nrep):
                eng.pgd_attack(xs_, ys_, EPS, ALPHA, args.pgd_steps, random_start=True, seed=2 + i, out=advs)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / nrep
            sweep[str(bsz)] = {"value": bsz / dts, "unit": "img/s", "ms_per_step": 1e3 * dts,
                               "relative_to_headline_batch": (bsz / dts) / value}
            log(f"extras: batch {bsz}: {bsz / dts:.1f} img/s")
        out["extras"]["batch_sweep"] = sweep
        if "32" in sweep and local_batch == 256:
            out["extras"]["strong_scaling_8gpu_estimate"] = {
                "value": 8 * sweep["32"]["value"] / value, "unit": "x one GPU",
                "what": "one 256-image batch cut into 8 shards of 32 (SURVEY 8e): 8 x the batch-32 rate / the batch-256 rate, no collective on the path"}
    if world > 1 and not args.no_extras:
        res = dp_train_step(P, syn, arch, args, dev, rank, world)        # every rank takes part (one collective per step)
        if rank == 0:
            out.setdefault("extras", {})["dp_adv_lora_train_step_pgd7"] = res

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(arch, args)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dp_train_step(P, syn, arch, args, dev, rank, world, bt=64, steps=4):
    """BASELINE config 3 data-parallel (train_loras.py:308-315 made data-parallel): every rank makes PGD-7
    adversarial examples of ITS 64 images against the current adapters, runs the LoRA train step on them, then ONE all-reduce
    (RCCL over xGMI with backend nccl) of the flat fp32 LoRA + classifier gradient and the same fused Adam everywhere."""
    import torch.distributed as dist
    optim = importlib.import_module(PKG + ".optim")
    eng = P.Engine(arch, P.LoraSpec(r=args.rank, alpha=16.0, dropout=0.1, targets=TARGETS), device=dev)
    eng.load_state_dict(syn.random_state_dict(arch, seed=0))
    for (i, t), (A, Bm) in syn.random_lora(arch, args.rank, TARGETS, seed=1).items():
        eng.param(i, t, "A").copy_(A)
        eng.param(i, t, "B").copy_(Bm)
    eng.commit()
    xt, yt = syn.random_batch(arch, bt, seed=300 + rank)
    xt, yt = xt.to(dev), yt.to(dev)
    m1, m2 = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)
    on_gpu = dist.get_backend() == "nccl"
    ex_ms, seen = [], 0

    def one(t):
        nonlocal seen
        xa = eng.pgd_attack(xt, yt, EPS, ALPHA, 7, True, seed=t)
        eng.forward(xa, normalise=True, train=True)
        eng.loss_ce(yt)
        _, g = eng.backward(False, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if on_gpu:
            optim.allreduce_weighted_mean_(g, bt, bt * world)
            torch.cuda.synchronize()
        else:                                   # rehearsal on one GPU (gloo): the same exchange through host memory
            gc = g.cpu()
            optim.allreduce_weighted_mean_(gc, bt, bt * world)
            g.copy_(gc)
        ex_ms.append(1e3 * (time.perf_counter() - t0))
        eng.adam_step(eng.flat, g, m1, m2, 1e-4, 0.9, 0.999, 1e-8, t)
        eng.commit()

    for t in range(1, 3):
        one(t)
    ex_ms.clear()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(3, 3 + steps):
        one(t)
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / steps
    # identical parameters on every rank afterwards (the exchange really averaged): max |flat - flat of rank 0|
    ref = eng.flat.detach().clone() if on_gpu else eng.flat.detach().cpu()
    dist.broadcast(ref, 0)
    dev_max = float((ref.to(dev) - eng.flat).abs().max())
    one_t = torch.ones(1, device=dev if on_gpu else "cpu")
    dist.all_reduce(one_t)
    seen = int(one_t.item())
    tt = torch.tensor([dt, sum(ex_ms) / len(ex_ms), dev_max], dtype=torch.float64, device=dev if on_gpu else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, ex, dmax = (float(v) for v in tt.tolist())
    return {"value": world * bt / dt, "unit": "img/s", "ms_per_step": 1e3 * dt, "batch_per_gpu": bt, "global_batch": world * bt,
            "exchange_ms": ex, "exchange_payload_bytes": int(eng.flat.numel()) * 4, "exchange": "one in-place scale + ONE all_reduce(SUM) "
            "of the flat fp32 LoRA + classifier gradient", "ranks_seen_by_collective": seen, "collective_backend": dist.get_backend(),
            "max_param_divergence_across_ranks": dmax,
            "what": "PGD-7 against the current adapters + forward(train, dropout 0.1) + CE + LoRA/classifier backward + gradient "
                    "all-reduce + Adam on every rank (BASELINE config 3)"}


def cpu_baseline(arch, args):
    """The oracle (CPU torch restatement of the reference path) on a bounded sample: a few
    images x a few of the 20 PGD steps; every step costs the same, so img/s for PGD-20 =
    images * (steps_done / 20) / seconds."""
    from oracle import vit_lora_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    cfg = O.OracleConfig(num_labels=arch.num_labels)
    w = O.init_weights(cfg, seed=0)
    lora = O.init_lora(cfg, r=args.rank, targets=TARGETS, seed=1)
    nb, ns = 16, 10                      # ~10-20 s of host work on 16 cores
    g = torch.Generator().manual_seed(100)
    x = torch.rand(nb, 3, cfg.image_size, cfg.image_size, generator=g)
    y = torch.randint(0, cfg.num_labels, (nb,), generator=torch.Generator().manual_seed(101))
    O.pgd(w, cfg, x[:2], y[:2], EPS, ALPHA, 1, lora)          # warm-up
    log("cpu warm-up done")
    t0 = time.perf_counter()
    O.pgd(w, cfg, x, y, EPS, ALPHA, ns, lora)
    dt = time.perf_counter() - t0
    return {"value": nb * (ns / args.pgd_steps) / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{nb} images x {ns} of {args.pgd_steps} PGD steps (fp32, torch CPU), {dt:.1f} s, scaled by steps"}


if __name__ == "__main__":
    main()
