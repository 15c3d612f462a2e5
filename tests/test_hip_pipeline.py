"""The reference's on-disk pipeline end to end on the GPU, with its own file formats at every hand-over:

    train.py            -> <models>/<model>/<source>/<model>_best_model_finetuned.pth + class_mappings.txt
    whitebox_attacks.py -> <adv>/<model>/<source>/<split>/<attack>/images/*.png + metadata.csv   (save_images bytes)
    train_loras.py      -> <loras>/<model>/<source>/<attack>/rank{r}_best_adapter, rank{r}_final_adapter, results.json
    eval_compose.py     -> finds those adapters (no --synthetic), evaluates base / single / merged models

on a tiny architecture and a synthetic dataset tree in the reference's layout (Utils.py:12-82)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from helpers import PKG, pkg

pytestmark = pytest.mark.gpu


def free_port() -> str:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])

CLASSES = ["stop", "yield", "speed_limit", "no_entry"]


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = tmp_path_factory.mktemp("pipeline")
    syn = importlib.import_module(PKG + ".synthetic")
    data = str(root / "data")
    syn.write_dataset_tree(data, CLASSES, {"train": 40, "val": 12, "test": 12}, image_size=64, seed=3)
    return {"root": str(root), "data": data, "models": str(root / "models"), "adv": str(root / "adv"),
            "loras": str(root / "loras")}


def test_pipeline_train_attack_lora_compose(tree):
    import eval_compose
    import train
    import train_loras
    import whitebox_attacks
    from PIL import Image
    P = pkg()
    train.main(["--data_root", tree["data"], "--output_dir", tree["models"], "--synthetic", "--arch", "tiny", "--source", "mapillary"])
    ckpt = os.path.join(tree["models"], "google_vit", "mapillary", "google_vit_best_model_finetuned.pth")
    mapping = os.path.join(tree["models"], "google_vit", "mapillary", "class_mappings.txt")
    assert os.path.exists(ckpt) and open(mapping).read().splitlines() == [f"{i}: {c}" for i, c in enumerate(sorted(CLASSES))]

    eps = 8 / 255
    whitebox_attacks.main(["--data_root", tree["data"], "--models", "google_vit", "--sources", "mapillary", "--model_base_path",
                           tree["models"], "--output_dir", tree["adv"], "--arch", "tiny", "--batch_size", "16", "--pgd_iters", "3",
                           "--epsilon", str(eps)])
    import pandas as pd
    for split, n in (("train", 40), ("val", 12), ("test", 12)):
        for attack in ("fgsm", "pgd"):
            d = os.path.join(tree["adv"], "google_vit", "mapillary", split, attack)
            meta = pd.read_csv(os.path.join(d, "metadata.csv"))
            assert len(meta) == n and len(os.listdir(os.path.join(d, "images"))) == n
            assert all(os.path.exists(p) for p in meta["image_path"])
            assert list(meta.columns) == ["image_path", "unified_class", "source"]
    # FGSM PNG = clean PNG +- eps (quantised with save_images' truncation): at most ceil(eps*255) grey levels apart
    clean = np.asarray(Image.open(os.path.join(tree["data"], "test", "images", "test_00003.png")), dtype=np.int32)
    adv = np.asarray(Image.open(os.path.join(tree["adv"], "google_vit", "mapillary", "test", "fgsm", "images", "test_00003.png")), dtype=np.int32)
    # (the 64-pixel tiny arch resizes 64 -> 73 -> centre crop 64, so compare statistics, not pixels)
    assert adv.shape == (64, 64, 3) and 1 <= np.abs(adv - adv.mean()).max()

    base = os.path.join(tree["models"], "{model}", "{source}", "{model}_best_model_finetuned.pth")
    res = train_loras.main(["--models", "google_vit", "--sources", "mapillary", "--attacks", "fgsm", "pgd", "--model_base_path", base,
                            "--adv_root", tree["adv"], "--data_root", tree["data"], "--output_dir", tree["loras"], "--ranks", "4",
                            "--epochs", "2", "--arch", "tiny", "--batch_size", "16", "--lr", "1e-3"])
    for attack in ("fgsm", "pgd"):
        d = os.path.join(tree["loras"], "google_vit", "mapillary", attack)
        for kind in ("best", "final"):                      # the reference's directory names (train_loras.py:343,353)
            a = os.path.join(d, f"rank4_{kind}_adapter")
            assert sorted(os.listdir(a)) == ["adapter_config.json", "adapter_model.safetensors"]
        r = json.load(open(os.path.join(d, "results.json")))["4"]
        assert set(r) == {"train_loss", "train_acc", "val_loss", "val_acc", "val_f1", "clean_test_acc", "clean_test_f1",
                          "adv_test_acc", "adv_test_f1", "best_val_acc"}
        assert len(r["train_loss"]) == 2 and len(r["val_acc"]) == 2 and r["best_val_acc"] == max(r["val_acc"])
        assert all(np.isfinite(v) for v in r["train_loss"])
    g = json.load(open(os.path.join(tree["loras"], "global_results.json")))
    assert set(g["google_vit"]["mapillary"]) == {"fgsm", "pgd"} and res["google_vit"]["mapillary"]["fgsm"][4]["train_loss"]

    out = os.path.join(tree["root"], "compose.json")
    r = eval_compose.main(["--model_path", ckpt, "--lora_root", tree["loras"], "--adv_root", tree["adv"], "--data_root", tree["data"],
                           "--attacks", "fgsm", "pgd", "--rank", "4", "--arch", "tiny", "--output_file", out])
    assert {"base_model", "fgsm_lora", "pgd_lora", "fgsm+pgd_merged"} <= set(r)
    assert set(r["base_model"]) == {"clean", "fgsm", "pgd"}
    assert all(0.0 <= v["accuracy"] <= 1.0 for v in r["fgsm+pgd_merged"].values())
    assert json.load(open(out))["rank"] == 4


def test_train_loras_synthetic_with_pgd_inner_loop(tmp_path):
    """BASELINE config 3 through the CLI: PGD-K against the CURRENT adapters inside every train step."""
    import train_loras
    res = train_loras.main(["--output_dir", str(tmp_path), "--attacks", "pgd", "--ranks", "4", "--epochs", "2", "--synthetic", "24",
                            "--num_classes", "5", "--arch", "tiny", "--batch_size", "8", "--pgd-inner-steps", "2", "--lr", "1e-3",
                            "--lora_dropout", "0.1"])
    r = res["google_vit"]["mapillary"]["pgd"][4]
    assert len(r["train_loss"]) == 2 and all(np.isfinite(v) for v in r["train_loss"] + r["val_loss"])
    assert os.path.isdir(os.path.join(str(tmp_path), "google_vit", "mapillary", "pgd", "rank4_best_adapter"))


def test_two_rank_attack_generation_equals_single_process(tmp_path):
    """whitebox_attacks.py under `torch.distributed.run` with two ranks (both on cuda:0: VITLORA_SHARE_GPU=1, the only GPU of
    the box; coordination over gloo as in production): interleaved shards, no data-path collective, barrier + file-name
    exchange before rank 0 reports.  Every PNG must be byte-identical to the single-process run's: FGSM of an image does not
    depend on which rank or batch it was in, as long as the batches have the same SIZE (the mean loss puts 1/B into the
    gradient before the sign; 32 images in batches of 8 on both sides -- a ragged split would round 1/5 and 1/3 differently
    and may flip the sign of a near-zero gradient entry, in the reference just the same)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--models", "google_vit", "--sources", "mapillary", "--synthetic", "32", "--arch", "tiny", "--attacks", "fgsm",
              "--splits", "test", "--batch_size", "8", "--num_classes", "5"]
    env = dict(os.environ, VITLORA_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r1 = subprocess.run([sys.executable, os.path.join(root, "whitebox_attacks.py"), "--output_dir", one] + common, env=env,
                        capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", free_port(), os.path.join(root, "whitebox_attacks.py"), "--output_dir", two] + common,
                        env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d1 = os.path.join(one, "google_vit", "mapillary", "test", "fgsm", "images")
    d2 = os.path.join(two, "google_vit", "mapillary", "test", "fgsm", "images")
    names = sorted(os.listdir(d1))
    assert len(names) == 32 and sorted(os.listdir(d2)) == names
    for n in names:
        assert open(os.path.join(d1, n), "rb").read() == open(os.path.join(d2, n), "rb").read(), n
    assert "32 images per attack" in r2.stdout


def test_two_rank_lora_training_equals_single_process(tmp_path):
    """train_loras.py --synthetic under `torch.distributed.run` with two ranks (one GPU shared, gloo instead of RCCL: the
    collectives are the same calls) against one process: same number of optimizer steps on both ranks although 40 samples in
    global batches of 16 leave a ragged last batch, ONE weighted all-reduce per step, initial adapters broadcast from rank 0.
    The gradient of a global batch is the size-weighted mean of the shard gradients, so the trained adapters agree up to
    summation order (fp32 mode, dropout 0: the LoRA dropout mask is indexed inside the LOCAL batch)."""
    import subprocess
    import sys
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--synthetic", "40", "--arch", "tiny", "--attacks", "fgsm", "--ranks", "4", "--epochs", "2", "--batch_size", "16",
              "--lora_dropout", "0", "--precision", "f32", "--num_classes", "5", "--lr", "1e-3"]
    env = dict(os.environ, VITLORA_SHARE_GPU="1", VITLORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    outs = []
    for tag, launcher in (("one", [sys.executable]),
                          ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", free_port()])):
        out = str(tmp_path / tag)
        r = subprocess.run(launcher + [os.path.join(root, "train_loras.py"), "--output_dir", out] + common, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stdout[-1500:], r.stderr[-1500:])
        d = os.path.join(out, "google_vit", "mapillary", "fgsm")
        res = json.load(open(os.path.join(d, "results.json")))
        outs.append((load_file(os.path.join(d, "rank4_final_adapter", "adapter_model.safetensors")), res))
    (a1, r1), (a2, r2) = outs
    assert a1.keys() == a2.keys() and len(a1) > 0
    moved = 0.0
    for k in a1:
        x, y = a1[k].double(), a2[k].double()
        assert (x - y).norm() <= 1e-3 * max(x.norm().item(), 1e-3), k
        moved += x.norm().item()
    assert moved > 0.0
    k4 = next(iter(r1))                                    # results.json is keyed by the rank
    for key, tol in (("train_loss", 1e-4), ("val_loss", 1e-4), ("train_acc", 1e-6), ("val_acc", 1e-6)):       # one entry per epoch
        assert len(r1[k4][key]) == 2 and np.allclose(r1[k4][key], r2[k4][key], atol=tol), (key, r1[k4][key], r2[k4][key])


_FAIL_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import train_loras as TL
rank = int(os.environ["RANK"])
orig = TL.opt.Adam.step
calls = [0]
def step(self, *a, **k):
    calls[0] += 1
    if rank == 1 and calls[0] == 2:
        raise RuntimeError("injected failure on rank 1, before its all-reduce")
    return orig(self, *a, **k)
TL.opt.Adam.step = step
TL.main(["--output_dir", {out!r}, "--synthetic", "40", "--arch", "tiny", "--attacks", "fgsm", "--ranks", "4", "--epochs", "2",
         "--batch_size", "16", "--lora_dropout", "0", "--num_classes", "5"])
'''


def test_one_sided_failure_ends_the_job_instead_of_desynchronising_it(tmp_path):
    """Round-2 ADVICE: rank 1 raises in the middle of the step loop while rank 0 is inside (or on its way to) the gradient
    all-reduce.  The failing rank must not enter any collective from its failure path (the old all_ok() all-reduce would have
    paired up with rank 0's gradient exchange); it exits non-zero at once and the job ends -- no hang, no success code."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fail_worker.py"
    script.write_text(_FAIL_WORKER.format(root=root, out=str(tmp_path / "out")))
    env = dict(os.environ, VITLORA_SHARE_GPU="1", VITLORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0, r.stdout[-1500:]
    assert "injected failure on rank 1" in (r.stdout + r.stderr)
    assert time.time() - t0 < 200
    assert not os.path.exists(os.path.join(str(tmp_path / "out"), "global_results.json"))


def test_two_rank_patch_attack_cli(tmp_path):
    """patch_attack.py --synthetic under `torch.distributed.run` with two ranks (one shared GPU, gloo): shards of every global
    batch, ONE shard-size-weighted all-reduce of the patch gradient per step, batches of the split patched round-robin, rank 0
    writes patch.npy.  With the deterministic "pgd" patch optimiser at a FIXED location / scale / zero rotation the
    transformations do not depend on the rank's random stream, so the two-rank patch equals the single-process patch up to
    the sign of near-zero gradient entries; every image of the split is written exactly once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--model", "google_vit", "--source", "mapillary", "--synthetic", "24", "--arch", "tiny", "--num_classes", "5", "--patch_type", "square", "--splits", "test", "--batch_size", "8",
              "--patch_sample_size", "24", "--patch_size", "16", "--max_iter", "3", "--optimizer", "pgd", "--learning_rate", "0.05",
              "--rotation_max", "0", "--scale_min", "0.3", "--scale_max", "0.3", "--patch_location_x", "8", "--patch_location_y", "8",
              "--precision", "f32"]
    env = dict(os.environ, VITLORA_SHARE_GPU="1", VITLORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    patches = []
    for tag, launcher in (("one", [sys.executable]),
                          ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                   "--master-addr", "127.0.0.1", "--master-port", free_port()])):
        out = str(tmp_path / tag)
        r = subprocess.run(launcher + [os.path.join(root, "patch_attack.py"), "--output_dir", out] + common, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stdout[-1500:], r.stderr[-1500:])
        base = os.path.join(out, "google_vit", "mapillary", "test", "patch_square")
        assert sorted(os.listdir(os.path.join(base, "images"))) == [f"test_{i:06d}.png" for i in range(24)]
        patches.append(np.load(os.path.join(base, "patch.npy")))
    p1, p2 = patches
    assert p1.shape == p2.shape == (3, 16, 16) and np.abs(p1 - 0.5).max() > 0.04          # three sign steps of 0.05 moved it
    assert (np.abs(p1 - p2) < 1e-6).mean() > 0.97


def test_bench_contract_one_and_two_ranks():
    """bench.py prints ONE JSON line with the driver's keys; under `torch.distributed.run` (two ranks sharing the GPU, gloo:
    BENCH_BACKEND / BENCH_SHARE_GPU) rank 0 alone prints it, n_gpus = 2 and the images of both ranks are counted."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--steps", "1", "--warmup", "1", "--batch", "32", "--pgd-steps", "2", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    lines = {}
    # n = 1; n = 2 under the driver's launcher; "self": plain `python bench.py --gpus 2`, which starts its own two ranks
    for tag, n, launcher in ((1, 1, [sys.executable]),
                             (2, 2, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                     "--master-addr", "127.0.0.1", "--master-port", free_port()]),
                             ("self", 2, [sys.executable])):
        e = dict(env)
        if tag == "self":
            e.pop("WORLD_SIZE", None)
        r = subprocess.run(launcher + [os.path.join(root, "bench.py"), "--gpus", str(n)] + small, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(js) == 1, r.stdout[-500:]
        lines[tag] = json.loads(js[0])
    # what north_star names at N > 1: strong scaling (the GLOBAL batch split into contiguous shards) and BASELINE
    # config 3's data-parallel step -- PGD-7 + LoRA train step on every rank WITH the flat-gradient all-reduce
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--batch", "32", "--pgd-steps", "2", "--no-cpu-baseline", "--no-roofline", "--scaling", "strong"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    dp = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert dp["scaling"] == "strong" and dp["config"]["global_batch"] == 32 and dp["config"]["per_gpu_batch"] == 16 and dp["n_gpus"] == 2
    assert abs(dp["value"] - 32 / (dp["ms_per_step"] / 1e3)) < 1e-6 * dp["value"]
    ex = dp["extras"]["dp_adv_lora_train_step_pgd7"]
    assert ex["ranks_seen_by_collective"] == 2 and ex["batch_per_gpu"] == 64 and ex["global_batch"] == 128
    assert ex["exchange_payload_bytes"] == (958464 + 768 * 21 + 21) * 4 and ex["exchange_ms"] > 0.0
    assert ex["max_param_divergence_across_ranks"] == 0.0          # one exchange, the same fused Adam: identical adapters everywhere
    assert ex["value"] > 0 and abs(ex["value"] - 128 / (ex["ms_per_step"] / 1e3)) < 1e-6 * ex["value"]
    assert lines["self"]["n_gpus"] == 2 and lines["self"]["config"]["ranks_seen_by_collective"] == 2
    assert lines[2]["config"]["ranks_seen_by_collective"] == 2 and lines[1]["config"]["ranks_seen_by_collective"] == 1
    kt = lines[1]["roofline"]["kernels"]
    assert kt and all(0.0 < v["frac"] < 1.0 and v["bound"] in ("mfma", "hbm") for v in kt.values())
    for tag, d in lines.items():
        n = 2 if tag == "self" else tag
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, (n, key)
        assert d["n_gpus"] == n and d["steps"] == 1 and d["warmup"] == 1 and d["unit"] == "img/s" and d["scaling"] == "weak"
        assert d["dtype"] == "f16" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["higher_is_better"] is True
        assert "workload" in d["config"] and d["config"]["global_batch"] == 32 * n
        assert abs(d["value"] - 32 * n / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
        rf = d["roofline"]
        assert rf["bound"] in ("mfma", "hbm") and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
