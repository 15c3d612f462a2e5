#!/usr/bin/env python3
"""Base / single-adapter / merged-adapter accuracy sweep on MI355X -- the forward-only composition
experiment of the reference's eval_compose.py (flags :436-447; adapter discovery
<lora_root>/google_vit/mapillary/<attack>/rank<r>_best_adapter :197-208; sequential
PeftModel.from_pretrained -> merge_and_unload :102-114; result JSON layout :473-494).

Every forward runs on the HIP engine; a merge is W <- W + (alpha/r) B A on the device
(vl_merge_weight), adapter after adapter, as peft's merge_and_unload does it; the classifier of
the LAST merged adapter wins (peft modules_to_save semantics).

  --synthetic N   no dataset / checkpoints on disk: seeded random-init base weights, one seeded
                  random adapter per attack name and N random images per test set (plumbing and
                  throughput runs; accuracies are then meaningless).
"""
import argparse
import importlib
import itertools
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")
TARGETS = ["query", "key", "value", "output.dense"]          # train_loras.py:81


def accuracy_and_weighted_f1(labels, preds, num_classes):
    """accuracy_score and f1_score(average='weighted') of sklearn, restated (eval_compose.py:55-56)."""
    labels, preds = torch.as_tensor(labels), torch.as_tensor(preds)
    acc = float((labels == preds).float().mean()) if len(labels) else 0.0
    f1w, n = 0.0, max(1, len(labels))
    for c in range(num_classes):
        support = int((labels == c).sum())
        if not support:
            continue
        tp = int(((labels == c) & (preds == c)).sum())
        fp = int(((labels != c) & (preds == c)).sum())
        fn = support - tp
        f1 = 2 * tp / (2 * tp + fp + fn) if tp else 0.0
        f1w += f1 * support / n
    return acc, f1w


@torch.no_grad()
def test_model(model, batches, device, num_classes):
    preds, labels, t0, n = [], [], time.perf_counter(), 0
    for images, y, _ in batches():
        out = model.base_model(pixel_values=images.to(device)) if hasattr(model, "base_model") else model(pixel_values=images.to(device))
        preds.append(V.get_model_output(out).argmax(1).cpu())
        labels.append(torch.as_tensor(y).cpu())
        n += len(y)
    torch.cuda.synchronize()
    acc, f1 = accuracy_and_weighted_f1(torch.cat(labels), torch.cat(preds), num_classes)
    return acc, f1, n / max(1e-9, time.perf_counter() - t0)


def merge_lora_adapters(base_model, adapter_paths):
    cur = base_model
    for i, p in enumerate(adapter_paths):
        print(f"Loading adapter {i + 1}/{len(adapter_paths)} from {p}")
        cur = V.PeftModel.from_pretrained(cur, p).merge_and_unload()
    return cur


def main(argv=None):
    ap = argparse.ArgumentParser(description="Test base model and LoRA adapters (MI355X / HIP)")
    ap.add_argument("--model_path", default=None, help="Path to base fine-tuned model")
    ap.add_argument("--lora_root", default=None, help="Root directory containing LoRA adapters")
    ap.add_argument("--adv_root", default=None, help="Root directory for adversarial examples")
    ap.add_argument("--data_root", default=None, help="Root directory for clean examples")
    ap.add_argument("--attacks", nargs="+", required=True, help="List of attacks to evaluate")
    ap.add_argument("--rank", type=int, required=True, help="Rank value to evaluate (e.g., 16)")
    ap.add_argument("--output_file", default="test_results.json")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--test_mode", choices=["all", "base_only", "individual_only", "combinations_only"], default="all")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N")
    ap.add_argument("--num_classes", type=int, default=21, help="only with --synthetic")
    ap.add_argument("--arch", choices=["tiny", "vit_b", "vit_l"], default="vit_b")
    ap.add_argument("--model_name", default="google_vit")
    ap.add_argument("--source", default="mapillary")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    syn = importlib.import_module(V.__name__ + ".synthetic")
    iomod = importlib.import_module(V.__name__ + ".io")
    mean, std = V.get_normalization("google_vit")
    tmp = None

    if args.synthetic:
        num_classes = args.num_classes
        arch = syn.arch_by_name(args.arch, num_classes)
        base_sd = syn.random_state_dict(arch, seed=args.seed)

        def load_base():
            m = V.create_vit_model(num_classes, arch=arch)
            m.load_state_dict(base_sd)
            return m.to(device).eval()

        tmp = tempfile.TemporaryDirectory()
        adapters = {}
        for k, attack in enumerate(args.attacks):        # one seeded adapter per attack name, peft directory format
            pm = V.setup_peft_lora(load_base(), rank=args.rank, alpha=16, dropout=0.0, target_modules=TARGETS)
            eng = pm._vit._engine()
            for (i, t), (A, B) in syn.random_lora(arch, args.rank, ("q", "k", "v", "o", "fc2"), seed=args.seed + 100 + k).items():
                eng.param(i, t, "A").copy_(A)
                eng.param(i, t, "B").copy_(B)
            adapters[attack] = os.path.join(tmp.name, attack)
            pm.save_pretrained(adapters[attack])
        sets = {}
        for k, name in enumerate(["clean"] + list(args.attacks)):
            x, y = syn.random_batch(arch, args.synthetic, seed=args.seed + 500 + k)
            xn = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
            sets[name] = (lambda xn=xn, y=y: ((xn[i:i + args.batch_size], y[i:i + args.batch_size], None)
                                             for i in range(0, len(y), args.batch_size)))
    else:
        for need in ("model_path", "lora_root", "adv_root", "data_root"):
            if not getattr(args, need):
                raise SystemExit(f"--{need} is required unless --synthetic N is given")
        class_to_idx = iomod.read_class_mappings(os.path.join(os.path.dirname(args.model_path), "class_mappings.txt"))
        num_classes = len(class_to_idx)

        arch = syn.arch_by_name(args.arch, num_classes)

        def load_base():
            m = V.create_vit_model(num_classes, arch=arch)
            m.load_state_dict(torch.load(args.model_path, map_location="cpu", weights_only=True))
            return m.to(device).eval()

        def loader(root, meta, sources=None):
            ds = iomod.FolderDataset(root, meta, class_to_idx, image_size=arch.image_size, sources=sources, normalise=(mean, std))
            return lambda: torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=False,
                                                       num_workers=iomod.loader_workers(len(ds)))

        sets = {"clean": loader(args.data_root, os.path.join(args.data_root, "test", "metadata.csv"), [args.source])}
        adv_base = os.path.join(args.adv_root, args.model_name, args.source, "test")
        for name in sorted(os.listdir(adv_base)) if os.path.isdir(adv_base) else []:
            meta = os.path.join(adv_base, name, "metadata.csv")
            if os.path.exists(meta):
                sets[name] = loader(os.path.join(adv_base, name), meta)
        adapters = {}
        for attack in args.attacks:
            p = os.path.join(args.lora_root, args.model_name, args.source, attack, f"rank{args.rank}_best_adapter")
            if os.path.exists(p):
                adapters[attack] = p
            else:
                print(f"Warning: LoRA adapter not found for {attack} (rank {args.rank}) at {p}")
        if not adapters:
            print("No LoRA adapters found for the specified attacks and rank!")
            return

    results = {"rank": args.rank, "attacks_evaluated": args.attacks, "test_datasets": list(sets.keys())}

    def sweep(key, model):
        res = {}
        for name, batches in sets.items():
            acc, f1, rate = test_model(model, batches, device, num_classes)
            res[name] = {"accuracy": acc, "f1_score": f1}
            print(f"{key} on {name}: Accuracy = {acc:.4f}, F1 = {f1:.4f}  ({rate:.0f} img/s)")
        results[key] = res

    if args.test_mode in ("all", "base_only"):
        sweep("base_model", load_base())
    if args.test_mode in ("all", "individual_only"):
        for attack, path in adapters.items():
            sweep(f"{attack}_lora", V.PeftModel.from_pretrained(load_base(), path))
    if args.test_mode in ("all", "combinations_only") and len(adapters) >= 2:
        names = list(adapters)
        combos = [c for k in (2, 3) if k < len(names) for c in itertools.combinations(names, k)] + [tuple(names)]
        for combo in dict.fromkeys(combos):               # 2-, 3- and all-adapter merges (eval_compose.py:275-433)
            sweep("+".join(combo) + "_merged", merge_lora_adapters(load_base(), [adapters[a] for a in combo]))

    with open(args.output_file, "w") as f:
        json.dump(results, f, indent=4)
    print(f"\nResults saved to: {args.output_file}")
    if tmp is not None:
        tmp.cleanup()
    return results


if __name__ == "__main__":
    main()
