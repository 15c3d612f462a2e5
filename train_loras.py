#!/usr/bin/env python3
"""LoRA adversarial-defence training on MI355X -- command-line compatible with the reference's
train_loras.py (flags :427-442) and writing what it writes:

    <output_dir>/<model>/<source>/<attack>/rank{r}_best_adapter/    best epoch by validation accuracy (:331-351)
    <output_dir>/<model>/<source>/<attack>/rank{r}_final_adapter/   after the last epoch (:353-354)
    <output_dir>/<model>/<source>/<attack>/results.json             {rank: {train_loss, train_acc, val_loss, val_acc,
                                                                     val_f1, clean_test_acc, clean_test_f1, adv_test_acc,
                                                                     adv_test_f1, best_val_acc}} (:372-385)
    <output_dir>/global_results.json                                (:472-475)

One peft-style adapter per (attack, rank): frozen backbone, trainable LoRA A/B + classifier, Adam(lr),
CrossEntropyLoss -- the loop of train_loras.py:295-324 written against the same objects
(`peft_model.base_model(pixel_values=x).logits`, `criterion(logits, labels).backward()`, `optimizer.step()`),
then validate() (:17-53), best-adapter selection and test_model() (:56-76) on the forward kernel.

Differences, all documented in INTEGRATION.md:
  * the reference only proceeds for ('google_vit', 'mapillary') (:120-122); any pair with a checkpoint runs here;
  * --batch_size is the GLOBAL batch; under torch.distributed.run it is sharded over the ranks, every rank runs
    the same number of optimizer steps (optim.global_batch_plan) and the flat LoRA + classifier gradient is summed
    with ONE all-reduce per step (RCCL over xGMI), weighted by shard size;
  * --pgd-inner-steps K   generate the adversarial batch on the fly with PGD-K on the CURRENT model (BASELINE
                          config 3) instead of reading pre-generated PNGs;
  * --synthetic N         seeded random images / weights, no files needed;  --arch tiny|vit_b|vit_l.
"""
import argparse
import importlib
import json
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
V = importlib.import_module("adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd")
iomod = importlib.import_module(V.__name__ + ".io")
syn = importlib.import_module(V.__name__ + ".synthetic")
opt = importlib.import_module(V.__name__ + ".optim")


def build_parser():
    p = argparse.ArgumentParser(description="Train LoRA for adversarial defense (MI355X / HIP)")
    p.add_argument("--models", nargs="+", default=["google_vit"])
    p.add_argument("--sources", nargs="+", default=["mapillary"])
    p.add_argument("--attacks", nargs="+", default=["patch_circle", "patch_square", "pgd", "fgsm"])
    p.add_argument("--model_base_path", default="./train24/{model}/{source}/{model}_best_model_finetuned.pth")
    p.add_argument("--adv_root", default=None, help="Root directory for adversarial examples")
    p.add_argument("--data_root", default=None, help="Root directory for clean examples")
    p.add_argument("--output_dir", required=True, help="Base directory to save LoRA parameters")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--epochs", type=int, default=4)
    p.add_argument("--ranks", nargs="+", type=int, default=[8, 16, 32])
    p.add_argument("--lora_dropout", type=float, default=0.1)
    p.add_argument("--pgd-inner-steps", type=int, default=0)
    p.add_argument("--epsilon", type=float, default=8 / 255)
    p.add_argument("--pgd_alpha", type=float, default=2 / 255)
    p.add_argument("--synthetic", type=int, default=0, metavar="N")
    p.add_argument("--num_classes", type=int, default=21, help="only with --synthetic")
    p.add_argument("--arch", choices=sorted(syn.ARCHS), default="vit_b")
    p.add_argument("--precision", choices=["f16", "bf16", "f32"], default="f16")
    p.add_argument("--seed", type=int, default=0)
    return p


class Dist:
    """Rank bookkeeping + the few collectives the script needs (no-ops in a single process)."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        self.local = int(os.environ.get("LOCAL_RANK", 0))
        # rehearsal on a one-GPU box (tests/test_hip_pipeline.py): VITLORA_SHARE_GPU=1 puts every rank on cuda:0 and
        # VITLORA_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
        if os.environ.get("VITLORA_SHARE_GPU") == "1":
            self.local = 0
        self.device = torch.device("cuda", self.local)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local)
            if not dist.is_initialized():
                backend = os.environ.get("VITLORA_DIST_BACKEND", "nccl")
                if backend == "nccl":
                    dist.init_process_group("nccl", device_id=self.device)
                else:
                    dist.init_process_group(backend)

    def sum_(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t)
        return t

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()


class TensorSet:
    """An in-memory split (synthetic mode): images [N,3,S,S] in [0,1] and labels."""

    def __init__(self, x, y):
        self.x, self.y = x, y

    def __len__(self):
        return len(self.y)

    def fetch(self, idx):
        return self.x[idx], self.y[idx]


class FolderSet:
    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def fetch(self, idx):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:      # PNG decode off the main thread
            items = list(ex.map(self.ds.__getitem__, idx))
        return torch.stack([it[0] for it in items]), torch.tensor([it[1] for it in items], dtype=torch.int64)


def confusion_stats(conf):
    """accuracy and sklearn's f1_score(average='weighted') from a confusion matrix [true, pred]."""
    conf = conf.double()
    n = conf.sum().clamp_min(1)
    acc = float(conf.diag().sum() / n)
    tp, support, predicted = conf.diag(), conf.sum(1), conf.sum(0)
    f1 = torch.where(tp > 0, 2 * tp / (support + predicted).clamp_min(1), torch.zeros_like(tp))
    return acc, float((f1 * support).sum() / n)


@torch.no_grad()
def evaluate(peft_model, dset, args, D, mean, std, criterion=None):
    """validate() / test_model() of the reference (:17-76): loss, accuracy, weighted F1 over a split, forward only.
    Ranks take shards of every batch; counts are summed with one all-reduce at the end."""
    peft_model.eval()
    peft_model._vit._engine().set_normalization(mean, std)
    C = peft_model._vit.arch.num_labels
    conf = torch.zeros(C, C, device=D.device)
    loss_sum = torch.zeros((), device=D.device)
    for idx, _ in opt.global_batch_plan(len(dset), args.batch_size, D.rank, D.world):
        if not idx:
            continue
        x, y = dset.fetch(idx)
        x, y = x.to(D.device), y.to(D.device)
        logits = peft_model.base_model(pixel_values=x, normalise=True).logits      # (x - mean) / std inside the patch gather
        if criterion is not None:
            loss_sum += criterion(logits, y) * len(idx)
        conf.index_put_((y, logits.argmax(1)), torch.ones(len(idx), device=D.device), accumulate=True)
    D.sum_(conf)
    D.sum_(loss_sum)
    acc, f1 = confusion_stats(conf.cpu())
    return float(loss_sum / max(1, len(dset))), acc, f1


TELEMETRY = {}        # (out_dir, rank) -> {"fp16_skipped_steps", "optimizer_steps", "precision"} of the last run in this process


def train_rank(args, D, base_model, sets, rank_r, out_dir, mean, std):
    """One adapter: train_loras.py:269-385."""
    peft_model = V.setup_peft_lora(base_model, rank=rank_r, dropout=args.lora_dropout, seed=args.seed + 31 * rank_r)   # reproducible init
    vit = peft_model._vit
    engine = vit._engine()
    if D.world > 1:
        import torch.distributed as dist
        dist.broadcast(vit.trainable_flat().data, src=0)           # identical initial adapters on every rank
        vit.mark_dirty()
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = V.Adam(peft_model.parameters(), lr=args.lr, model=peft_model)
    res = {"train_loss": [], "train_acc": [], "val_loss": [], "val_acc": [], "val_f1": []}
    best_val_acc = 0.0
    skipped = steps_total = 0            # fp16 telemetry: optimizer steps dropped because a gradient left the fp16 range
    train = sets["train"]
    for epoch in range(args.epochs):
        if D.rank == 0:
            print(f"\nEpoch {epoch + 1}/{args.epochs}")
        peft_model.train()
        engine.set_normalization(mean, std)
        stats = torch.zeros(3, device=D.device)                  # loss * n, correct, n  (no per-step host sync)
        for step, (idx, n_global) in enumerate(opt.global_batch_plan(len(train), args.batch_size, D.rank, D.world,
                                                                    shuffle_seed=args.seed + 1000 * rank_r + epoch)):
            optimizer.zero_grad()
            ok = True
            if idx:
                x, y = train.fetch(idx)
                x, y = x.to(D.device), y.to(D.device)
                try:
                    xa = x
                    if args.pgd_inner_steps > 0:
                        # adversarial batch against the CURRENT adapters (the library commits the last Adam step itself)
                        engine.set_normalization(mean, std)
                        xa = engine.pgd_attack(x, y, args.epsilon, args.pgd_alpha, args.pgd_inner_steps, random_start=True,
                                               seed=args.seed + 7919 * epoch + step)
                    logits = peft_model.base_model(pixel_values=xa, normalise=True).logits
                    loss = criterion(logits, y)
                    loss.backward()
                    # fp16 mode: consume the out-of-range flag HERE, so that it belongs to THIS step (code 2: a backward kernel
                    # clamped a gradient to +-65504 -- everything downstream is finite but wrong, Adam would apply it in full)
                    engine.check()
                except V.NonFiniteGradient as e:
                    ok = False
                    print(f"  [rank {D.rank}] epoch {epoch + 1} step {step}: fp16 gradient out of range, the step is dropped on every rank ({e})",
                          flush=True)
                if ok:
                    stats += torch.stack([loss.detach() * len(idx), (logits.detach().argmax(1) == y).sum().float(),
                                          torch.tensor(float(len(idx)), device=D.device)])
            if not ok:
                # what an AMP skip-step does, data-parallel: a NaN gradient rides the ONE all-reduce of this step to every rank
                # and the fused Adam leaves every element (parameter and moments) untouched where its gradient is not finite
                flat = optimizer.params[0]
                flat.grad = torch.full_like(flat.data, float("nan"))
            optimizer.step(local_count=len(idx), global_count=n_global)
            try:
                engine.check()            # the Adam kernel's own flag (code 3): identical on every rank after the all-reduce
            except V.NonFiniteGradient:
                skipped += 1
                optimizer.drop_last_step()      # a dropped step does not advance Adam's bias-correction count
            steps_total += 1
        D.sum_(stats)
        res["train_loss"].append(float(stats[0] / stats[2].clamp_min(1)))
        res["train_acc"].append(float(stats[1] / stats[2].clamp_min(1)))
        if sets.get("val") is not None:
            vl, va, vf = evaluate(peft_model, sets["val"], args, D, mean, std, criterion)
            res["val_loss"].append(vl); res["val_acc"].append(va); res["val_f1"].append(vf)
            score = va
            if D.rank == 0:
                print(f"Train Loss: {res['train_loss'][-1]:.4f} Acc: {res['train_acc'][-1]:.4f}")
                print(f"Val Loss: {vl:.4f} Acc: {va:.4f} F1: {vf:.4f}")
        else:
            score = res["train_acc"][-1]
            if D.rank == 0:
                print(f"Train Loss: {res['train_loss'][-1]:.4f} Acc: {res['train_acc'][-1]:.4f}")
        if score > best_val_acc:                                   # identical on every rank (all-reduced counts)
            best_val_acc = score
            if D.rank == 0:
                best = os.path.join(out_dir, f"rank{rank_r}_best_adapter")
                peft_model.save_pretrained(best)
                print(f"New best LoRA adapter saved to: {best}")
    if D.rank == 0:
        final = os.path.join(out_dir, f"rank{rank_r}_final_adapter")
        peft_model.save_pretrained(final)
        print(f"Final LoRA adapter saved to: {final}")
        if not os.path.isdir(os.path.join(out_dir, f"rank{rank_r}_best_adapter")):     # no epoch beat 0.0 accuracy
            peft_model.save_pretrained(os.path.join(out_dir, f"rank{rank_r}_best_adapter"))
    _, clean_acc, clean_f1 = evaluate(peft_model, sets["test_clean"], args, D, mean, std)
    if sets.get("test_adv") is not None:
        _, adv_acc, adv_f1 = evaluate(peft_model, sets["test_adv"], args, D, mean, std)
    else:
        adv_acc, adv_f1 = 0.0, 0.0
    if D.rank == 0:
        print(f"Clean Test Accuracy: {clean_acc:.4f}, F1: {clean_f1:.4f}")
        print(f"Adversarial Test Accuracy: {adv_acc:.4f}, F1: {adv_f1:.4f}")
    if D.rank == 0:
        print(f"fp16 range: {skipped} of {steps_total} optimizer steps dropped (VL_ERR_NONFINITE events)")
    res.update({"clean_test_acc": clean_acc, "clean_test_f1": clean_f1, "adv_test_acc": adv_acc, "adv_test_f1": adv_f1,
                "best_val_acc": best_val_acc})
    # (results.json keeps the reference's schema, train_loras.py:366-385: the range telemetry goes beside it)
    TELEMETRY[(out_dir, rank_r)] = {"fp16_skipped_steps": skipped, "optimizer_steps": steps_total, "precision": args.precision}
    if D.rank == 0:
        with open(os.path.join(out_dir, f"rank{rank_r}_fp16_telemetry.json"), "w") as f:
            json.dump(TELEMETRY[(out_dir, rank_r)], f)
    return res


def load_sets(args, model_name, source, attack, class_to_idx, arch):
    """train / val / test_adv from <adv_root>/<model>/<source>/<split>/<attack>/ (whitebox_attacks.py output),
    test_clean from <data_root>/test filtered by source (train_loras.py:157-235).  Images are returned in [0,1];
    normalisation happens on the device."""
    S = arch.image_size
    if args.synthetic:
        n = args.synthetic
        mk = lambda k, m: TensorSet(*syn.random_batch(arch, m, seed=args.seed + 7 + k))
        return {"train": mk(0, n), "val": mk(1, max(8, n // 4)), "test_adv": mk(2, max(8, n // 4)),
                "test_clean": mk(3, max(8, n // 4))}
    sets = {}
    for split, key in (("train", "train"), ("val", "val"), ("test", "test_adv")):
        d = os.path.join(args.adv_root, model_name, source, split, attack)
        meta = os.path.join(d, "metadata.csv")
        if os.path.exists(meta):
            sets[key] = FolderSet(iomod.FolderDataset(d, meta, class_to_idx, image_size=S))
    if "train" not in sets:
        return None
    sets.setdefault("val", None)
    sets.setdefault("test_adv", None)
    clean_meta = os.path.join(args.data_root, "test", "metadata.csv")
    sets["test_clean"] = FolderSet(iomod.FolderDataset(args.data_root, clean_meta, class_to_idx, image_size=S, sources=[source]))
    return sets


def train_lora_for_model_and_attack(model_name, source, attack, args, D):
    mean, std = V.get_normalization(model_name)
    if args.synthetic:
        arch = syn.arch_by_name(args.arch, args.num_classes)
        sd = syn.random_state_dict(arch, seed=args.seed)
        class_to_idx = {f"class_{i}": i for i in range(args.num_classes)}
    else:
        path = args.model_base_path.format(model=model_name, source=source)
        mapping = os.path.join(os.path.dirname(path), "class_mappings.txt")
        if not (os.path.exists(path) and os.path.exists(mapping)):
            print(f"Class mapping or checkpoint not found: {mapping} / {path}")
            return {}
        class_to_idx = iomod.read_class_mappings(mapping)
        arch = syn.arch_by_name(args.arch, len(class_to_idx))
        sd = torch.load(path, map_location="cpu", weights_only=True)
    out_dir = os.path.join(args.output_dir, model_name, source, attack)
    if D.rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    sets = load_sets(args, model_name, source, attack, class_to_idx, arch)
    if sets is None:
        print(f"No data found for attack: {attack}")
        return {}
    all_results = {}
    for r in args.ranks:
        if D.rank == 0:
            print(f"\n{'=' * 50}\nTraining {model_name} on {source} with {attack} attack, rank {r}\n{'=' * 50}")
        base = V.create_vit_model(arch.num_labels, arch=arch, device=D.device, precision=args.precision)
        base.load_state_dict(sd)
        all_results[r] = train_rank(args, D, base, sets, r, out_dir, mean, std)
    if D.rank == 0:
        with open(os.path.join(out_dir, "results.json"), "w") as f:
            json.dump(all_results, f, indent=4)
        print(f"\nAll results saved to: {os.path.join(out_dir, 'results.json')}")
    return all_results


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.synthetic and not (args.adv_root and args.data_root):
        raise SystemExit("--adv_root and --data_root are required unless --synthetic N is given")
    D = Dist()
    if D.rank == 0:
        print(f"Using device: {D.device} (world size {D.world})")
    global_results = {}
    for model_name in args.models:
        for source in args.sources:
            for attack in args.attacks:
                err = None
                results = {}
                try:
                    results = train_lora_for_model_and_attack(model_name, source, attack, args, D)
                except Exception as e:            # skip-and-continue like the reference (:464-470) ...
                    err = e
                    print(f"Error training {model_name} on {source} with {attack}: {e}", flush=True)
                    print(traceback.format_exc(), flush=True)
                    if D.world > 1:
                        # ... but ONLY in a single process.  With several ranks the others may be anywhere inside the
                        # unit -- the gradient all-reduce of Adam.step, evaluate()'s sums, the initial broadcast -- and any
                        # collective entered from here would pair up with one of those (a hang or silently wrong sums under
                        # RCCL, a size mismatch under gloo).  So no collective from the failure path: this rank exits
                        # non-zero at once and torch.distributed.run tears the job down.
                        sys.stdout.flush()
                        sys.stderr.flush()
                        os._exit(1)
                global_results.setdefault(model_name, {}).setdefault(source, {})[attack] = results
    if D.rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        path = os.path.join(args.output_dir, "global_results.json")
        with open(path, "w") as f:
            json.dump(global_results, f, indent=4)
        print(f"\nGlobal results saved to: {path}")
    D.barrier()
    return global_results


if __name__ == "__main__":
    main()
