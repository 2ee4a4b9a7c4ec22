"""Training harness -- the build's counterpart of reference spectre_vit/repl/train.py.

Same loop structure (train.py:207-295 classification, :298-361 distillation): seeds, model from a parsed config,
AdamW(betas, lr, weight_decay), autocast, per-epoch eval, accuracy bookkeeping on device, best-validation
``state_dict`` checkpoint.  What differs: synthetic CIFAR-shaped data resident on the GPU (no torchvision / network),
bf16 autocast (no GradScaler needed), scalars to a JSON-lines file instead of TensorBoard, optional data parallelism
(one process per GPU, RCCL all-reduce through spectre_vit.dp.GradReducer), and a synthetic frozen teacher for the
distillation path (the DINOv3 weights are unavailable offline).

    python -m spectre_vit.harness --config spectre_vit/configs/spectre_vit_cifar100.py --epochs 2 --steps-per-epoch 20
"""
from __future__ import annotations

import argparse
import json
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import nn, optim

from spectre_vit.configs.parser import parse_config
from spectre_vit.distillation import SyntheticTeacher, distillation_loss
from spectre_vit.dp import GradReducer, broadcast_module
from spectre_vit.loss import CrossEntropyLoss
from spectre_vit.models.spectre.spectre import SpectreViT

CIFAR_MEAN = (0.5071, 0.4867, 0.4408)  # train.py:109-112
CIFAR_STD = (0.2675, 0.2565, 0.2761)


def seed_everything(seed: int):
    """train.py:31-35"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_model(c, mixer="permut", device="cuda"):
    """train.py:48-59"""
    return SpectreViT(img_size=c.img_size, patch_size=c.patch_size, in_channels=c.in_channels, num_classes=c.num_classes,
                      embed_dim=c.embed_dim, num_encoders=c.num_encoders, num_heads=c.num_heads, hidden_dim=c.hidden_dim,
                      dropout=c.dropout, activation=c.activation, mixer=mixer).to(device)


class SyntheticCifar:
    """class-conditional uint8 images resident in HBM: a fixed random template per class plus noise, so a model can
    actually learn something; normalised like train.py:109-112 when a batch is drawn."""

    def __init__(self, n, c, device, seed=0):
        g = torch.Generator().manual_seed(seed)
        templates = torch.rand(c.num_classes, c.in_channels, c.img_size, c.img_size, generator=torch.Generator().manual_seed(1))
        self.labels = torch.randint(0, c.num_classes, (n,), generator=g)
        noise = torch.rand(n, c.in_channels, c.img_size, c.img_size, generator=g)
        self.images = ((0.6 * templates[self.labels] + 0.4 * noise) * 255).to(torch.uint8).to(device)
        self.labels = self.labels.to(torch.uint8 if c.num_classes <= 256 else torch.int64).to(device)  # uint8 as train.py:218
        self.mean = torch.tensor(CIFAR_MEAN[:c.in_channels], device=device).view(1, -1, 1, 1)
        self.std = torch.tensor(CIFAR_STD[:c.in_channels], device=device).view(1, -1, 1, 1)

    def batches(self, batch_size, shuffle, generator=None, rank=0, world=1, raw_uint8=False, drop_last=True):
        """raw_uint8: yield the uint8 NHWC batch itself; the model's patch gather normalises it.
        drop_last=False (validation): the short tail batch is yielded too, so every sample of the rank's shard is seen
        (the reference's DataLoader default, train.py:151-155)."""
        n = self.images.shape[0]
        idx = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        idx = idx[rank::world].to(self.images.device)
        stop = idx.numel() - batch_size + 1 if drop_last else idx.numel()
        for i in range(0, stop, batch_size):
            sel = idx[i:i + batch_size]
            if raw_uint8:
                yield self.images[sel].permute(0, 2, 3, 1).contiguous(), self.labels[sel]
                continue
            img = (self.images[sel].float() / 255.0 - self.mean) / self.std
            yield img, self.labels[sel]


def train(config_path, mixer="permut", epochs=1, steps_per_epoch=None, batch_size=None, n_train=4096, n_val=1024,
          use_amp=True, distill=False, out_dir="runs/spectre_vit", log=print, uint8_input=False, graph=False):
    """graph=True (not with distill): the training step -- zero_grad, forward, loss, backward, AdamW -- is replayed from HIP graphs
    (spectre_vit.graph: one graph in a single process; as a rank of a torch.distributed job two graphs around ONE all-reduce of the
    flat gradient buffer) with the one-launch optimizer (spectre_vit.optim.FusedAdamW: torch.optim.AdamW's rule and state layout).
    The default is the reference's own loop shape (train.py:216-238) with the overlapped bucket exchange under data parallelism."""
    c = parse_config(config_path)
    seed = getattr(c, "random_seed", 42)
    lr = getattr(c, "learning_rate", 1e-3)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)  # always: kernels launch on the current device's stream (also with a pre-initialised group)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    device = torch.device("cuda", local_rank)
    seed_everything(seed)
    model = build_model(c, mixer, device)
    broadcast_module(model)
    batch_size = batch_size or c.batch_size
    train_set = SyntheticCifar(n_train, c, device, seed=seed)
    val_set = SyntheticCifar(n_val, c, device, seed=seed + 1)
    criterion = CrossEntropyLoss()  # nn.CrossEntropyLoss() of train.py:196 on the HIP path (spectre_vit/loss.py)
    if graph and distill:
        raise ValueError("graph=True replays the plain training step; the distillation step (teacher forward + KD loss) runs eagerly")
    gstep = None
    if graph:
        from spectre_vit.optim import FusedAdamW
        optimizer = FusedAdamW(model.parameters(), betas=c.adam_betas, lr=lr, weight_decay=c.adam_weight_decay, capturable=True,
                               static_grads=True)
        reducer = None   # the graphed step owns its own (fixed-address) gradient buffer
    else:
        optimizer = optim.AdamW(model.parameters(), betas=c.adam_betas, lr=lr, weight_decay=c.adam_weight_decay)  # train.py:199-201
        reducer = GradReducer(model)
    teacher = SyntheticTeacher(c.num_classes, 384, c.in_channels).to(device) if distill else None
    os.makedirs(out_dir, exist_ok=True)
    log_f = open(os.path.join(out_dir, "scalars.jsonl"), "a") if rank == 0 else None
    gen = torch.Generator().manual_seed(seed)
    best_acc, history = 0.0, []
    start = time.perf_counter()
    for epoch in range(epochs):
        model.train()
        running = torch.zeros((), device=device)
        correct = torch.zeros((), device=device, dtype=torch.int64)
        total, steps = 0, 0
        for img, label in train_set.batches(batch_size, True, gen, rank, world, raw_uint8=uint8_input and not distill):
            if graph:
                if gstep is None:   # built on the first batch (its shape is the captured one); warm-up steps are real training steps
                    from spectre_vit.graph import GraphedDPStep, GraphedTrainStep
                    cls = GraphedDPStep if world > 1 else GraphedTrainStep
                    gstep = cls(model, optimizer, criterion, img, label.long(), autocast_dtype=torch.bfloat16 if use_amp else None, warmup=1)
                    loss, y_pred = gstep.warm_loss, gstep.warm_out   # the warm-up step WAS this batch's training step
                else:
                    loss = gstep(img, label.long())
                    y_pred = gstep.out
                correct += (label == torch.argmax(y_pred, dim=1)).sum()
                total += label.size(0)
                running += loss.detach()
                steps += 1
                if steps_per_epoch and steps >= steps_per_epoch:
                    break
                continue
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp and not distill):  # distill: use_amp False, train.py:299
                if distill:
                    student_logits, _ = model(img, return_features=True)
                    with torch.no_grad():
                        teacher_logits, _ = teacher(nn.functional.interpolate(img, size=64, mode="bicubic"), return_features=True)
                    loss, _, _ = distillation_loss(student_logits, teacher_logits, label.long())
                    y_pred = student_logits
                else:
                    y_pred = model(img)
            correct += (label == torch.argmax(y_pred, dim=1)).sum()
            total += label.size(0)
            if not distill:
                loss = criterion(y_pred, label.long())
            reducer.zero_grad()
            loss.backward()
            reducer.finish()
            optimizer.step()
            running += loss.detach()  # accumulated on device: no host sync per step (train.py:243 syncs every step)
            steps += 1
            if steps_per_epoch and steps >= steps_per_epoch:
                break
        train_loss = (running / max(steps, 1)).item()
        train_acc = correct.item() / max(total, 1)

        model.eval()
        v_correct = torch.zeros((), device=device, dtype=torch.int64)
        v_loss = torch.zeros((), device=device)
        v_total, v_steps = 0, 0
        with torch.no_grad():
            for img, label in val_set.batches(min(getattr(c, "val_batch_size", batch_size), n_val), False, None, rank, world,
                                              raw_uint8=uint8_input and not distill, drop_last=False):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp and not distill):
                    y_pred = model(img)
                v_correct += (label == torch.argmax(y_pred, dim=1)).sum()
                v_loss += criterion(y_pred, label.long()) * label.size(0)  # sample-weighted: the tail batch is short
                v_total += label.size(0)
                v_steps += 1
        stats = torch.stack([v_correct.float(), torch.tensor(float(v_total), device=device), v_loss])
        if world > 1:
            dist.all_reduce(stats)
        val_acc = (stats[0] / stats[1].clamp(min=1)).item()
        val_loss = (stats[2] / stats[1].clamp(min=1)).item()
        rec = {"epoch": epoch + 1, "Loss/Train": train_loss, "Loss/Validation": val_loss, "Accuracy/Train": train_acc,
               "Accuracy/Validation": val_acc, "steps": steps, "val_samples": int(stats[1].item())}
        history.append(rec)
        if rank == 0:
            log_f.write(json.dumps(rec) + "\n")
            log_f.flush()
            log(rec)
            if val_acc > best_acc or epoch == 0:  # train.py:288-290
                best_acc = max(best_acc, val_acc)
                torch.save(model.state_dict(), os.path.join(out_dir, "model_best.pt"))
    if gstep is not None:
        gstep.close()
    if rank == 0:
        log_f.write(json.dumps({"Training time": time.perf_counter() - start}) + "\n")
        log_f.close()
    return model, history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="spectre_vit/configs/spectre_vit_cifar100.py")
    ap.add_argument("--mixer", default="permut")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps-per-epoch", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--distill", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the training step from HIP graphs (spectre_vit.graph)")
    ap.add_argument("--out", default="runs/spectre_vit")
    a = ap.parse_args()
    train(a.config, a.mixer, a.epochs, a.steps_per_epoch, a.batch_size, distill=a.distill, out_dir=a.out, graph=a.graph)


if __name__ == "__main__":
    main()
