#!/usr/bin/env python3
"""Headline benchmark: images/sec of the Spectre-ViT-Small training step (CIFAR-100-shaped synthetic input,
bs 512 per GPU, bf16) on N MI355X -- BASELINE.json's metric.  One JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mixer fft|permut|dwt_embed|dwt_token]

A step = forward + CrossEntropy + backward + (gradient all-reduce) + AdamW on one batch resident in HBM
(reference loop: spectre_vit/repl/train.py:216-238).  N > 1: launched by torch.distributed.run, one rank per GPU,
RCCL all-reduce of the gradients overlapped with backward; weak scaling (512 images per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# configs/spectre_vit_cifar100.py:3-20 of the reference (Small / CIFAR-100)
SMALL = dict(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=4, num_heads=16,
             hidden_dim=768, dropout=0.001, activation="gelu")


def cpu_baseline(mixer, seconds_budget=20.0):
    """The numpy oracle (a port of the reference's CPU path, validated against it by tests/golden) timed on this
    host: fp32, same model, same step definition, on a bounded sample (bs 32 per step)."""
    import numpy as np
    import torch
    from oracle import spectre_oracle as O
    from spectre_vit.models.spectre.spectre import SpectreViT
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.manual_seed(42)
    cfg = dict(SMALL, dropout=0.0)
    m = SpectreViT(**cfg, mixer=mixer)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    bs = 32
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(bs, 3, 32, 32, generator=g).numpy()
    labels = torch.randint(0, 100, (bs,), generator=g).numpy()
    state = {}

    def step():
        loss, _, _, grads = O.train_step(img, labels, sd, cfg["num_encoders"], cfg["patch_size"], mixer, np.float32)
        for k, gk in grads.items():
            mv = state.setdefault(k, [np.zeros_like(gk), np.zeros_like(gk)])
            sd[k], mv[0], mv[1] = O.adamw_step(sd[k], gk.astype(np.float32), mv[0], mv[1], 1)
        return loss

    step()  # warm-up (page in BLAS, build DFT tables)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < seconds_budget and n < 50):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return dict(value=round(bs / dt, 2), unit="images/sec", cores=cores, kind="port",
                sample=f"numpy fp32 oracle, {n} train steps of bs {bs} (same model/mixer, dropout 0), {dt:.2f} s/step")


def attach_pmc_traffic(roof, mixer):
    """roofline.traffic = measured HBM bytes per launch of the dominant kernel, from the newest committed rocprofv3 PMC
    collection (tools/collect_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note
    in MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process, hence the committed file."""
    import glob
    if not roof:
        return
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{mixer}_pmc_traffic.json")))
    if not files:
        return
    data = json.load(open(files[-1]))
    if roof["kernel"] == "gemm":
        M, N, K = roof["shape"]
        nt_grid = ((M + 127) // 128) * ((N + 127) // 128) * 256
        cands = []
        if N % 256 == 0 and N <= 2048 and K % 128 == 0 and M >= 8192:
            # the strip kernel (spv_gemm.hip strip_plan): min(256 // strips, row blocks) row groups x strips workgroups of 512
            strips = N // 256
            strip_grid = min(256 // strips, (M + 31) // 32) * strips * 512
            cands += [e for e in data["kernels"] if e["kernel"].startswith("gemm_nt_strip") and e["grid_size"] == strip_grid
                      and e["kernel"].endswith("false>")]
        cands += [e for e in data["kernels"] if e["kernel"].startswith("gemm_nt_kernel") and e["grid_size"] == nt_grid]
        cands += [e for e in data["kernels"] if e["kernel"].startswith("gemm_tn") and e["grid_size"] % nt_grid == 0 and K > 4096]
    else:
        B = roof["shape"][0]
        cands = [e for e in data["kernels"] if e["kernel"].startswith("fnet_mfma") and e["grid_size"] == B * 512]
    if cands:
        roof["traffic"] = cands[0]["hbm_bytes_corrected"]
        roof["traffic_source"] = os.path.basename(files[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mixer", default="fft", choices=["fft", "permut", "dwt_embed", "dwt_token"])
    ap.add_argument("--batch", type=int, default=512, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="spectre", choices=["spectre", "vit"],
                    help="vit = the reference's baseline ViT (MHSA through the HIP attention kernels), not the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from spectre_vit import hip_ops
    from spectre_vit.dp import GradReducer, broadcast_module
    from spectre_vit.models.spectre.spectre import SpectreViT

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # SPV_BENCH_REHEARSAL=1: all ranks on device 0 over gloo -- lets the multi-rank control flow (collectives, barriers,
    # rank-0-only sections) be exercised on a one-GPU box; never used for reported numbers
    rehearsal = os.environ.get("SPV_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # the layer GEMMs leave 16 CUs to RCCL (spectre_vit.dp.RESERVED_CUS); keep RCCL inside them.  One step exchanges 80 MB
            # of fp32 gradients, which 16 channels move well inside the backward.  An explicit NCCL_MAX_NCHANNELS wins.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
            dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(42)
    if args.model == "vit":
        from spectre_vit.models.vit.vit import ViT
        model = ViT(**SMALL).to(dev)
        args.no_cpu_baseline = True  # the CPU leg times the Spectre oracle
    else:
        model = SpectreViT(**SMALL, mixer=args.mixer).to(dev)
    broadcast_module(model)
    model.train()
    torch.manual_seed(1234 + rank)  # per-rank dropout / data streams
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
    labels = torch.randint(0, 100, (args.batch,), generator=g).to(torch.uint8).to(dev)  # uint8 as train.py:218
    labels = labels.long()
    reducer = GradReducer(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, fused=True)
    crit = torch.nn.CrossEntropyLoss()
    use_bf16 = args.dtype == "bf16"

    def step():
        reducer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            out = model(img)
        loss = crit(out, labels)
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    # roofline pass: the same step, same process, right after the timed region, with HIP events recorded on the launch
    # stream around every GEMM / FNet-mixer launch.  Kept out of the headline timing because the ~60 event pairs per
    # step perturb it (measured: 6.3 ms/step bracketed vs 4.9 ms/step clean).
    # Every rank runs these steps (each one contains the gradient all-reduce: a rank that skipped them would leave the
    # others waiting in the collective); only rank 0 records events.
    timer = None
    if not args.no_roofline:
        if rank == 0:
            timer = hip_ops.KernelTimer()
            hip_ops.set_kernel_timer(timer)
        for _ in range(min(args.steps, 20)):
            step()
        torch.cuda.synchronize()
        hip_ops.set_kernel_timer(None)
    if world > 1:
        dist.barrier()
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    final_loss = float(loss.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        total_images = args.batch * world * args.steps
        rec = {
            "metric": "images/sec training, Spectre-ViT-S CIFAR-100 bs512" if args.model == "spectre"
                      else "images/sec training, baseline ViT-S CIFAR-100 bs512",
            "value": round(total_images / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if use_bf16 else "f32",
            "data": "synthetic CIFAR-shaped randn images / randint labels resident in HBM, random-init weights (seed 42)",
            "config": {"workload": (f"Spectre-ViT-Small (E512 H16 F768 L4 P4 N65, 100 classes, dropout 0.001), {args.mixer} mixer, "
                                    if args.model == "spectre" else
                                    "baseline ViT-Small (E512 H16 F768 L4 P4 N65, MHSA with the reference's batch_first=False axis), ")
                                   + f"train step fwd+CE+bwd+AdamW, bs {args.batch}/GPU",
                       **({"rehearsal": "all ranks on one device over gloo: control-flow check, not a measurement"} if rehearsal else {}),
                       "mixer": args.mixer if args.model == "spectre" else "attention", "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 4),
        }
        if timer is not None:
            rec["roofline"] = timer.roofline()
            rec["kernels"] = timer.summary()
            attach_pmc_traffic(rec["roofline"], args.mixer)
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(args.mixer)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
