#!/usr/bin/env python3
"""Headline benchmark: images/sec of the Spectre-ViT-Small training step (CIFAR-100-shaped synthetic input,
bs 512 per GPU, bf16) on N MI355X -- BASELINE.json's metric.  One JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mixer fft|permut|dwt_embed|dwt_token] [--eager] [--dp-sequence]

A step = forward + CrossEntropy + backward + (gradient all-reduce) + AdamW on one batch resident in HBM
(reference loop: spectre_vit/repl/train.py:216-238).  N > 1: one rank per GPU; a rank replays two HIP graphs around ONE RCCL
all-reduce of its flat gradient buffer (spectre_vit.graph.GraphedDPStep; the HEAD mixer's 80 MB: eager, bucket all-reduces
overlapped with backward); weak scaling (512 images per GPU).  ``python bench.py --gpus N`` starts its own ranks (a child
``torch.distributed.run`` spawned before this process touches the GPU); under an external ``torch.distributed.run`` (WORLD_SIZE
set) it is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# configs/spectre_vit_cifar100.py:3-20 of the reference (Small / CIFAR-100)
SMALL = dict(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=4, num_heads=16,
             hidden_dim=768, dropout=0.001, activation="gelu")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mixer", default="fft", choices=["fft", "permut", "dwt_embed", "dwt_token"])
    ap.add_argument("--batch", type=int, default=512, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="spectre", choices=["spectre", "vit"],
                    help="vit = the reference's baseline ViT (MHSA through the HIP attention kernels), not the headline workload")
    ap.add_argument("--eager", action="store_true",
                    help="headline = eager launches (default at 1 GPU: the step replayed from ONE HIP graph, spectre_vit.graph."
                         "GraphedTrainStep, with the eager time reported beside it; multi-GPU runs are always eager)")
    ap.add_argument("--graph", action="store_true", help="(kept for compatibility: graph replay is the single-GPU default)")
    ap.add_argument("--variants", default=None,
                    help="comma list of other mixers to time for a few steps each and report under \"variants\" "
                         "(default at 1 GPU with the fft mixer: permut,dwt_embed; 'none' disables)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed steps of the CPU baseline leg (bs = --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dp-sequence", action="store_true",
                    help="one GPU: the headline is the launch sequence of a data-parallel rank (graph A -> all-reduce in a one-rank RCCL "
                         "group -> graph B) instead of the single graph")
    ap.add_argument("--dp-graph", action="store_true", help="N GPUs, permut mixer: use the two-graph rank step as well (default: eager, overlapped)")
    ap.add_argument("--no-dp-sequence", action="store_true", help="skip the dp_sequence leg of the single-GPU line")
    ap.add_argument("--no-script-leg", action="store_true", help="skip the as_script leg (the reference loop verbatim) of the single-GPU line")
    ap.add_argument("--no-base224", action="store_true", help="skip the Base/224 student leg (BASELINE config 5, one GPU) of the default single-GPU line")
    ap.add_argument("--no-every-row", action="store_true",
                    help="skip the second graph-replayed measurement with the last layer's feed-forward half over every row (profiling runs)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start N ranks with torch.distributed.run as a CHILD process (this parent has not
    imported torch or touched HIP), relay its stdout / stderr and return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: needed by RCCL on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    child = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def cpu_baseline(mixer, batch, steps):
    """The reference's CPU path restated on stock ATen ops (oracle/spectre_torch_cpu.py, pinned to the reference by
    tests/golden): fp32, same model / mixer / batch size / step definition, timed on this host's cores."""
    import torch
    from oracle import spectre_torch_cpu as T
    from spectre_vit.models.spectre.spectre import SpectreViT
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # 32 threads: measured on the GPU box's 256-core host the step is SLOWER with 128 threads (9.1 s) than with 8 (4.3 s in the
    # build container): the port is allocation- and memory-bound, like the reference's eager path
    threads = max(1, min(cores, int(os.environ.get("SPV_CPU_THREADS", "32"))))
    torch.set_num_threads(threads)
    if mixer not in ("fft", "permut"):
        mixer_cpu = "fft"  # the Haar mixers are not restated on ATen ops; same GEMM / row work as the FFT configuration
    else:
        mixer_cpu = mixer
    torch.manual_seed(42)
    cfg = dict(SMALL, dropout=0.0)
    sd = SpectreViT(**cfg, mixer=mixer_cpu).state_dict()
    st = T.TrainState(sd)
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(batch, 3, 32, 32, generator=g)
    labels = torch.randint(0, 100, (batch,), generator=g)
    st.step(img, labels, cfg["num_encoders"], cfg["patch_size"], mixer_cpu)  # warm-up
    times = []
    for _ in range(max(1, steps)):
        t0 = time.perf_counter()
        st.step(img, labels, cfg["num_encoders"], cfg["patch_size"], mixer_cpu)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    rec = dict(value=round(batch / dt, 2), unit="images/sec", cores=threads, kind="port",
               sample=f"torch-CPU port of the reference step (stock ATen ops + autograd + AdamW, fp32, {mixer_cpu} mixer, dropout 0): "
                      f"1 warm-up + {len(times)} timed steps of bs {batch}, median {dt:.2f} s/step, {threads} threads of {cores} host cores")
    if threads != 8 and cores >= 8 and steps > 0:
        # one run at n = 8 as well, for comparability with the survey's measurement of the reference itself
        # (25 img/s on 8 vCPUs, MHPermutMix); one timed step
        torch.set_num_threads(8)
        st.step(img, labels, cfg["num_encoders"], cfg["patch_size"], mixer_cpu)
        t0 = time.perf_counter()
        st.step(img, labels, cfg["num_encoders"], cfg["patch_size"], mixer_cpu)
        d8 = time.perf_counter() - t0
        rec["at_8_threads"] = dict(value=round(batch / d8, 2), unit="images/sec", cores=8, sample=f"1 warm-up + 1 timed step, {d8:.2f} s/step")
        torch.set_num_threads(threads)
    return rec


def base224_leg(dev, steps=5, warmup=2, batch=64):
    """BASELINE config 5, student side, on one GPU: Spectre-ViT-Base (the reference's SpectreViT defaults, spectre.py:162-171: E 768,
    12 layers, 12 heads, F 3072, dropout 0.1, HEAD mixer) at 224 / 16 -> 197 tokens, train step fwd + CE + bwd + AdamW in bf16, replayed
    from one HIP graph.  (The distillation loop adds a frozen teacher forward and the KD loss, train.py:298-361; the teacher is not
    available offline -- SURVEY 8c -- so the student step is what can be measured.)"""
    import torch
    from spectre_vit.graph import GraphedTrainStep
    from spectre_vit.loss import CrossEntropyLoss
    from spectre_vit.models.spectre.spectre import SpectreViT
    from spectre_vit.optim import FusedAdamW
    torch.manual_seed(0)
    m = SpectreViT(img_size=224, patch_size=16, in_channels=3, num_classes=100, embed_dim=768, num_encoders=12, num_heads=12,
                   hidden_dim=3072, dropout=0.1, mixer="permut").to(dev).train()
    img = torch.randn(batch, 3, 224, 224, device=dev)
    lab = torch.randint(0, 100, (batch,), device=dev)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, capturable=True, static_grads=True)
    step = GraphedTrainStep(m, opt, CrossEntropyLoss(), img, lab, autocast_dtype=torch.bfloat16, warmup=2)
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out = dict(value=round(batch / dt, 1), unit="images/sec", ms_per_step=round(dt * 1e3, 2), steps=steps, batch=batch, launch="graph",
                   workload="Spectre-ViT-Base student (E768 H12 F3072 L12, 224/16 -> 197 tokens, MHPermutMix), train step fwd+CE+bwd+AdamW, bf16",
                   final_loss=round(float(loss.item()), 4), parameters=sum(p.numel() for p in m.parameters()))
    finally:
        step.close()
    del step, opt, m
    torch.cuda.empty_cache()
    return out


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
    pats = {"gemm_tn_batch": ("gemm_tn_batch_wide", "gemm_tn_batch"), "splitk_reduce_batch": ("splitk_reduce_batch",),
            "gemm": ("gemm_nt_strip", "gemm_nt_kernel", "gemm_nt_glds"), "gemm_acc": ("gemm_nt_strip", "gemm_nt_kernel"), "gemm_tn": ("gemm_tn", "wgrad"),
            "gemm_pool_bwd": ("gemm_nt_strip", "gemm_nt_pool", "gemm_nt_kernel"), "fnet_ln_fwd": ("fnet",), "fnet_ln_bwd": ("fnet",),
            "fnet_mix": ("fnet",), "tail_fwd": ("tail_fwd",), "tail_bwd": ("tail_bwd",), "tail_bwd_up": ("tail_bwd",),
            "tail_ln_fwd": ("tail_fwd",), "tail_ln_bwd": ("tail_bwd",), "gather_fwd": ("gather_fwd",), "gather_bwd": ("gather_bwd",)}
    want = pats.get(roof["kernel"], (roof["kernel"],))
    # the candidate whose duration in the PMC run is closest to the live bracket
    cands = []
    for w in want:   # patterns in order of preference: the first one the collection holds decides (the strip kernel before the generic one)
        cands = [e for e in data.get("kernels", []) if e["kernel"].startswith(w) and e.get("hbm_bytes_corrected")]
        if cands:
            break
    if cands:
        best = min(cands, key=lambda e: abs((e.get("avg_us") or roof["avg_us"]) - roof["avg_us"]))
        roof["traffic"] = best["hbm_bytes_corrected"]
        roof["traffic_source"] = os.path.basename(files[-1])
        roof["traffic_kernel"] = best["kernel"][:80]


class Job:
    """model + optimizer + resident batch for one mixer; step() is the measured unit.

    launch = "graph"     one HIP graph per step (spectre_vit.graph.GraphedTrainStep): the single-GPU headline
             "dp_graph"  a data-parallel rank: graph A (forward + loss + backward) -> ONE all-reduce of the flat gradient buffer ->
                         graph B (AdamW) (spectre_vit.graph.GraphedDPStep)
             "eager"     every kernel launched by the host; gradients exchanged bucket by bucket, overlapped with the backward
    eager_step() runs the SAME launch sequence as step() with the host issuing every launch (the roofline pass brackets it)."""

    def __init__(self, args, mixer, dev, rank, launch="eager", stand_in=False, force_collective=False):
        import torch
        from spectre_vit.dp import GradReducer, broadcast_module
        self.torch = torch
        self.launch = launch
        torch.manual_seed(42)
        if stand_in:  # CPU rehearsal of the launcher / collectives only (no GPU in this process): a small stock model
            model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3 * 32 * 32, 64), torch.nn.GELU(), torch.nn.Linear(64, 100))
        elif args.model == "vit":
            from spectre_vit.models.vit.vit import ViT
            model = ViT(**SMALL).to(dev)
        else:
            from spectre_vit.models.spectre.spectre import SpectreViT
            model = SpectreViT(**SMALL, mixer=mixer).to(dev)
        broadcast_module(model)
        model.train()
        self.model = model
        torch.manual_seed(1234 + rank)  # per-rank dropout / data streams
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.img = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
        labels = torch.randint(0, 100, (args.batch,), generator=g).to(torch.uint8).to(dev)  # uint8 as train.py:218
        self.labels = labels.long()
        self.use_bf16 = args.dtype == "bf16" and not stand_in
        self.dev = dev
        self.param_bytes = sum(p.numel() for p in model.parameters()) * 4
        self.timer = None
        self.gstep = None
        if stand_in:
            self.crit = torch.nn.CrossEntropyLoss()
        else:  # nn.CrossEntropyLoss() semantics on one launch each way (spectre_vit/loss.py; parity-tested against torch's)
            from spectre_vit.loss import CrossEntropyLoss
            self.crit = CrossEntropyLoss()
        if launch in ("graph", "dp_graph"):
            from spectre_vit.graph import GraphedDPStep, GraphedTrainStep
            from spectre_vit.optim import FusedAdamW
            self.opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, capturable=True, static_grads=True)
            cls = GraphedDPStep if launch == "dp_graph" else GraphedTrainStep
            self.gstep = cls(model, self.opt, self.crit, self.img, self.labels, autocast_dtype=torch.bfloat16 if self.use_bf16 else None,
                             force_collective=force_collective)
            self.reducer = self.gstep.reducer
            return
        self.reducer = GradReducer(model, always=not stand_in)  # fixed gradient addresses (one-launch optimizer's pointer table)
        if stand_in or os.environ.get("SPV_TORCH_ADAMW") == "1":  # A/B aid: torch's own fused AdamW
            self.opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, fused=not stand_in)
        else:  # the same update rule on one launch (spectre_vit/optim.py; parity-tested against torch.optim.AdamW)
            from spectre_vit.optim import FusedAdamW
            self.opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, static_grads=True)
        self.one = torch.ones((), dtype=torch.float32, device=dev)

    def step(self):
        if self.gstep is not None:
            return self.gstep()
        return self.eager_step()

    def eager_step(self):
        torch = self.torch
        if self.gstep is not None:   # the graph's own launch sequence (held / batched weight gradients included), issued by the host
            with self.gstep._hold():
                return self.gstep._eager_step()[0]
        self.reducer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_bf16):
            out = self.model(self.img)
        loss = self.crit(out, self.labels)
        loss.backward(self.one)   # a kept 1.0: backward() without it fills a fresh one every step (a launch)
        self.reducer.finish()
        if self.timer is not None and isinstance(self.opt, torch.optim.AdamW):  # fused AdamW reads p, g, m, v and writes p, m, v
            self.timer.bracket("torch:adamw_fused", (7 * self.param_bytes,), self.opt.step)
        else:
            self.opt.step()  # (FusedAdamW's launch is bracketed by the C-ABI hook like every other kernel)
        return loss

    def close(self):
        if self.gstep is not None:
            self.gstep.close()
            self.gstep = None


def timed_region(job, sync, steps, warmup, eager=False):
    fn = job.eager_step if eager else job.step
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = fn()
    sync()
    return time.perf_counter() - t0, loss


def side_leg(args, mixer, dev, sync, launch, steps, warmup, force_collective=False):
    """another configuration timed in the same process, in the SAME launch mode as the headline unless said otherwise"""
    import torch
    job = Job(args, mixer, dev, 0, launch=launch, force_collective=force_collective)
    try:
        el, loss = timed_region(job, sync, steps, warmup)
        out = dict(value=round(args.batch * steps / el, 1), unit="images/sec", ms_per_step=round(el / steps * 1e3, 3), steps=steps,
                   launch=launch, final_loss=round(float(loss.item()), 4))
    finally:
        job.close()
        del job
        torch.cuda.empty_cache()
    return out


def script_leg(args, mixer, dev, steps, warmup):
    """The reference script's own loop on the mirror package, verbatim (spectre_vit/repl/train.py:216-238): fp16 autocast (served by the
    bf16 kernels), GradScaler("cuda"), torch.optim.AdamW, zero_grad(set_to_none=True), loss.item() every step -- no gradient sinks,
    no held weight gradients, no graph: what a user gets who changes nothing but the import path."""
    import torch
    import warnings
    from spectre_vit.models.spectre.spectre import SpectreViT
    torch.manual_seed(42)
    model = SpectreViT(**SMALL, mixer=mixer).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(1234)
    img = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
    label = torch.randint(0, 100, (args.batch,), generator=g).type(torch.uint8).to(dev)
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.AdamW(model.parameters(), betas=(0.9, 0.999), lr=1e-3, weight_decay=0.01)
    scaler = torch.amp.GradScaler("cuda")

    def one():
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            y_pred = model(img)
        loss = criterion(y_pred, label)
        optimizer.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
        return loss.item()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = one()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    del model, optimizer
    torch.cuda.empty_cache()
    return dict(value=round(args.batch * steps / el, 1), unit="images/sec", ms_per_step=round(el / steps * 1e3, 3), steps=steps,
                launch="the reference loop verbatim (train.py:216-238): fp16 autocast -> bf16 kernels, GradScaler, torch.optim.AdamW, "
                       "zero_grad(set_to_none=True), loss.item() per step; eager, no sinks, no batching",
                final_loss=round(float(last), 4))


def dp_sequence_child(args, steps, warmup):
    cmd = [sys.executable, os.path.abspath(__file__), "--dp-sequence", "--steps", str(steps), "--warmup", str(warmup), "--mixer", args.mixer,
           "--batch", str(args.batch), "--dtype", args.dtype, "--no-cpu-baseline", "--no-roofline", "--no-every-row", "--no-script-leg",
           "--no-base224", "--variants", "none"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child exited with {r.returncode}: {r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else 'no output'}"}
        c = json.loads(line[-1])
        return dict(value=c["value"], unit="images/sec", ms_per_step=c["ms_per_step"], steps=steps, launch="dp_graph", final_loss=c["final_loss"],
                    collective=("torch.distributed all_reduce in a one-rank RCCL group" if c.get("rccl_ranks") == 1 else "skipped (no process group)"),
                    gradient_exchange=c.get("gradient_exchange"), process="child process (python bench.py --dp-sequence)")
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def one_rank_group(dev):
    """a one-rank RCCL process group in this process (single-GPU --dp-sequence leg: the collective call of a rank is issued for real)"""
    import torch.distributed as dist
    if dist.is_initialized():
        return True
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
        return True
    except Exception as exc:  # no RCCL in this environment: the leg still runs the two graphs, with the collective skipped
        sys.stderr.write(f"bench.py: one-rank RCCL group unavailable ({exc}); dp_sequence runs without the collective call\n")
        return False


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(self_launch(args, argv))  # before torch / HIP are touched in this process

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL on this driver), also under an external launcher
    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL prints a five-line version banner to
    # stdout when a communicator is created), so file descriptor 1 points at stderr until the line itself is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    # SPV_BENCH_REHEARSAL=1: all ranks on device 0 over gloo -- lets the multi-rank control flow (launcher, collectives, barriers,
    # rank-0-only sections) be exercised on a one-GPU box; with no GPU at all a small stock model stands in on the CPU, so that
    # the launcher and the collectives can be tested in the build container.  Never used for reported numbers.
    rehearsal = os.environ.get("SPV_BENCH_REHEARSAL") == "1"
    stand_in = rehearsal and not torch.cuda.is_available()
    if rehearsal:
        local_rank = 0
    if stand_in:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # launch mode of the headline.  One GPU: the step replayed from one HIP graph.  N GPUs: a rank replays two graphs around one
    # all-reduce (FFT / DWT models: 13 MB of gradients; the exposed collective costs less than the eager path's host time); the HEAD
    # mixer's 80 MB keep the eager step whose bucket all-reduces overlap the backward.
    graphable = args.model == "spectre" and not stand_in and not args.eager
    if world > 1 or args.dp_sequence:
        launch = "dp_graph" if graphable and (args.mixer != "permut" or args.dp_graph) else "eager"
    else:
        launch = "graph" if graphable else "eager"
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            backend = "gloo"
            dist.init_process_group("gloo")
        else:
            if launch == "eager":
                # the layer GEMMs leave 16 CUs to RCCL (spectre_vit.dp.RESERVED_CUS); keep RCCL inside them.  One step exchanges
                # 80 MB of fp32 gradients, which 16 channels move well inside the backward.  An explicit NCCL_MAX_NCHANNELS wins.
                os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
            backend = "nccl"
            dist.init_process_group("nccl", device_id=dev)
    force_collective = False
    if world == 1 and args.dp_sequence and launch == "dp_graph":
        force_collective = one_rank_group(dev)
        backend = "nccl (one-rank group)" if force_collective else None
    if args.model == "vit" or stand_in:
        args.no_cpu_baseline = True  # the CPU leg times the Spectre port
    if stand_in:
        args.no_roofline = True

    def sync():
        if not stand_in:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not stand_in:
            torch.cuda.synchronize()

    job, build_error = None, None
    try:
        job = Job(args, args.mixer, dev, rank, launch=launch, stand_in=stand_in, force_collective=force_collective)
    except Exception as exc:   # noqa: BLE001 -- reported in the line; only the multi-rank graph step has a fallback
        if not (world > 1 and launch == "dp_graph"):
            raise
        build_error = f"{type(exc).__name__}: {exc}"
    if world > 1 and launch == "dp_graph":
        # The two-graph rank step has never run over real xGMI links when this line is written (one GPU per development box): if its
        # construction fails on ANY rank -- the capture next to a live RCCL communicator is the untested part -- EVERY rank falls back
        # to the eager step with the overlapped bucket exchange, and the line says so.
        ok = torch.tensor([0 if job is None else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if job is not None:
                job.close()
            job = None
            torch.cuda.empty_cache()
            launch = "eager"
            if rank == 0:
                sys.stderr.write(f"bench.py: GraphedDPStep could not be built on every rank ({build_error or 'another rank failed'}); "
                                 "falling back to the eager data-parallel step\n")
            job = Job(args, args.mixer, dev, rank, launch=launch, stand_in=stand_in)
    elapsed, loss = timed_region(job, sync, args.steps, args.warmup)
    final_loss = float(loss.item())
    # roofline pass: the SAME launch sequence as the timed region (in the graph modes: the captured sequence issued by the host, with the
    # layer weight gradients held and batched exactly as in the graph), same process, right after it, with HIP events recorded on the
    # launch stream around EVERY C-ABI launch.  Kept out of the headline timing because ~70 event pairs per step perturb it.  Every rank
    # runs these steps (each one contains the gradient exchange); only rank 0 records events.
    timer = None
    rsteps = min(args.steps, 10)
    eager_fig = None
    post_error = None
    if not args.no_roofline:
        from spectre_vit import hip_ops
        hip_ops.TIME_HELD = launch != "eager"
        try:
            if rank == 0:
                timer = hip_ops.KernelTimer()
                hip_ops.set_kernel_timer(timer)
                job.timer = timer
            for _ in range(rsteps):
                job.eager_step()
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001 -- the headline is measured already: a failure here must not cost the line (one process only:
            if world > 1:         # with several ranks the others are inside the same collective and an error has to surface)
                raise
            post_error = f"roofline pass: {type(exc).__name__}: {str(exc)[:300]}"
            timer = None
        finally:
            hip_ops.set_kernel_timer(None)
            hip_ops.TIME_HELD = False
            job.timer = None
    if launch != "eager" and world == 1 and post_error is None:
        # the same launch sequence with the host issuing every launch: what the graph removes
        try:
            esteps = max(5, min(args.steps, 20))
            eel, eloss = timed_region(job, sync, esteps, 2, eager=True)
            eager_fig = {"value": round(args.batch * esteps / eel, 1), "ms_per_step": round(eel / esteps * 1e3, 3), "steps": esteps,
                         "final_loss": round(float(eloss.item()), 4)}
        except Exception as exc:  # noqa: BLE001
            post_error = f"eager leg: {type(exc).__name__}: {str(exc)[:300]}"
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    launch_text = {"graph": "one HIP graph per step (spectre_vit.graph.GraphedTrainStep); the same launch sequence issued by the host under \"eager\"",
                   "dp_graph": "per rank: HIP graph A (forward + loss + backward, batched weight gradients) -> one all-reduce of the flat "
                               "gradient buffer -> HIP graph B (AdamW) (spectre_vit.graph.GraphedDPStep)",
                   "eager": "every launch issued by the host; gradient buckets all-reduced during the backward pass (spectre_vit.dp.GradReducer)"}[launch]
    rec = None
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
            "dtype": "bf16" if job.use_bf16 else "f32",
            "data": "synthetic CIFAR-shaped randn images / randint labels resident in HBM, random-init weights (seed 42)",
            "config": {"workload": "stand-in stock MLP on the CPU (launcher rehearsal only)" if stand_in else
                                   (f"Spectre-ViT-Small (E512 H16 F768 L4 P4 N65, 100 classes, dropout 0.001), {args.mixer} mixer, "
                                    if args.model == "spectre" else
                                    "baseline ViT-Small (E512 H16 F768 L4 P4 N65, MHSA with the reference's batch_first=False axis), ")
                                   + f"train step fwd+CE+bwd+AdamW, bs {args.batch}/GPU",
                       **({"rehearsal": ("launcher / collective control flow on the CPU with a stock stand-in model: not a measurement"
                                         if stand_in else
                                         "all ranks on one device over gloo: control-flow check, not a measurement")} if rehearsal else {}),
                       "mixer": args.mixer if args.model == "spectre" else "attention", "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "launch": launch_text},
            **({"dp_graph_fallback": build_error or "another rank failed to build the two-graph step"}
               if (world > 1 and launch == "eager" and graphable and (args.mixer != "permut" or args.dp_graph)) else {}),
            "rccl_ranks": world if backend == "nccl" else (1 if force_collective else 0),
            "backend": backend or "none (single process)",
            "final_loss": round(final_loss, 4),
        }
        if launch != "eager" and not stand_in:
            rec["gradient_exchange"] = ({"bytes": int(job.reducer.flat.numel() * 4), "calls_per_step": 1,
                                         "form": "one all-reduce (mean) of the flat fp32 gradient buffer between graph A and graph B"}
                                        if launch == "dp_graph" else None)
        if eager_fig is not None:
            rec["eager"] = eager_fig
        if post_error is not None:
            rec["post_measurement_error"] = post_error
        if timer is not None:
            rec["roofline"] = timer.roofline()
            kern = timer.summary()
            for d in kern:  # per-step figures
                d["launches_per_step"] = round(d["launches"] / rsteps, 2)
                d["ms_per_step"] = round(d.pop("total_ms") / rsteps, 4)
                d.pop("launches")
            rec["kernels"] = kern
            covered = sum(d["ms_per_step"] for d in kern)
            rec["kernels_coverage"] = {"bracketed_ms_per_step": round(covered, 3), "frac_of_step": round(covered / ms, 3),
                                       "note": "sum of the raw HIP-event brackets (one per C-ABI launch, the timed step's own launch "
                                               "sequence issued by the host; every bracket includes its event pair, ~"
                                               f"{timer.overhead_s * 1e6:.1f} us, so the sum exceeds the replayed step) / ms_per_step"}
            if rec["roofline"]:
                rec["roofline"]["launches"] = round(rec["roofline"]["launches"] / rsteps, 2)
                attach_pmc_traffic(rec["roofline"], args.mixer)
    job.close()

    # single GPU, default workload: the other configurations SURVEY 8d asks for, each in the headline's launch mode
    if rank == 0 and world == 1 and not stand_in and args.model == "spectre" and launch == "graph":
        from spectre_vit import hip_ops
        del job
        torch.cuda.empty_cache()
        vsteps, vwarm = max(5, min(args.steps, 20)), min(args.warmup, 5)
        # Exact dead-row elimination, stated with its counterpart: SpectreViT reads only the CLS row of the stack's output (reference
        # spectre.py:198) and the last layer's feed-forward half works row by row, so by default it runs at the CLS rows only -- same
        # logits, loss and gradients (tests/test_gpu_bench_shapes.py checks both forms against the oracle).  The same graph-replayed
        # step with every row computed, as the reference does, is measured beside it.
        rec["config"]["last_layer_feed_forward"] = ("CLS rows only -- exact dead-row elimination: the model reads only the CLS row of the stack's output (reference spectre.py:198), so "
                                                  "the last layer's row-wise half (and its mixer's row 0) is computed for that row alone; logits, loss and EVERY parameter "
                                                  "gradient are those of the every-row computation (tests/test_gpu_bench_shapes.py runs both forms against the oracle); "
                                                  "the every-row step is timed beside it under \"every_row_of_last_layer\" (SPV_FULL_LAST_LAYER=1 selects it)"
                                                  if hip_ops.LAST_LAYER_CLS_ONLY else "every row")
        def guarded(fn, *a, **kw):
            """a side leg must never cost the headline its line: an exception becomes the leg's record"""
            try:
                return fn(*a, **kw)
            except Exception as exc:  # noqa: BLE001
                torch.cuda.empty_cache()
                return {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}

        if hip_ops.LAST_LAYER_CLS_ONLY and not args.no_every_row:
            hip_ops.LAST_LAYER_CLS_ONLY = False
            try:
                rec["every_row_of_last_layer"] = guarded(side_leg, args, args.mixer, dev, sync, "graph", vsteps, vwarm)
            finally:
                hip_ops.LAST_LAYER_CLS_ONLY = True
        variants = args.variants
        if variants is None:
            variants = "permut,dwt_embed" if args.mixer == "fft" else "none"
        if variants != "none":
            rec["variants"] = {mx: guarded(side_leg, args, mx, dev, sync, "graph", vsteps, vwarm) for mx in variants.split(",") if mx}
        if not args.no_dp_sequence:
            # what ONE rank of the N-GPU job executes, measured on this GPU: two graphs around the (one-rank) collective call.  Run as
            # a CHILD process (`bench.py --dp-sequence`): creating an RCCL communicator is the one thing in this file that has never
            # been exercised by the driver, and a native crash there must not take the headline line with it.
            rec["dp_sequence"] = dp_sequence_child(args, vsteps, vwarm)
        if not args.no_script_leg:
            rec["as_script"] = guarded(script_leg, args, args.mixer, dev, vsteps, vwarm)
        if not args.no_base224 and args.mixer == "fft" and args.batch == 512:
            rec["base224_student"] = guarded(base224_leg, dev)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                rec["cpu_baseline"] = cpu_baseline(args.mixer, args.batch, args.cpu_steps)
            except Exception as exc:  # noqa: BLE001
                rec["cpu_baseline"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(rec), flush=True)
        os.dup2(2, 1)   # (whatever is printed while the process winds down goes to stderr again)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
