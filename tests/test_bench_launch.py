"""bench.py --gpus N must start its own ranks (the driver runs `python bench.py --gpus 8`, no external launcher).  Here, without
a GPU, SPV_BENCH_REHEARSAL=1 swaps the model for a small stock one on the CPU over gloo: what is tested is the launcher
(child torch.distributed.run spawned before torch is imported in the parent), the rendezvous, GradReducer's all-reduce inside
the step, the max-over-ranks timing and the single JSON line of rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, *args):
    env = dict(os.environ, SPV_BENCH_REHEARSAL="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=600)


def test_bench_self_launches_two_ranks():
    r = _run(None, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]   # nothing but the line on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 32
    assert rec["config"]["parallelism"] == "dp2" and rec["backend"] == "gloo" and "rehearsal" in rec["config"]
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0 and rec["higher_is_better"] is True


def test_bench_rejects_mismatched_world_size():
    r = _run({"WORLD_SIZE": "1", "RANK": "0"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_parent_does_not_import_torch_before_spawning():
    """the parent of a self-launched run must not initialise HIP: it may not even import torch before the child exists."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert "\nimport torch" not in head and "\nfrom torch" not in head
    body = src[src.index("def main():"):]
    assert body.index("self_launch(args, argv)") < body.index("import torch")
