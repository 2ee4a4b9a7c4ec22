"""2-rank gloo (CPU) test of the data-parallel gradient exchange (spectre_vit/dp.py): after GradReducer.finish() every
rank holds the mean of the shard gradients, which equals the single-process gradient of the concatenated batch for a
mean loss.  The reducer is model agnostic, so a small stock model stands in (our modules need a GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.GELU(), torch.nn.LayerNorm(32), torch.nn.Linear(32, 5))


def _worker(rank, world, port, bucket_mb, outdir, overlap=True):
    sys.path.insert(0, PKG)
    from spectre_vit.dp import GradReducer, broadcast_module
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    if rank == 1:  # rank 1 starts from different weights/buffers: broadcast must fix that
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    broadcast_module(m)
    red = GradReducer(m, bucket_mb=bucket_mb, overlap=overlap)
    if not overlap:
        # the exchange of spectre_vit.graph.GraphedDPStep: ONE flat buffer behind all buckets, nothing launched by the hooks
        assert red.flat is not None and all(b["flat"].untyped_storage().data_ptr() == red.flat.untyped_storage().data_ptr() for b in red.buckets)
    g = torch.Generator().manual_seed(99)
    x, y = torch.randn(8, 12, generator=g), torch.randint(0, 5, (8,), generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):  # two steps: buckets are re-armed by zero_grad
        red.zero_grad()
        torch.nn.functional.cross_entropy(m(xs), ys).backward()
        if not overlap:
            assert all(b["handle"] is None for b in red.buckets)   # no collective was started during the backward pass
        red.finish()   # overlap=False: allreduce_flat() -- one call over the flat buffer
    torch.save((rank, [p.grad.clone() for p in m.parameters()], [p.detach().clone() for p in m.parameters()]),
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])  # bucket all-reduces during the backward pass / one all-reduce of the flat buffer after it
@pytest.mark.parametrize("bucket_mb", [32.0, 0.0005])  # one bucket / several tiny buckets
def test_grad_reducer_two_ranks(bucket_mb, overlap, tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, str(tmp_path), overlap)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(2)]
    m = _model()
    g = torch.Generator().manual_seed(99)
    x, y = torch.randn(8, 12, generator=g), torch.randint(0, 5, (8,), generator=g)
    torch.nn.functional.cross_entropy(m(x), y).backward()
    for (_, g0, w0), (_, g1, w1) in [(res[0], res[1])]:
        for a, b, p in zip(g0, g1, m.parameters()):
            torch.testing.assert_close(a, b, rtol=0, atol=0)          # ranks agree bit for bit
            torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-6)  # == full-batch gradient
        for a, b, p in zip(w0, w1, m.parameters()):
            torch.testing.assert_close(a, b, rtol=0, atol=0)
            torch.testing.assert_close(a, p.detach(), rtol=0, atol=0)  # broadcast restored rank 0's weights
