"""`not gpu`: the N > 1 path on world_size-2 gloo (CPU).  The data-parallel plumbing (flat buckets, broadcast,
single all-reduce, shard ranges, max-over-ranks timing) is device-agnostic host logic; per-rank gradients come
from the CPU oracle so the expected result -- the mean of the per-shard gradients -- is known exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from dynamorph_amd import dist as D
    from oracle import vqvae_oracle as O
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world

    # different init per rank -> broadcast must make the replicas identical
    torch.manual_seed(100 + rank)
    model = O.OracleVQVAE()
    fp = D.FlatParams(model.parameters())
    D.broadcast_(fp.flat, list(model.buffers()))
    flat0 = fp.flat.clone()

    # per-rank shard of a common global batch
    g = torch.Generator().manual_seed(7)
    xg = torch.randn(4 * world, 2, 128, 128, generator=g)
    lo, hi = D.shard_range(xg.shape[0], rank, world)
    _, ld = model(xg[lo:hi])
    ld["total_loss"].backward()
    for p in fp.params:
        fp.gview(p).copy_(p.grad)
    local = fp.grad.clone()
    D.allreduce_mean_(fp.grad)
    # autograd-path helper (VQ_VAE_z32 + torch optimizer): .grad tensors averaged through one bucket
    plist = [p for p in model.parameters() if p.grad is not None]
    for p in plist:
        p.grad = p.grad.clone()
    D.allreduce_grads_(plist)
    helper = torch.cat([p.grad.reshape(-1) for p in fp.params])
    # inference hand-over: contiguous shards gathered on the host in rank order
    n_items = 7
    a, b_ = D.shard_range(n_items, rank, world)
    full = np.arange(n_items * 3, dtype=np.float32).reshape(n_items, 3)
    got = D.gather_shards((full[a:b_], 2 * full[a:b_]))
    assert np.array_equal(got[0], full) and np.array_equal(got[1], 2 * full)
    t = D.max_over_ranks(1.0 + rank)
    torch.save({"flat0": flat0, "local": local, "mean": fp.grad.clone(), "helper": helper, "tmax": t, "range": (lo, hi)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_plumbing_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(res[0]["flat0"], res[1]["flat0"])                 # same replica after broadcast
    assert not torch.equal(res[0]["local"], res[1]["local"])             # different shards, different grads
    expect = (res[0]["local"] + res[1]["local"]) / world
    for r in res:
        assert torch.allclose(r["mean"], expect, rtol=0, atol=1e-7)      # ONE collective gives the mean of per-shard grads
        assert torch.allclose(r["helper"], expect, rtol=0, atol=1e-7)    # allreduce_grads_ (autograd path) gives the same
        assert r["tmax"] == float(world)                                 # max over ranks (bench timing contract)
    assert res[0]["range"] == (0, 4) and res[1]["range"] == (4, 8)
