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
    got = D.gather_shards((full[a:b_], 2 * full[a:b_]))          # to rank 0 only: nothing is replicated
    if rank == 0:
        assert np.array_equal(got[0], full) and np.array_equal(got[1], 2 * full)
    else:
        assert got is None
    one = D.gather_shards(full[a:b_].astype(np.float64), dst=world - 1)
    assert (np.array_equal(one, full) and one.dtype == np.float64) if rank == world - 1 else one is None
    few = D.gather_shards(full[:1] if rank == world - 1 else full[:0])      # empty shards everywhere but on the last rank
    assert np.array_equal(few, full[:1]) if rank == 0 else few is None
    t = D.max_over_ranks(1.0 + rank)
    # the helpers bench.py's N > 1 line and tests/test_gpu_dist.py's RCCL evidence are built from
    ev_same = D.collective_evidence(fp.flat)                     # replicas identical after the broadcast
    drift = fp.flat.clone()
    if rank == world - 1:
        drift[5] += 1e-7
    ev_drift = D.collective_evidence(drift)                      # one rank one ulp-ish off: every rank must see it
    assert ev_same == {"backend": "gloo", "world": world, "nccl_version": None, "replicas_bit_equal": True}
    assert ev_drift["replicas_bit_equal"] is False
    assert D.gather_rank_values(2.5 * rank) == [2.5 * k for k in range(world)]
    assert abs(sum(D.gather_rank_values(D.shard_weight(7, rank, world))) - world) < 1e-12
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


# ------------------------------------------------------------------------------------------------------------------
# train() itself at world size 2.  The product modules have no CPU path, so the module that stands in here is the CPU
# oracle (same constructor / forward contract); what is under test is the LOOP: one seed for split and shuffles, contiguous
# per-rank shards of every global batch, the weighted single gradient exchange, the collective early-stopping decision
# and the rank-0 checkpoint.
LR = 1e-3
# (samples, batch, epochs, validation ratio).  The second case has a 1-sample last training batch and a 1-sample
# validation set: rank 1's shard of both is EMPTY, so it joins the gradient exchange with zeros and the epoch-loss
# exchange with no keys of its own (the vector must still have rank 0's length on every rank).
# The third runs FOUR ranks (ragged last batch of 5 over 4 ranks, validation batches smaller than the world).
TRAIN_CASES = [(22, 6, 2, 0.2), (7, 5, 1, 0.15), (26, 8, 1, 0.2)]
TRAIN_WORLDS = [2, 2, 4]


def _train_worker(rank, world, port, out_dir, case):
    n_samples, batch, epochs, ratio = case
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from dynamorph_amd import dist as D
    from dynamorph_amd.train import train
    from oracle import vqvae_oracle as O
    D.init_from_env(backend="gloo")
    torch.manual_seed(500 + rank)                       # different replicas: train() must broadcast rank 0's
    np.random.seed(11 + rank)                           # different host generators: the split must still agree
    model = O.OracleVQVAE()
    data = torch.utils.data.TensorDataset(torch.randn(n_samples, 2, 128, 128, generator=torch.Generator().manual_seed(3)))
    rows = {}

    class Scalars:
        def add_scalar(self, key, value, epoch):
            rows.setdefault(key, []).append(float(value))
    train(model, data, os.path.join(out_dir, "run"), n_epochs=epochs, lr=LR, batch_size=batch, device="cpu",
          transform=None, val_split_ratio=ratio, patience=5, writer=Scalars(), fused=False)
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "rows": rows},
               os.path.join(out_dir, f"train_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", TRAIN_CASES, ids=["even-shards", "empty-shards", "four-ranks"])
def test_train_loop_world2_shards_batches_and_checkpoints_on_rank0(tmp_path, case):
    from dynamorph_amd import dist as D
    from oracle import vqvae_oracle as O
    n_samples, batch, epochs, ratio = case
    world = TRAIN_WORLDS[TRAIN_CASES.index(case)]
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"train_rank{r}.pt")) for r in range(world)]
    # identical replicas after training, and rank 0 alone reported the epoch scalars
    for r in range(1, world):
        for a, b in zip(res[0]["params"], res[r]["params"]):
            assert torch.equal(a, b)
        assert not res[r]["rows"]
    assert len(res[0]["rows"]["Loss/total_loss"]) == epochs
    assert len(res[0]["rows"]["Val loss/total_loss"]) == epochs
    ck = torch.load(os.path.join(tmp_path, "run", "model.pt"))
    assert not os.path.exists(os.path.join(tmp_path, "run", "model.pt.tmp"))

    # single-process emulation of the same schedule: per global batch, the shards' gradients weighted by their sizes
    torch.manual_seed(500)                              # rank 0's replica is the one that was broadcast
    np.random.seed(11)
    model = O.OracleVQVAE()
    opt = O.make_adam(model, LR)
    data = torch.randn(n_samples, 2, 128, 128, generator=torch.Generator().manual_seed(3))
    seed = int(np.random.randint(0, 2 ** 31 - 1))       # what train() drew on rank 0
    order = np.random.RandomState(seed)
    split = int(np.floor(ratio * n_samples))
    start = int(order.randint(0, n_samples - split))
    ids = list(range(n_samples))
    train_ids, val_ids = ids[:start] + ids[start + split:], ids[start:start + split]
    if case == TRAIN_CASES[1]:
        assert len(val_ids) == 1 and len(train_ids) % batch == 1         # the empty-shard situations really occur
    params = [p for p in model.parameters() if p.requires_grad]
    for epoch in range(epochs):
        for s0 in range(0, len(train_ids), batch):
            gb = train_ids[s0:s0 + batch]
            acc = [torch.zeros_like(p) for p in params]
            for r in range(world):
                lo, hi = D.shard_range(len(gb), r, world)
                if hi > lo:
                    model.zero_grad()
                    model(data[gb[lo:hi]])[1]["total_loss"].backward()
                    for a, p in zip(acc, params):
                        a += p.grad * ((hi - lo) / len(gb))
            for a, p in zip(acc, params):
                p.grad = a
            opt.step()
        # the exchanged epoch value: sample-weighted mean over every rank's shards of the validation batches
        tot = cnt = 0.0
        for s0 in range(0, len(val_ids), batch):
            gb = val_ids[s0:s0 + batch]
            for r in range(world):
                lo, hi = D.shard_range(len(gb), r, world)
                if hi > lo:
                    with torch.no_grad():
                        tot += float(model(data[gb[lo:hi]])[1]["total_loss"]) * (hi - lo)
                    cnt += hi - lo
        assert abs(res[0]["rows"]["Val loss/total_loss"][epoch] - tot / cnt) <= 2e-5 * max(1.0, abs(tot / cnt))
    # (biases of convolutions that feed a train-mode BatchNorm have an identically zero gradient; autograd returns
    # +-1e-9 of rounding noise there, which Adam turns into +-lr steps whose signs depend on the summation order)
    noise = ("enc.1.bias", "enc.4.bias", "enc.7.bias", "enc.10.bias", ".1.bias", ".4.bias")
    for (name, w_), g in zip(model.named_parameters(), res[0]["params"]):
        if name.endswith(noise) and "enc" in name:
            assert (g - w_).abs().max() <= 2.5 * LR * epochs * 4
            continue
        assert torch.allclose(g, w_.detach(), rtol=0, atol=2e-6), (name, float((g - w_).abs().max()))
    # the checkpoint holds one of the epochs' weights (the best validation loss), written by rank 0
    assert set(ck.keys()) == set(model.state_dict().keys())


# (the one-process draws, aggregation and checkpoint of train() are held against the reference's own run in
# tests/test_train_loop_golden.py)
