"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-device code at all; what the path needs is small:
  * inference latents (process_VAE): patches are independent -> contiguous shards, NO collective;
  * training: ONE all-reduce per step over a single flat fp32 gradient bucket (24 058 floats = 96 KB for
    the default model, latency-bound), then the identical fused Adam on every rank.  Every loss is a mean
    over the local batch (vq_vae.py:74-75, 322), so the mean of per-rank gradients is the gradient of the
    global-batch mean loss given rank-local BatchNorm statistics (standard DDP semantics).

Everything here is host logic on torch tensors of any device, so it is covered by gloo tests on the CPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world, local_rank); a single-process run needs no group and returns (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DM_DIST_BACKEND=gloo: rehearse the multi-process path where RCCL cannot run (several ranks on one GPU)
            backend = os.environ.get("DM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f"rank {rank}: RCCL needs one GPU per rank, local rank {local_rank} has none among the "
                                 f"{torch.cuda.device_count()} visible (DM_DIST_BACKEND=gloo rehearses on fewer)")
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def get_rank(group=None):
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def barrier(group=None):
    if world_size(group) > 1:
        dist.barrier(group=group)


def broadcast_object(obj, src=0, group=None):
    """A small picklable object from `src` to every rank (seeds, split points); the identity in a single process."""
    if world_size(group) == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def allreduce_sum_host(values, device=None, group=None):
    """Element-wise sum over ranks of a short list of Python floats (epoch loss sums: one collective per epoch)."""
    if world_size(group) == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n independent units for `rank`; sizes differ by at most one and the
    shards tile [0, n) in rank order (so gathering in rank order restores file-path order)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_weight(n, rank, world):
    """Weight of `rank`'s gradient in the data-parallel mean for a global batch of n samples cut by shard_range: every
    loss is a mean over the LOCAL shard, so n_local * world / n makes mean-over-ranks the global-batch mean (1 for even
    shards, 0 for an empty one; the weights of all ranks sum to `world`)."""
    lo, hi = shard_range(n, rank, world)
    return (hi - lo) * world / n


def gather_rank_values(value, device=None, group=None):
    """[value on rank 0, ..., value on rank world-1] on every rank (bench: each rank's own wall time of the timed region)."""
    if world_size(group) == 1:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world_size(group))]
    dist.all_gather(parts, mine, group=group)
    return [float(t.item()) for t in parts]


class FlatParams:
    """All trainable tensors of a module as views of ONE flat buffer, gradients as views of a second one."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, device=dev, dtype=dt)
        self.grad = torch.zeros(n, device=dev, dtype=dt)
        self._gview = {}
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)        # parameter storage now lives in the flat buffer
            self._gview[id(p)] = self.grad[off:off + k].view_as(p)
            off += k

    def gview(self, p):
        return self._gview[id(p)]

    def expose_grads(self):
        for p in self.params:
            p.grad = self._gview[id(p)]


def broadcast_(flat, buffers=(), src=0, group=None):
    """Same replica on every rank before the first step (parameters + BatchNorm buffers)."""
    if world_size(group) == 1:
        return
    dist.broadcast(flat, src=src, group=group)
    for b in buffers:
        dist.broadcast(b, src=src, group=group)


def allreduce_mean_(t, group=None):
    """In-place mean over ranks of one flat bucket: a single collective per step."""
    w = world_size(group)
    if w == 1:
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.mul_(1.0 / w)
    return t


def allreduce_grads_(params, group=None):
    """Mean over ranks of the .grad of `params` (autograd path: VQ_VAE_z32 with a torch optimizer) -- packed into one
    flat bucket, ONE collective, scattered back.  No-op in a single process."""
    w = world_size(group)
    if w == 1:
        return
    gs = [p.grad for p in params if p.grad is not None]
    if not gs:
        return
    flat = torch.cat([g.reshape(-1) for g in gs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / w)
    off = 0
    for g in gs:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


_host_groups = {}


def host_group(group=None):
    """A gloo group over the same ranks for hand-overs of HOST arrays: under the nccl (RCCL) backend every collective
    moves device memory, so shipping results that already live in host memory through it would be a host -> HBM ->
    xGMI -> HBM -> host round trip.  The default group itself when it is gloo already."""
    if dist.get_backend(group) == "gloo":
        return group
    key = id(group)
    if key not in _host_groups:
        ranks = None if group is None else dist.get_process_group_ranks(group)
        _host_groups[key] = dist.new_group(ranks=ranks, backend="gloo")     # (collective: every rank reaches this call)
    return _host_groups[key]


def gather_shards(local, group=None, dst=0):
    """Host arrays of the contiguous per-rank shards (dist.shard_range order) -> the full arrays ON RANK `dst` ONLY (None
    elsewhere), in rank order = input order (patch_VAE.py:454,459 stack in file-path order).  The data path itself has no
    collective; this is the final hand-over of results on the host: rank `dst` allocates the (N, cols) result once and
    every other rank's shard is received straight into its rows (point-to-point over gloo, no pickling, nothing
    replicated -- round 3 all-gathered pickled copies of every shard onto every rank)."""
    if world_size(group) == 1:
        return local
    import numpy as np
    single = not isinstance(local, (tuple, list))
    arrs = [np.ascontiguousarray(a) for a in ((local,) if single else local)]
    hg = host_group(group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    to_global = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    # every rank's row count and the column layout of rank dst's result (a rank with an empty shard may not know it)
    meta = [None] * world
    dist.all_gather_object(meta, [(a.shape, str(a.dtype)) for a in arrs], group=hg)       # a few bytes per rank
    if rank != dst:
        for a in arrs:
            if a.shape[0] > 0:
                dist.send(torch.from_numpy(a), dst=to_global(dst), group=hg)
        return None
    out = []
    for i, a in enumerate(arrs):
        rows = [m[i][0][0] for m in meta]
        tail = next((m[i][0][1:] for m in meta if m[i][0][0] > 0), a.shape[1:])
        full = np.empty((sum(rows),) + tuple(tail), dtype=a.dtype)
        out.append(full)
    # (same order as the senders: array by array inside a rank)
    for i, full in enumerate(out):
        off = 0
        for r in range(world):
            n = meta[r][i][0][0]
            if n > 0:
                if r == dst:
                    full[off:off + n] = arrs[i]
                else:
                    dist.recv(torch.from_numpy(full[off:off + n]), src=to_global(r), group=hg)
            off += n
    return out[0] if single else type(local)(out)


def collective_evidence(flat, group=None):
    """What a reader needs to see that the collective library really carried the job: backend ("nccl" = RCCL on ROCm),
    its version, the world size, and whether every rank's copy of `flat` (the flat parameter buffer after a step) is
    bit-equal to rank 0's -- identical replicas are the invariant of data-parallel training with one all-reduce per step.
    Collective: every rank calls it."""
    w = world_size(group)
    rec = {"backend": dist.get_backend(group) if w > 1 else None, "world": w, "nccl_version": None, "replicas_bit_equal": True}
    if w == 1:
        return rec
    if rec["backend"] == "nccl":
        try:
            rec["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001  (a reporting field)
            rec["nccl_version"] = f"unavailable ({type(e).__name__})"
    ref = flat.detach().clone()
    dist.broadcast(ref, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
    same = torch.tensor([1.0 if torch.equal(ref, flat.detach()) else 0.0], device=flat.device)
    dist.all_reduce(same, op=dist.ReduceOp.MIN, group=group)
    rec["replicas_bit_equal"] = bool(same.item() == 1.0)
    return rec


def max_over_ranks(value, device=None, group=None):
    """Scalar max over ranks (bench timing contract)."""
    if world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
