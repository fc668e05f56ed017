"""Batch feeding for train(): where the samples of a batch come from and how they reach the step's input buffer.

The reference's loop (run_training.py:509-520) gathers every batch on the host (`dataset[ids][0]`, a fancy-index copy of
B x 131 KB), copies it over PCIe synchronously (`.to(device)`), slices the relation matrix with scipy and the masks with
another fancy index, then augments sample by sample in Python.  At B = 2048 that is 268 MB per step against a 2.2 ms GPU
step.  Three feeds, same batches bit for bit:

  resident   the MI355X-first one: the whole dataset (fp32), the mask planes and the CSR relation matrix are uploaded
             ONCE into HBM (288 GB: 2 M patches of 2 x 128 x 128 fit); a batch is then one gather + augment launch
             (dm_gather_augment) that writes straight into the input buffer the captured step replays on, one row
             gather for the masks (dm_gather_rows) and one CSR block launch (dm_csr_block).  The ids of a whole phase
             and its flip / rotation codes are uploaded once per phase: nothing crosses PCIe per step and the host
             never waits for the device inside an epoch.
  stream     the dataset does not fit: a helper thread gathers batch i+1 into pinned staging while batch i trains, a copy
             stream moves it into one of two device slots, and the same gather + augment kernel (identity ids) writes it
             into the step's input buffer.  Bounded by the host gather and PCIe (268 MB per 2048 patches), not the GPU.
  sync       the reference's loop as it is (host gather, pageable copy, float() per step) for dataset objects that only
             support `dataset[ids]`, and as the A/B partner of the tests.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops


def dataset_tensor(dataset):
    """The (N, C, H, W) host tensor behind a TensorDataset-like object (`dataset[ids][0]` == tensor[ids]), or None when
    the object only supports indexing."""
    if torch.is_tensor(dataset):
        return dataset
    ts = getattr(dataset, "tensors", None)
    if ts is not None and len(ts) >= 1 and torch.is_tensor(ts[0]):
        return ts[0]
    if isinstance(dataset, np.ndarray):
        return torch.from_numpy(dataset)
    return None


def upload_zscored(raw, channel_mean, channel_std, device, chunk_bytes=1 << 29):
    """What run_training.py:880 makes of a pickled dataset -- zscore(np.squeeze(dataset), channel_mean, channel_std)
    .astype(np.float32) -- as a DEVICE tensor, without the host passes: the raw float64 / float32 patches go up in chunks
    and are z-scored there (dm_zscore_channels, bit-equal to the numpy expression).  `raw`: (N, C, H, W) or
    (N, C, 1, H, W) numpy array or host tensor.  The result can be handed to train() as the dataset: the resident feed uses
    a device tensor where it lies."""
    t = torch.as_tensor(raw)
    if t.dim() == 5 and t.shape[2] == 1:
        t = t[:, :, 0]
    if t.dim() != 4 or t.dtype not in (torch.float64, torch.float32):
        raise ValueError("upload_zscored: an (N, C, [1,] H, W) float64 or float32 array")
    dev = torch.device(device)
    out = torch.empty(tuple(t.shape), dtype=torch.float32, device=dev)
    rows = max(1, chunk_bytes // max(1, t[0].numel() * t.element_size()))
    with torch.cuda.device(dev):
        for lo in range(0, t.shape[0], rows):
            part = t[lo:lo + rows].contiguous().to(dev)
            ops.zscore_channels(part, channel_mean, channel_std, out=out[lo:lo + rows])
    return out


def mask_plane(mask_tensor):
    """run_training.py:371-372 for every sample at once: the second mask channel (the large mask), {-1, 1} -> {0, 1}."""
    m = mask_tensor[:, 1:2, :, :]
    return ((m + 1.) / 2.).to(torch.float32).contiguous()


def _csr_arrays(relation_mat):
    """(indptr int64, indices int32, data float32, n) of the (n, n) relation matrix, duplicates summed (what
    `.todense()` of run_training.py:350 does) -- from a scipy sparse matrix or a dense array."""
    import scipy.sparse as sp
    m = relation_mat if sp.issparse(relation_mat) else sp.csr_matrix(np.asarray(relation_mat))
    m = m.tocsr().copy()
    m.sum_duplicates()
    if m.shape[0] != m.shape[1]:
        raise ValueError("relation matrix must be square")
    return (torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int32)),
            torch.from_numpy(np.asarray(m.data).astype(np.float32)), m.shape[0])


def resident_budget(device):
    """Bytes of HBM the resident feed may take: DM_RESIDENT_BYTES if set, else DM_RESIDENT_FRACTION (default 0.6) of what
    is free now (the step's activations, ~1.6 MB per patch of the batch, have to fit beside it)."""
    if "DM_RESIDENT_BYTES" in os.environ:
        return int(os.environ["DM_RESIDENT_BYTES"])
    free, _ = torch.cuda.mem_get_info(device)
    return int(free * float(os.environ.get("DM_RESIDENT_FRACTION", "0.6")))


class Feed:
    """One per train() call.  `phase(batches, transform)` yields, for every non-empty list of sample ids in `batches`,
    (n, x, kwargs): the batch as a device tensor and the model kwargs (`time_matching_mat`, `batch_mask`) -- written
    into `trainer.static_inputs(...)` when a FusedTrainer is given, into buffers of the feed otherwise."""

    def __init__(self, dataset, device, mode="auto", mask=None, relation_mat=None, batch_size=None, trainer=None):
        self.dev = torch.device(device)
        if self.dev.type == "cuda" and self.dev.index is None:      # 'cuda' names the current device: compare like with like
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.trainer = trainer
        src = dataset_tensor(dataset)
        if src is None or src.dim() != 4:
            raise ValueError("Feed: the dataset must expose an (N, C, H, W) tensor (TensorDataset / tensor / ndarray); "
                             "use feed='sync' for objects that only support dataset[ids]")
        if src.shape[2] != src.shape[3]:
            raise ValueError("Feed: square patches only")
        self.src = src
        self.N, self.C, self.H = src.shape[0], src.shape[1], src.shape[2]
        self.bs = int(batch_size or 1)
        mten = dataset_tensor(mask) if mask is not None else None
        if mask is not None and (mten is None or mten.shape[0] != self.N or mten.shape[1] < 2):
            raise ValueError("Feed: masks must expose an (N, >= 2, H, W) tensor")
        self.has_mask = mten is not None
        data_bytes = self.N * self.C * self.H * self.H * 4
        mask_bytes = self.N * mten.shape[2] * mten.shape[3] * 4 if self.has_mask else 0
        if mode not in ("auto", "resident", "stream"):
            raise ValueError(f"Feed: unknown mode {mode!r}")
        if src.is_cuda:
            # a dataset that already lives in HBM (upload_zscored) IS resident: it is used where it lies, only the mask
            # planes still have to fit, and the streaming path (host gather, pinned staging) has nothing to read from
            if src.device != self.dev or src.dtype != torch.float32 or not src.is_contiguous():
                raise ValueError(f"Feed: a device dataset must be a contiguous float32 tensor on {self.dev} "
                                 f"(got {src.dtype} on {src.device}, contiguous={src.is_contiguous()})")
            if mode == "stream":
                raise ValueError("Feed: feed='stream' reads the dataset from host memory; this one is on the device")
            if mode == "auto" and mask_bytes > resident_budget(self.dev):
                raise ValueError(f"Feed: the dataset is on the device but its {mask_bytes} bytes of mask planes do not fit "
                                 "beside it (DM_RESIDENT_BYTES / DM_RESIDENT_FRACTION)")
            mode, data_bytes = "resident", 0
        need = data_bytes + mask_bytes
        if mode == "auto":
            mode = "resident" if need <= resident_budget(self.dev) else "stream"
        self.mode = mode
        self.bytes_resident = 0
        with torch.cuda.device(self.dev):
            if mode == "resident":
                self.data = self._upload(src)
                self.mplane = self._upload(mask_plane(mten)) if self.has_mask else None
                self.bytes_resident = need              # what this feed added to HBM (a device dataset was there before)
            else:
                self.mplane_host = mask_plane(mten) if self.has_mask else None
                self._init_stream()
            self.csr = None
            if relation_mat is not None:
                indptr, indices, data, n = _csr_arrays(relation_mat)
                if n != self.N:
                    raise ValueError("relation matrix and dataset disagree on the number of samples")
                self.csr = (indptr.to(self.dev), indices.to(self.dev), data.to(self.dev), n)
                self.pos = torch.zeros(n, dtype=torch.int64, device=self.dev)
                self.stamp = 0
        self._own = {}               # buffers of the feed (no FusedTrainer, or shapes it does not own)

    # ------------------------------------------------------------------------------------------ resident
    def _upload(self, t, chunk_bytes=1 << 29):
        """Host tensor -> fp32 device tensor in chunks (a pinned source goes at the link rate; a pageable one through the
        runtime's staging; another dtype is converted on the way).  A contiguous fp32 tensor already on this device (e.g.
        from upload_zscored) is used where it lies."""
        if t.is_cuda and t.device == self.dev and t.dtype == torch.float32 and t.is_contiguous():
            return t
        out = torch.empty(tuple(t.shape), dtype=torch.float32, device=self.dev)
        rows = max(1, chunk_bytes // max(1, t[0].numel() * t.element_size()))
        nb = t.is_pinned() and t.dtype == torch.float32           # (anything else is staged by the runtime: blocking)
        for lo in range(0, t.shape[0], rows):
            out[lo:lo + rows].copy_(t[lo:lo + rows], non_blocking=nb)
        return out

    # ------------------------------------------------------------------------------------------ streaming
    def _init_stream(self):
        shape = (self.bs, self.C, self.H, self.H)
        self.pin = [torch.empty(shape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.slot = [torch.empty(shape, dtype=torch.float32, device=self.dev) for _ in range(2)]
        if self.has_mask:
            mshape = (self.bs,) + tuple(self.mplane_host.shape[1:])
            self.mpin = [torch.empty(mshape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self.mslot = [torch.empty(mshape, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.ev_ready = [torch.cuda.Event() for _ in range(2)]      # the batch has reached slot k
        self.ev_used = [torch.cuda.Event() for _ in range(2)]       # the kernels reading slot k have finished
        self.helper = ThreadPoolExecutor(1)
        # the host gather of a batch (B x 131 KB rows picked from the dataset) is cut into chunks over a few threads: one
        # thread copies ~19 GB/s on the MI355X host, PCIe takes 55 GB/s
        self.gather_threads = max(1, min(int(os.environ.get("DM_FEED_THREADS", "6")), (os.cpu_count() or 2) // 2))
        self.gatherers = ThreadPoolExecutor(self.gather_threads)

    def _stage(self, k, ids):
        """Helper thread: host gather of one batch into pinned staging k, then its copy into device slot k."""
        n = len(ids)
        ids = np.asarray(ids, dtype=np.int64)
        run = n > 0 and int(ids[-1] - ids[0]) == n - 1 and bool((np.diff(ids) == 1).all())
        self.ev_ready[k].synchronize()                               # staging k's previous copy has left the host
        if run and self.src.dtype == torch.float32 and self.src.is_pinned():
            src_x = self.src[int(ids[0]):int(ids[0]) + n]            # consecutive samples of a pinned dataset: DMA from where they lie
        else:
            src_x = self.pin[k][:n]
            self._host_gather(self.src, ids, run, src_x)
        if self.has_mask:
            self._host_gather(self.mplane_host, ids, run, self.mpin[k][:n])
        with torch.cuda.device(self.dev), torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.ev_used[k])             # slot k is no longer being read
            self.slot[k][:n].copy_(src_x, non_blocking=True)
            if self.has_mask:
                self.mslot[k][:n].copy_(self.mpin[k][:n], non_blocking=True)
            self.ev_ready[k].record(self.copy_stream)

    def _host_gather(self, src, ids, run, out):
        """out[j] = float32(src[ids[j]]) on the host, in chunks over the gather threads (torch releases the GIL inside the
        copies); a run of consecutive ids is a plain slice copy."""
        n = len(ids)
        step = max(1, -(-n // self.gather_threads))

        def part(lo):
            hi = min(n, lo + step)
            first = int(ids[0])
            if run and src.dtype == torch.float32:
                np.copyto(out[lo:hi].numpy(), src[first + lo:first + hi].numpy())     # (a plain memcpy per thread)
            elif run:
                out[lo:hi].copy_(src[first + lo:first + hi])
            elif src.dtype == torch.float32:
                torch.index_select(src, 0, torch.from_numpy(ids[lo:hi]), out=out[lo:hi])
            else:
                out[lo:hi].copy_(src.index_select(0, torch.from_numpy(ids[lo:hi])))   # the reference's cast to fp32
        list(self.gatherers.map(part, range(0, n, step)))

    # ------------------------------------------------------------------------------------------ targets
    def _targets(self, n, want_tm, fused):
        xs = (n, self.C, self.H, self.H)
        ms = ((n,) + tuple((self.mplane if self.mode == "resident" else self.mplane_host).shape[1:])) if self.has_mask else None
        ts = (n, n) if want_tm else None
        if fused and self.trainer is not None:
            return self.trainer.static_inputs(xs, ms, ts)
        key = (n, want_tm)
        if key not in self._own:
            self._own[key] = (torch.empty(xs, device=self.dev), torch.empty(ms, device=self.dev) if ms else None,
                              torch.empty(ts, device=self.dev) if ts else None)
        return self._own[key]

    def phase(self, batches, transform=None, fused=True):
        """Generator over the non-empty batches of one phase (lists of sample ids): (n, x, kwargs)."""
        batches = [np.asarray(b) for b in batches if len(b)]
        if not batches:
            return
        total = sum(len(b) for b in batches)
        with torch.cuda.device(self.dev):
            # the ids and the augmentation codes of the WHOLE phase in ONE small asynchronous upload from pinned staging.
            # The codes are drawn batch by batch in the order the reference draws them (nothing else draws from numpy's
            # generator inside a phase).
            parts = [np.concatenate(batches).astype(np.int32, copy=False)]
            if transform is not None:
                # (one draw for the phase: per sample and in batch order, exactly the interleaved stream of the batch loop)
                parts += list(ops.augment_codes(total))
            # two pinned staging blocks, used in turn (the copy out of one may still be queued when the next phase fills
            # the other); a third phase waits for the first block's copy
            need = len(parts) * total
            k = self._meta_turn = 1 - getattr(self, "_meta_turn", 1)
            if not hasattr(self, "_meta_pin"):
                self._meta_pin, self._meta_ev = [None, None], [torch.cuda.Event(), torch.cuda.Event()]
            if self._meta_pin[k] is None or self._meta_pin[k].numel() < need:
                self._meta_pin[k] = torch.empty(max(need, 1 << 16), dtype=torch.int32, pin_memory=True)
            else:
                self._meta_ev[k].synchronize()
            stage = self._meta_pin[k][:need]
            np.concatenate(parts, out=stage.numpy())
            meta_dev = stage.to(self.dev, non_blocking=True)
            self._meta_ev[k].record(torch.cuda.current_stream(self.dev))
            ids_dev = meta_dev[:total]
            flips, rots = (meta_dev[total:2 * total], meta_dev[2 * total:]) if transform is not None else (None, None)
            compute = torch.cuda.current_stream(self.dev)
            pending = None
            if self.mode == "stream":
                pending = self.helper.submit(self._stage, 0, batches[0])
            off = 0
            for i, ids in enumerate(batches):
                n = len(ids)
                x, m, tm = self._targets(n, self.csr is not None, fused)
                sl = slice(off, off + n)
                fl, ro = (flips[sl], rots[sl]) if flips is not None else (None, None)
                if self.mode == "resident":
                    ops.gather_augment(self.data, ids_dev[sl], fl, ro, x, n)
                    if m is not None:
                        ops.gather_rows(self.mplane, ids_dev[sl], m, n)
                else:
                    k = i & 1
                    pending.result()
                    if i + 1 < len(batches):
                        pending = self.helper.submit(self._stage, 1 - k, batches[i + 1])
                    compute.wait_event(self.ev_ready[k])
                    ops.gather_augment(self.slot[k], None, fl, ro, x, n)
                    if m is not None:
                        ops.gather_rows(self.mslot[k], None, m, n)
                    self.ev_used[k].record(compute)
                if tm is not None:
                    self.stamp = self.stamp % ((1 << 31) - 2) + 1
                    ops.csr_block(*self.csr, ids_dev[sl], self.pos, self.stamp, tm)
                off += n
                yield n, x, {"time_matching_mat": tm, "batch_mask": m}
        assert off == total

    def close(self):
        if self.mode == "stream":
            self.helper.shutdown(wait=True)
            self.gatherers.shutdown(wait=True)
