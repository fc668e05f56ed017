"""Training step of the VQ-VAE path.

Two ways to train, same kernels underneath:

  * drop-in (reference call pattern, run_training.py:404-408):
        _, loss_dict = model(batch, **kwargs); loss_dict['total_loss'].backward(); optimizer.step()
    with any torch optimizer -- goes through torch.autograd (dynamorph_amd.vq_vae).

  * FusedTrainer.step(batch): the MI355X-first path.  All 43 trainable tensors are views of ONE flat
    fp32 buffer, their gradients views of a second one, so a step is
        forward kernels -> backward kernels (write straight into the flat gradient buffer)
        -> ONE RCCL all-reduce of that buffer (data parallel, one process per GPU)
        -> ONE fused Adam launch,
    no autograd bookkeeping, no host synchronisation, and the forward+backward launch sequence is
    captured into a HIP graph and replayed (hipGraph instead of a tracing compiler).

`run_one_batch` / `train` mirror run_training.py:377-417 / :455-551 (same arguments and loop).
"""
import os

import numpy as np
import torch

from . import dist as D
from . import engine as E
from . import ops
from .train_utils import EarlyStopping

LOSS_KEYS = ("recon_loss", "commitment_loss", "total_loss", "perplexity")


class FusedTrainer:
    """Adam(lr, betas=(.9,.999), eps=1e-8) exactly as run_training.py:485 builds it, fused."""

    def __init__(self, model, lr=1e-3, betas=(.9, .999), eps=1e-8, process_group=None, use_graph=True):
        from .vq_vae import VQ_VAE
        if not isinstance(model, VQ_VAE):
            raise TypeError("FusedTrainer is built for VQ_VAE / VQ_VAE_z16; train other modules with a torch optimizer")
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.group = process_group
        self.world = D.world_size(process_group)
        params = [p for p in model.parameters() if p.requires_grad]
        if params[0].device.type != "cuda":
            raise RuntimeError("FusedTrainer: move the model to the GPU first (no CPU fallback)")
        self.fp = D.FlatParams(params)
        self.params, self.flat, self.grad = self.fp.params, self.fp.flat, self.fp.grad
        dev, n = self.flat.device, self.flat.numel()
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.step_dev = torch.zeros(2, device=dev)        # completed-step counter, ping-ponged between the two words
        self._step_slot = 0
        self.w_recon = torch.tensor([float(model.weight_recon)], device=dev)
        self.w_commit = torch.tensor([float(model.weight_commitment)], device=dev)
        self.use_graph = use_graph
        self._graph = None
        self._static_x = None
        self._static_mask = None
        self._static_tm = None
        self._static_out = None
        D.broadcast_(self.flat, list(model.buffers()), group=self.group)    # same replica everywhere

    # ------------------------------------------------------------------------------------------
    def G(self, p):
        return self.fp.gview(p)

    def expose_grads(self):
        """Make p.grad point at the flat gradient views (for inspection / torch tooling)."""
        self.fp.expose_grads()

    def _time_matching(self, sim, tm):
        """(loss, d loss / d sim) of the pairwise term on the (B, B) matrix of mean-squared latent distances:
        vq_vae.py:330-331 (sum of sim * matrix) or, for VQ_VAE_z16, vae.py:327-336 (weights, hinge, mean)."""
        model = self.model
        if not getattr(model, "_z16_loss", False):
            return (sim * tm).sum(), tm
        wts = torch.where(tm == 2, torch.full_like(tm, model.w_a),
                          torch.where(tm == 1, torch.full_like(tm, model.w_t),
                                      torch.where(tm == 0, torch.full_like(tm, model.w_n), tm)))
        val = sim * wts
        hinge = tm == 0
        live = torch.where(hinge, (val + model.margin >= 0).to(sim.dtype), torch.ones_like(sim))
        val = torch.where(hinge, torch.clamp(val + model.margin, min=0), val)
        return val.mean(), wts * live / float(sim.numel())

    def forward_backward(self, x, mask=None, time_matching_mat=None):
        """One forward + backward; returns the device tensor (recon, commitment, total, perplexity[, time matching])."""
        model = self.model
        L = E.Layers(model)
        cc = float(model.commitment_cost)
        z, ecx = E.encoder_forward(L, x)
        zq, idx, vsc = E.vq_forward(L.codebook.weight, z, cc)
        # the tail of the decoder (dec.4, dec.6, loss) runs inside decoder_backward, fused with its own backward
        dec, dcx = E.decoder_forward(L, zq, x, mask, defer_tail=True)
        B, NIN, H, W = x.shape
        g_zq = E.decoder_backward(L, dcx, self.w_recon, None, self.G)
        scalars = ops.loss_finalize(dcx.loss_slabs, B * NIN * H * W, vsc, float(model.weight_recon),
                                    float(model.weight_commitment))
        gcb = self.G(L.codebook.weight)
        # codebook gradient as slabs, added in the encoder's single slab reduction: no atomics, nothing to zero
        dz, cb_slabs = ops.vq_backward_slabs(z, L.codebook.weight.detach(), idx, g_zq, self.w_commit, cc)
        extra = [(cb_slabs, gcb)]
        if time_matching_mat is not None:
            # pairwise term on z_before (vq_vae.py:324-332): HIP kernels for the (B, B) distances and their gradient,
            # the B*B weighting in torch
            zf = z.reshape(B, -1)
            sim = ops.pair_msd(zf)
            tml, g_sim = self._time_matching(sim, time_matching_mat.to(sim.dtype))
            wm = float(model.weight_matching)
            dz = dz + ops.pair_msd_backward(zf, (g_sim * wm).contiguous()).reshape(z.shape)
            scalars = torch.cat([scalars[:2], (scalars[2] + wm * tml).reshape(1), scalars[3:4], tml.reshape(1)])
        # the flat gradient buffer starts at zero and nothing ever writes the BatchNorm-fed conv biases' slots
        E.encoder_backward(L, ecx, dz, self.G, zero_fed_biases=False, pending_extra=extra)
        return scalars

    def _allreduce(self):
        D.allreduce_mean_(self.grad, self.group)     # every loss is a mean over the local batch

    def _adam(self):
        a, b = self._step_slot, 1 - self._step_slot
        ops.adam_counted(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                         self.step_dev[a:a + 1], self.step_dev[b:b + 1])
        self._step_slot = b

    def step(self, x, mask=None, time_matching_mat=None):
        """One optimisation step on a device batch; returns the device tensor of LOSS_KEYS values (+ the time-matching
        loss as a fifth entry when a matrix is given)."""
        if not x.is_cuda:
            raise RuntimeError("FusedTrainer.step: batch must be on the GPU")
        x = x.contiguous()
        if not self.use_graph:
            out = self.forward_backward(x, mask, time_matching_mat)
        else:
            out = self._graph_step(x, mask, time_matching_mat)
        self._allreduce()
        self._adam()
        return out

    def _graph_step(self, x, mask, tm=None):
        key = (tuple(x.shape), None if mask is None else tuple(mask.shape), None if tm is None else tuple(tm.shape))
        if self._graph is None or self._graph_key != key:
            self._static_x = torch.empty_like(x)
            self._static_mask = torch.empty_like(mask) if mask is not None else None
            self._static_tm = torch.empty_like(tm, dtype=torch.float32) if tm is not None else None
            self._static_x.copy_(x)
            if mask is not None:
                self._static_mask.copy_(mask)
            if tm is not None:
                self._static_tm.copy_(tm)
            # warm-up on a side stream (allocator + lazy init), then capture.  The warm-up really executes,
            # so the BatchNorm running statistics it advanced are put back: only replays count as steps.
            bufs = list(self.model.buffers())
            saved = [b.clone() for b in bufs]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.forward_backward(self._static_x, self._static_mask, self._static_tm)
            torch.cuda.current_stream().wait_stream(s)
            for b, sv in zip(bufs, saved):
                b.copy_(sv)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._static_out = self.forward_backward(self._static_x, self._static_mask, self._static_tm)
            self._graph, self._graph_key = g, key
        else:
            if x.data_ptr() != self._static_x.data_ptr():      # a loader may write straight into input_buffer()
                self._static_x.copy_(x)
            if mask is not None and mask.data_ptr() != self._static_mask.data_ptr():
                self._static_mask.copy_(mask)
            if tm is not None:
                self._static_tm.copy_(tm)
        self._graph.replay()
        return self._static_out

    def input_buffer(self):
        """The captured graph's input tensor (None before the first step): fill it in place to skip the copy."""
        return self._static_x


class GraphedTrainer:
    """Any of the modules (VQ_VAE_z32 in particular, which FusedTrainer does not cover): the reference's step --
    model(x) / total_loss.backward() / Adam.step() (run_training.py:404-408, 485) -- recorded once per input shape into a HIP
    graph through autograd and replayed.  Same arithmetic as the eager loop with torch.optim.Adam (capturable form: the
    step count lives on the device); what goes away is the per-launch host work.  Measured on VQ_VAE_z32 the GPU is already
    the limit of the eager loop (no gain at B = 256..2048), so train() uses it only on request (fused="graph"): it is for
    small batches and busy hosts.  Single process (no gradient exchange)."""

    def __init__(self, model, lr=1e-3, betas=(.9, .999), eps=1e-8):
        params = [p for p in model.parameters() if p.requires_grad]
        if not params or params[0].device.type != "cuda":
            raise RuntimeError("GraphedTrainer: move the model to the GPU first (no CPU fallback)")
        self.model = model
        self.opt = torch.optim.Adam(params, lr=lr, betas=betas, eps=eps, capturable=True, foreach=True)
        self._graphs = {}            # input shapes -> (graph, static x, static mask, static matrix, static output)

    def _eager(self, x, mask, tm):
        self.opt.zero_grad(set_to_none=True)      # backward then assigns fresh gradients (from the graph's pool on replay)
        _, ld = self.model(x, time_matching_mat=tm, batch_mask=mask)
        ld["total_loss"].backward()
        self.opt.step()
        vals = [ld[k].detach().reshape(()) for k in LOSS_KEYS]
        if tm is not None:
            vals.append(ld["time_matching_loss"].detach().reshape(()))
        return torch.stack(vals)

    def _capture(self, x, mask, tm):
        sx = x.clone()
        smask = mask.clone() if mask is not None else None
        stm = tm.clone().float() if tm is not None else None
        # warm-up on a side stream (allocator, lazy state of the optimizer); it really executes, so parameters, buffers
        # and the optimizer state it touched are put back afterwards: only replays count as steps
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        saved = [t.detach().clone() for t in tensors]
        had_state = len(self.opt.state) > 0
        opt_saved = [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state[p].items()}
                     for g in self.opt.param_groups for p in g["params"]] if had_state else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._eager(sx, smask, stm)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for t, sv in zip(tensors, saved):
                t.copy_(sv)
            i = 0
            for g in self.opt.param_groups:
                for p in g["params"]:
                    for k, v in self.opt.state[p].items():
                        if torch.is_tensor(v):
                            v.copy_(opt_saved[i][k]) if had_state else v.zero_()
                    i += 1
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._eager(sx, smask, stm)
        return g, sx, smask, stm, out

    def step(self, x, mask=None, time_matching_mat=None):
        """One optimisation step; returns the device tensor of LOSS_KEYS values (+ the time-matching loss when given)."""
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainer.step: batch must be on the GPU")
        tm = time_matching_mat
        key = (tuple(x.shape), None if mask is None else tuple(mask.shape), None if tm is None else tuple(tm.shape))
        if key not in self._graphs:
            self._graphs[key] = self._capture(x.contiguous(), mask, tm)       # (a ragged last batch gets its own graph)
        else:
            _, sx, smask, stm, _ = self._graphs[key]
            sx.copy_(x)
            if mask is not None:
                smask.copy_(mask)
            if tm is not None:
                stm.copy_(tm)
        g, _, _, _, out = self._graphs[key]
        g.replay()
        return out


# ================================================================ reference-style loop mirrors
def run_one_batch(model, batch, train_loss, model_kwargs=None, optimizer=None, transform=None, training=True):
    """run_training.py:377-417.  `optimizer` may be a torch optimizer (autograd path), a FusedTrainer or a GraphedTrainer."""
    model_kwargs = model_kwargs or {}
    if transform is not None:
        flips = torch.from_numpy(np.random.choice([0, 1, 2], size=len(batch))).to(device=batch.device, dtype=torch.int32)
        rots = torch.from_numpy(np.random.choice([0, 1, 2, 3], size=len(batch))).to(device=batch.device, dtype=torch.int32)
        batch = ops.augment(batch.contiguous(), flips, rots)       # one kernel instead of the O(B) python loop
    fused = isinstance(optimizer, (FusedTrainer, GraphedTrainer))
    if fused and training:
        vals = optimizer.step(batch, model_kwargs.get("batch_mask"), model_kwargs.get("time_matching_mat"))
        vals = vals.tolist()                                           # one device sync per step (reference: five)
        loss_dict = dict(zip(LOSS_KEYS, vals))
        loss_dict["time_matching_loss"] = vals[4] if len(vals) > 4 else 0.
    else:
        _, loss_dict = model(batch, **model_kwargs)
        if training:
            loss_dict['total_loss'].backward()
            D.allreduce_grads_(list(model.parameters()))       # data parallel: one flat bucket (no-op in one process)
            optimizer.step()
            model.zero_grad()
    for key, loss in loss_dict.items():
        train_loss.setdefault(key, []).append(float(loss))
    return model, train_loss


def get_relation_tensor(relation_mat, sample_ids, device='cuda:0'):
    """run_training.py:335-355: the (B, B) block of the symmetric sample-relation matrix (scipy sparse or dense) for
    this batch as a float32 tensor -- the `time_matching_mat` argument of VQ_VAE.forward."""
    if relation_mat is None:
        return None
    block = relation_mat[sample_ids, :][:, sample_ids]
    if hasattr(block, "todense"):
        block = block.todense()
    out = torch.from_numpy(np.ascontiguousarray(np.asarray(block), dtype=np.float32))
    return out.to(device) if device else out


def get_mask(mask, sample_ids, device='cuda:0'):
    """run_training.py:358-374: cell masks of this batch.  `mask` is a TensorDataset-like object whose first tensor is
    (N, 2, H, W) in {-1, 1}; the second channel (the large mask) is kept and mapped to {0, 1} -> (B, 1, H, W), the
    `batch_mask` argument of VQ_VAE.forward."""
    if mask is None:
        return None
    m = mask[sample_ids][0][:, 1:2, :, :]
    m = (m + 1.) / 2.
    return m.to(device)


def train(model, dataset, output_dir, relation_mat=None, mask=None, n_epochs=10, lr=0.001, batch_size=16,
          device='cuda:0', shuffle_data=False, transform=None, val_split_ratio=0.15, patience=20,
          get_relation_tensor=None, get_mask=None, writer=None, fused=True):
    """run_training.py:455-551: Adam, contiguous validation block, epoch/batch loops, EarlyStopping
    checkpoint of the state_dict to <output_dir>/model.pt.  `dataset` is a TensorDataset-like object
    indexable with a list of ids (dataset[ids][0] -> host tensor)."""
    assert val_split_ratio is None or 0 < val_split_ratio < 1
    if patience is not None:
        assert val_split_ratio is not None
    from .vq_vae import VQ_VAE
    if get_relation_tensor is None and relation_mat is not None:
        get_relation_tensor = globals()["get_relation_tensor"]
    if get_mask is None and mask is not None:
        get_mask = globals()["get_mask"]
    # the fused path is built for the 16x16-latent architecture; other modules (VQ_VAE_z32) train through autograd
    if D.world_size() > 1 and not isinstance(model, VQ_VAE):
        D.broadcast_(torch.zeros(1, device=device), [p.data for p in model.parameters()] + list(model.buffers()))   # same replica everywhere
    if fused == "graph" and D.world_size() == 1:
        optimizer = GraphedTrainer(model, lr=lr)          # any module: the autograd step as a replayed HIP graph
    elif fused and isinstance(model, VQ_VAE):
        optimizer = FusedTrainer(model, lr=lr)
    else:
        # VQ_VAE_z32 is GPU bound through autograd at every batch size measured (8.2 ms eager vs 8.5 ms replayed at
        # B = 2048), so the plain loop stays the default there
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, betas=(.9, .999))
    model.zero_grad()
    n_samples = len(dataset)
    sample_ids = list(range(n_samples))
    split = int(np.floor(val_split_ratio * n_samples))
    split_start = np.random.randint(0, n_samples - split)
    if shuffle_data:
        np.random.shuffle(sample_ids)
    val_ids = sample_ids[split_start: split_start + split]
    train_ids = sample_ids[:split_start] + sample_ids[split_start + split:]
    n_train, n_val = len(train_ids), len(val_ids)
    n_batches = int(np.ceil(n_train / batch_size))
    n_val_batches = int(np.ceil(n_val / batch_size))
    os.makedirs(output_dir, exist_ok=True)
    model_path = os.path.join(output_dir, 'model.pt')
    early_stopping = EarlyStopping(patience=patience, verbose=True, path=model_path)
    for epoch in range(n_epochs):
        train_loss, val_loss = {}, {}
        print('start epoch %d' % epoch)
        for phase, ids, nb, losses in (("train", train_ids, n_batches, train_loss), ("val", val_ids, n_val_batches, val_loss)):
            for i in range(nb):
                ids_batch = ids[i * batch_size:min((i + 1) * batch_size, len(ids))]
                batch = dataset[ids_batch][0].to(device)
                kw = {'time_matching_mat': get_relation_tensor(relation_mat, ids_batch, device=device) if get_relation_tensor else None,
                      'batch_mask': get_mask(mask, ids_batch, device=device) if get_mask else None}
                model, losses = run_one_batch(model, batch, losses, optimizer=optimizer, model_kwargs=kw,
                                              transform=transform, training=(phase == "train"))
        if shuffle_data:
            np.random.shuffle(train_ids)
        for key, loss in train_loss.items():
            train_loss[key] = sum(loss) / len(loss)
            if writer is not None:
                writer.add_scalar('Loss/' + key, train_loss[key], epoch)
        for key, loss in val_loss.items():
            val_loss[key] = sum(loss) / len(loss)
            if writer is not None:
                writer.add_scalar('Val loss/' + key, val_loss[key], epoch)
        early_stopping(val_loss['total_loss'], model)
        if early_stopping.early_stop:
            print("Early stopping")
            break
        print('epoch %d' % epoch)
        print('train: ', ''.join(['{}:{:0.4f}  '.format(key, loss) for key, loss in train_loss.items()]))
        print('validation: ', ''.join(['{}:{:0.4f}  '.format(key, loss) for key, loss in val_loss.items()]))
    return model
