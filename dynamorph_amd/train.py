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
# DM_VQ_JOIN=0 in the environment: the last residual join as its own launch (dm_apply) in front of the quantiser (A/B runs)
JOIN_IN_VQ = os.environ.get("DM_VQ_JOIN", "1") != "0"


class _Marks:
    """Four (stream event, host clock) pairs around the parts of one FusedTrainer.step (measurement only)."""

    def __init__(self):
        self.ev, self.t = [], []

    def __call__(self):
        import time
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append(e)
        self.t.append(time.perf_counter())


class FusedTrainer:
    """Adam(lr, betas=(.9,.999), eps=1e-8) exactly as run_training.py:485 builds it, fused."""

    def __init__(self, model, lr=1e-3, betas=(.9, .999), eps=1e-8, process_group=None, use_graph=True):
        from .vq_vae import VQ_VAE, VQ_VAE_z32
        if not isinstance(model, (VQ_VAE, VQ_VAE_z32)):
            raise TypeError("FusedTrainer is built for VQ_VAE / VQ_VAE_z16 / VQ_VAE_z32; train other modules with a torch optimizer")
        # (a VQ_VAE_z32 with extra_loss, vae.py:463-469: the caller's torch code runs on z_after between the forward and the
        # backward half of the step, _step_with_extra_losses; it needs the labels: step(..., labels=...))
        self._extra = getattr(model, "extra_loss", None) is not None
        if self._extra and not hasattr(model, "alpha"):
            raise AttributeError("FusedTrainer: extra_loss needs model.alpha (vae.py:467; pass alpha= to VQ_VAE_z32)")
        self.model = model
        self._z32 = isinstance(model, VQ_VAE_z32)
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
        # (VQ_VAE_z32 has no loss weights besides weight_matching: total = recon + commitment + matching, vae.py:456-470)
        self.w_recon = torch.tensor([float(getattr(model, "weight_recon", 1.0))], device=dev)
        self.w_commit = torch.tensor([float(getattr(model, "weight_commitment", 1.0))], device=dev)
        self.use_graph = use_graph
        self._graphs = {}            # input shapes -> {x, mask, tm: static inputs; train / eval: (graph, static output)}
        self._static_x = None        # input tensor of the graph replayed last
        D.broadcast_(self.flat, list(model.buffers()), group=self.group)    # same replica everywhere

    # ------------------------------------------------------------------------------------------
    def G(self, p):
        return self.fp.gview(p)

    def expose_grads(self):
        """Make p.grad point at the flat gradient views (for inspection / torch tooling)."""
        self.fp.expose_grads()

    def _time_matching(self, sim, tm, z16_form=None):
        """(loss, d loss / d sim) of the pairwise term on the (B, B) matrix of mean-squared latent distances:
        vq_vae.py:330-331 (sum of sim * matrix) or, for VQ_VAE_z16 / VQ_VAE_z32 (z16_form), vae.py:327-336 (weights,
        hinge, mean).  Only for latent lengths the MFMA kernels do not tile (n % 32 != 0)."""
        model = self.model
        if z16_form is None:
            z16_form = getattr(model, "_z16_loss", False)
        if not z16_form:
            return (sim * tm).sum(), tm
        wts = torch.where(tm == 2, torch.full_like(tm, model.w_a),
                          torch.where(tm == 1, torch.full_like(tm, model.w_t),
                                      torch.where(tm == 0, torch.full_like(tm, model.w_n), tm)))
        val = sim * wts
        hinge = tm == 0
        live = torch.where(hinge, (val + model.margin >= 0).to(sim.dtype), torch.ones_like(sim))
        val = torch.where(hinge, torch.clamp(val + model.margin, min=0), val)
        return val.mean(), wts * live / float(sim.numel())

    def _z32_forward_part(self, x, mask, tm):
        """VQ_VAE_z32 (vae.py:430-470), forward half of the step: two-conv stem + residual stack | VectorQuantizer | residual
        stack + BatchNorm tail, the weighted-hinge time-matching term on z_after.  Same kernels as the autograd path
        (dynamorph_amd.vq_vae), called in order: no autograd bookkeeping -- and a launch sequence a HIP graph can replay.
        Returns the state the backward half reads (st.scalars: recon, commitment, total, perplexity[, time matching];
        st.zq: z_after, what a caller-supplied extra loss acts on, vae.py:463-469)."""
        import types
        m = self.model
        enc, dec = m.enc, m.dec
        st = types.SimpleNamespace(er=enc[5]._handles(), dr=dec[0]._handles(), cc=float(m.commitment_cost))
        B, NIN, H, W = x.shape
        h, st.scx = E.z32_stem_forward(enc[0], enc[1], enc[3], enc[4], x)
        st.z, st.esaved = E.residual_forward(st.er, h)
        st.zq, st.idx, vqs = E.vq_forward(m.vq.w.weight, st.z, st.cc, defer_scalars=True)
        r, st.dsaved = E.residual_forward(st.dr, st.zq)
        _, st.tcx = E.z32_tail_forward(dec[1], dec[2], dec[4], r, x, mask, m.channel_var)
        # the pairwise term acts on z_after (vae.py:441-455); its gradient reaches z through the straight-through value
        st.tm_S = st.tm_fallback = st.zf = None
        st.wm = float(m.weight_matching)
        fin = (vqs.slabs, vqs.ws, vqs.K, vqs.D, vqs.positions, vqs.cc, st.tcx.loss_slabs, B * NIN * H * W, 1.0, 1.0)
        if tm is not None:
            st.zf = st.zq.reshape(B, -1)
            tmf = tm.to(torch.float32).contiguous()
            if ops.time_matching_supported(st.zf.shape[0], st.zf.shape[1]):
                tm_slabs, st.tm_S = ops.time_matching_forward(st.zf, tmf, *_tm_args(m, True), want_slabs=True)
                st.scalars = ops.vq_loss_finalize_tm(*fin, tm_slabs, st.wm)        # the five scalars in one launch
            else:
                tml, g_sim = self._time_matching(ops.pair_msd(st.zf), tmf, True)      # (VQ_VAE_z32: always the weighted-hinge form)
                st.tm_fallback = ops.pair_msd_backward(st.zf, (g_sim * st.wm).contiguous()).reshape(st.zq.shape)
                st.scalars = _with_matching(ops.vq_loss_finalize(*fin), tml, st.wm)
        else:
            st.scalars = ops.vq_loss_finalize(*fin)
        return st

    def _z32_backward_part(self, st, g_extra=None):
        """Backward half: decoder tail, decoder residual stack, the time-matching and (g_extra: d(sum of alpha * extra
        losses) / d z_after, from the caller's torch code) extra-loss gradients joining at z_after, the quantiser's
        straight-through backward, encoder -- ONE slab reduction for every weight / bias / codebook gradient."""
        m = self.model
        enc, dec = m.enc, m.dec
        pending = []
        g_r = E.z32_tail_backward(dec[1], dec[2], dec[4], st.tcx, self.w_recon, None, self.G, pending=pending,
                                  zero_fed_biases=False)
        g_zq, _ = E.residual_backward(st.dr, st.dsaved, g_r, self.G, None, pending=pending, zero_fed_biases=False)
        if st.tm_S is not None:
            g_zq = ops.time_matching_backward(st.zf, st.tm_S, None, st.wm, add=g_zq).reshape(st.zq.shape)     # (summed in the kernel's store)
        elif st.tm_fallback is not None:
            g_zq = g_zq + st.tm_fallback
        if g_extra is not None:
            g_zq = g_zq + g_extra
        gcb = self.G(m.vq.w.weight)
        dz, cb_slabs = ops.vq_backward_slabs(st.z, m.vq.w.weight.detach(), st.idx, g_zq, self.w_commit, st.cc)
        pending.append((cb_slabs, gcb))
        g_h, stats = E.residual_backward(st.er, st.esaved, dz, self.G, st.scx.a2, pending=pending, zero_fed_biases=False)
        E.z32_stem_backward(enc[0], enc[1], enc[3], enc[4], st.scx, g_h, self.G, stats=stats, pending=pending,
                            zero_fed_biases=False)
        ops.reduce_slabs_multi(pending)                     # every weight / bias / codebook gradient of the step
        return st.scalars

    def _forward_backward_z32(self, x, mask, tm):
        return self._z32_backward_part(self._z32_forward_part(x, mask, tm))

    def _extra_losses(self, zq, labels):
        """The caller's extra losses (vae.py:463-469) on z_after, as torch code between the two halves of the step:
        returns (sum of alpha * loss as a device scalar, its gradient w.r.t. z_after, {name: loss})."""
        m = self.model
        leaf = zq.detach().requires_grad_(True)
        flat = leaf.reshape((leaf.shape[0], -1))
        total, named = None, {}
        with torch.enable_grad():
            for name, fn in m.extra_loss.items():
                loss, _frac_pos = fn(labels, flat)
                named[name] = loss.detach()
                total = loss * m.alpha if total is None else total + loss * m.alpha
            total.backward()
        g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        return total.detach(), g.contiguous(), named

    def _step_with_extra_losses(self, x, mask, tm, labels):
        """One forward + backward of a VQ_VAE_z32 with extra_loss: the two halves as captured HIP graphs (one pair per input
        shape), the caller's torch code on z_after in between.  Returns the step's scalars with total_loss including the
        extra terms; self.last_extra_losses = {name: device scalar}."""
        if not self.use_graph:
            st = self._z32_forward_part(x, mask, tm)
            total, g, named = self._extra_losses(st.zq, labels)
            scal = self._z32_backward_part(st, g)
        else:
            key = ("extra", tuple(x.shape), None if mask is None else tuple(mask.shape), None if tm is None else tuple(tm.shape))
            ent = self._graphs.get(key)
            if ent is None:
                sx = x.clone()
                smask = mask.clone() if mask is not None else None
                stm = tm.clone().float() if tm is not None else None
                bufs = list(self.model.buffers())
                saved = [b.clone() for b in bufs]
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                              # warm-up (allocator, lazy init): really executes
                    st = self._z32_forward_part(sx, smask, stm)
                    self._z32_backward_part(st, torch.zeros_like(st.zq))
                torch.cuda.current_stream().wait_stream(side)
                for b, sv in zip(bufs, saved):
                    b.copy_(sv)                                            # ... so the running statistics are put back
                if self.world > 1:
                    torch.cuda.synchronize(self.flat.device)
                gF = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gF):
                    st = self._z32_forward_part(sx, smask, stm)
                gx = torch.zeros_like(st.zq)
                gB = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gB, pool=gF.pool()):
                    scal = self._z32_backward_part(st, gx)
                ent = self._graphs[key] = (gF, gB, st, gx, scal, sx, smask, stm)
            gF, gB, st, gx, scal, sx, smask, stm = ent
            if x.data_ptr() != sx.data_ptr():
                sx.copy_(x)
            if mask is not None:
                smask.copy_(mask)
            if tm is not None:
                stm.copy_(tm)
            gF.replay()
            total, g, named = self._extra_losses(st.zq, labels)
            gx.copy_(g)
            gB.replay()
        self.last_extra_losses = named
        out = scal.clone()
        out[2] += total                                                    # total_loss += alpha * extra (vae.py:467)
        return out

    def forward_backward(self, x, mask=None, time_matching_mat=None):
        """One forward + backward; returns the device tensor (recon, commitment, total, perplexity[, time matching])."""
        if self._z32:
            return self._forward_backward_z32(x, mask, time_matching_mat)
        model = self.model
        L = E.Layers(model)
        cc = float(model.commitment_cost)
        z, ecx = E.encoder_forward(L, x, defer_last_join=L.codebook.weight.shape[0] if JOIN_IN_VQ else 0)
        if z is None:       # the last residual join runs in the quantiser's load path (dm_vq_forward_join), which writes z
            z, zq, idx, vqs = E.vq_forward_joined(L.codebook.weight, ecx.pending_join, cc)
        else:
            zq, idx, vqs = E.vq_forward(L.codebook.weight, z, cc, defer_scalars=True)
        # the tail of the decoder (dec.4, dec.6, loss) runs inside decoder_backward, fused with its own backward
        dec, dcx = E.decoder_forward(L, zq, x, mask, defer_tail=True)
        B, NIN, H, W = x.shape
        dec_pending = []                 # the decoder's slabs ride in the encoder's single reduction at the end of the pass
        g_zq = E.decoder_backward(L, dcx, self.w_recon, None, self.G, pending=dec_pending)
        # (recon, commitment, total, perplexity[, time matching]): the VectorQuantizer's scalars, the reconstruction loss and
        # the pairwise term on z_before (vq_vae.py:324-332) in one launch
        fin = (vqs.slabs, vqs.ws, vqs.K, vqs.D, vqs.positions, vqs.cc, dcx.loss_slabs, B * NIN * H * W,
               float(model.weight_recon), float(model.weight_commitment))
        tm_S = tm_fallback = None
        if time_matching_mat is not None:
            zf = z.reshape(B, -1)
            wm = float(model.weight_matching)
            tm = time_matching_mat.to(torch.float32).contiguous()
            if ops.time_matching_supported(zf.shape[0], zf.shape[1]):
                tm_slabs, tm_S = ops.time_matching_forward(zf, tm, *_tm_args(model), want_slabs=True)
                scalars = ops.vq_loss_finalize_tm(*fin, tm_slabs, wm)
            else:       # latent lengths the MFMA kernels do not tile: distances from dm_pair_msd, the B x B weighting in torch
                tml, g_sim = self._time_matching(ops.pair_msd(zf), tm)
                tm_fallback = ops.pair_msd_backward(zf, (g_sim * wm).contiguous()).reshape(z.shape)
                scalars = _with_matching(ops.vq_loss_finalize(*fin), tml, wm)
        else:
            scalars = ops.vq_loss_finalize(*fin)
        gcb = self.G(L.codebook.weight)
        # codebook gradient as slabs, added in the encoder's single slab reduction: nothing to zero, no global atomics
        # (K <= 64: an ordered one-hot product on the matrix cores, bit-reproducible; larger K: LDS adds in arrival order)
        dz, cb_slabs = ops.vq_backward_slabs(z, L.codebook.weight.detach(), idx, g_zq, self.w_commit, cc)
        extra = [(cb_slabs, gcb)] + dec_pending
        if tm_S is not None:
            dz = ops.time_matching_backward(zf, tm_S, None, wm, add=dz).reshape(z.shape)      # (summed in the kernel's store)
        elif tm_fallback is not None:
            dz = dz + tm_fallback
        # the flat gradient buffer starts at zero and nothing ever writes the BatchNorm-fed conv biases' slots
        E.encoder_backward(L, ecx, dz, self.G, zero_fed_biases=False, pending_extra=extra)
        return scalars

    def _allreduce(self, weight=1.0):
        """SUM over ranks of the flat gradient bucket: ONE collective and nothing behind it -- the "x 1 / world" of the mean
        is applied by the optimizer's load (_adam, dm_adam_counted_scaled), so after the exchange the bucket holds the sum.
        Every loss is a mean over the LOCAL batch, so with equal shards the mean of the per-rank gradients is the gradient
        of the global-batch mean loss; a rank whose shard of a ragged batch is smaller passes weight = n_local * world /
        n_global (0 for an empty shard)."""
        if self.world == 1:
            return
        if weight != 1.0:
            self.grad.mul_(float(weight))
        torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM, group=self.group)

    def _adam(self):
        a, b = self._step_slot, 1 - self._step_slot
        ops.adam_counted(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                         self.step_dev[a:a + 1], self.step_dev[b:b + 1], grad_scale=1.0 / self.world)
        self._step_slot = b

    def step_without_data(self):
        """This rank's shard of a ragged global batch is empty: it contributes a zero gradient to the exchange and takes
        the same Adam step as the others."""
        with torch.cuda.device(self.flat.device):
            self.grad.zero_()
            self._allreduce()
            self._adam()

    def step(self, x, mask=None, time_matching_mat=None, grad_weight=1.0, timers=None, labels=None):
        """One optimisation step on a device batch; returns the device tensor of LOSS_KEYS values (+ the time-matching
        loss as a fifth entry when a matrix is given).  grad_weight: see _allreduce.
        timers: a list -> this step appends (events, host_times): four events on the launch stream and four
        time.perf_counter() readings around its three parts (forward+backward | gradient exchange | Adam); read them with
        FusedTrainer.timer_summary after a synchronize.  Measurement only: the default path records nothing."""
        if not x.is_cuda:
            raise RuntimeError("FusedTrainer.step: batch must be on the GPU")
        if x.device != self.flat.device:
            raise RuntimeError(f"FusedTrainer.step: batch on {x.device}, model on {self.flat.device}")
        x = x.contiguous()
        with torch.cuda.device(self.flat.device):       # graph capture / replay and the streams are the model's device's
            mark = _Marks() if timers is not None else None
            if mark: mark()
            if self._extra:
                out = self._step_with_extra_losses(x, mask, time_matching_mat, labels)
            elif not self.use_graph:
                out = self.forward_backward(x, mask, time_matching_mat)
            else:
                out = self._graph_step(x, mask, time_matching_mat)
            if mark: mark()
            self._allreduce(grad_weight)
            if mark: mark()
            self._adam()
            if mark:
                mark()
                timers.append(mark)
        return out

    @staticmethod
    def timer_summary(timers):
        """Averages (microseconds) over the steps recorded with step(..., timers=list): device time between the stream
        events and host time between the enqueue points, for forward+backward (graph replay), the gradient exchange and
        the fused Adam launch."""
        torch.cuda.synchronize()
        n = max(len(timers), 1)
        out = {}
        for i, part in enumerate(("fwd_bwd", "allreduce", "adam")):
            out[part + "_us"] = round(sum(m.ev[i].elapsed_time(m.ev[i + 1]) for m in timers) * 1e3 / n, 2)
            out[part + "_host_us"] = round(sum(m.t[i + 1] - m.t[i] for m in timers) * 1e6 / n, 2)
        out["steps"] = len(timers)
        return out

    def forward_only(self, x, mask=None, time_matching_mat=None):
        """The validation pass (run_training.py:522-531: forward with the module left in train mode, so BatchNorm uses
        batch statistics and advances its running statistics; no backward, no step).  Same kernels as forward_backward;
        returns the same device tensor (recon, commitment, total, perplexity[, time matching])."""
        m, tm = self.model, time_matching_mat
        B, NIN, H, W = x.shape
        cc = float(m.commitment_cost)
        if self._z32:
            enc, dec = m.enc, m.dec
            h, _ = E.z32_stem_forward(enc[0], enc[1], enc[3], enc[4], x)
            z, _ = E.residual_forward(enc[5]._handles(), h)
            zq, _, vqs = E.vq_forward(m.vq.w.weight, z, cc, defer_scalars=True)
            r, _ = E.residual_forward(dec[0]._handles(), zq)
            _, tcx = E.z32_tail_forward(dec[1], dec[2], dec[4], r, x, mask, m.channel_var)
            slabs, wr, wc, lat = tcx.loss_slabs, 1.0, 1.0, zq
        else:
            L = E.Layers(m)
            z, ecx = E.encoder_forward(L, x, defer_last_join=L.codebook.weight.shape[0] if JOIN_IN_VQ else 0)
            if z is None:
                z, zq, _, vqs = E.vq_forward_joined(L.codebook.weight, ecx.pending_join, cc)
            else:
                zq, _, vqs = E.vq_forward(L.codebook.weight, z, cc, defer_scalars=True)
            _, dcx = E.decoder_forward(L, zq, x, mask)
            slabs, wr, wc, lat = dcx.loss_slabs, float(m.weight_recon), float(m.weight_commitment), z
        fin = (vqs.slabs, vqs.ws, vqs.K, vqs.D, vqs.positions, vqs.cc, slabs, B * NIN * H * W, wr, wc)
        if tm is None:
            return ops.vq_loss_finalize(*fin)
        zf = lat.reshape(B, -1)
        tmf = tm.to(torch.float32).contiguous()
        if ops.time_matching_supported(zf.shape[0], zf.shape[1]):
            tm_slabs, _ = ops.time_matching_forward(zf, tmf, *_tm_args(m, self._z32), want_slabs=True)
            return ops.vq_loss_finalize_tm(*fin, tm_slabs, float(m.weight_matching))
        tml, _ = self._time_matching(ops.pair_msd(zf), tmf, self._z32 or getattr(m, "_z16_loss", False))
        return _with_matching(ops.vq_loss_finalize(*fin), tml, float(m.weight_matching))

    def evaluate(self, x, mask=None, time_matching_mat=None):
        """forward_only through a captured HIP graph (one per input shape); returns the device tensor of loss values."""
        if not x.is_cuda or x.device != self.flat.device:
            raise RuntimeError(f"FusedTrainer.evaluate: batch on {x.device}, model on {self.flat.device}")
        with torch.cuda.device(self.flat.device):
            if not self.use_graph:
                return self.forward_only(x.contiguous(), mask, time_matching_mat)
            return self._graph_step(x.contiguous(), mask, time_matching_mat, kind="eval")

    def prepare(self, x, mask=None, time_matching_mat=None):
        """Capture the HIP graph for this input shape WITHOUT taking a step (the capture's internal warm-up run has its
        BatchNorm side effects put back, parameters and Adam state are untouched): a caller that times steps can keep the
        one-off capture out of its timed region.  Returns the graph's input buffer (fill it in place to skip the copy)."""
        if not self.use_graph:
            return None
        with torch.cuda.device(self.flat.device):
            self._graph_step(x.contiguous(), mask, time_matching_mat, replay=False)
        return self._static_x

    def static_inputs(self, x_shape, mask_shape=None, tm_shape=None):
        """(x, mask, matrix) buffers the graphs of this input shape read: a loader that fills them in place and passes
        them to step() / evaluate() skips every copy (dynamorph_amd.train._Feed writes its gathered batches here).
        Allocated on first use, shared by the training and the validation graph of the shape; nothing is captured here."""
        key = (tuple(x_shape), None if mask_shape is None else tuple(mask_shape), None if tm_shape is None else tuple(tm_shape))
        ent = self._graphs.get(key)
        if ent is None:
            dev = self.flat.device
            ent = self._graphs[key] = {
                "x": torch.zeros(key[0], device=dev), "mask": torch.zeros(key[1], device=dev) if key[1] else None,
                "tm": torch.zeros(key[2], device=dev) if key[2] else None, "train": None, "eval": None}
        return ent["x"], ent["mask"], ent["tm"]

    def _graph_step(self, x, mask, tm=None, replay=True, kind="train"):
        """Graphs are cached per input shape (a ragged last batch gets its own, captured once, not once per epoch)."""
        sx, smask, stm = self.static_inputs(x.shape, None if mask is None else mask.shape, None if tm is None else tm.shape)
        if x.data_ptr() != sx.data_ptr():           # a loader may write straight into the static buffers
            sx.copy_(x)
        if mask is not None and mask.data_ptr() != smask.data_ptr():
            smask.copy_(mask)
        if tm is not None and tm.data_ptr() != stm.data_ptr():
            stm.copy_(tm)
        ent = self._graphs[(tuple(x.shape), None if mask is None else tuple(mask.shape), None if tm is None else tuple(tm.shape))]
        if ent[kind] is None:
            fn = self.forward_backward if kind == "train" else self.forward_only
            # warm-up on a side stream (allocator + lazy init), then capture.  The warm-up really executes,
            # so the BatchNorm running statistics it advanced are put back: only replays count as steps.
            bufs = list(self.model.buffers())
            saved = [b.clone() for b in bufs]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn(sx, smask, stm)
            torch.cuda.current_stream().wait_stream(s)
            for b, sv in zip(bufs, saved):
                b.copy_(sv)
            if self.world > 1:
                # no collective of this process may be in flight while the stream is capturing (the broadcast of the
                # constructor, the previous step's all-reduce: their completion is polled by the backend's watchdog thread)
                torch.cuda.synchronize(self.flat.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sout = fn(sx, smask, stm)
            ent[kind] = (g, sout)
        self._static_x = sx
        if replay:
            ent[kind][0].replay()
        return ent[kind][1]

    def input_buffer(self):
        """Input tensor of the graph replayed last (None before the first step): fill it in place to skip the copy."""
        return self._static_x


def _tm_args(model, z32=False):
    """(mode, w_a, w_t, w_n, margin) of dm_time_matching_forward for this model family (vq_vae.py:330-331 / vae.py:327-336)."""
    z16 = z32 or getattr(model, "_z16_loss", False)
    return (1 if z16 else 0, float(getattr(model, "w_a", 0.0)), float(getattr(model, "w_t", 0.0)),
            float(getattr(model, "w_n", 0.0)), float(getattr(model, "margin", 0.0)))


def _with_matching(scalars, tml, wm):
    """(recon, commitment, total, perplexity) + the pairwise term -> the five values of a step with a relation matrix."""
    tml = tml.reshape(1)
    return torch.cat([scalars[:2], scalars[2:3] + wm * tml, scalars[3:4], tml])


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
        with torch.cuda.device(x.device):
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
def _augment(batch):
    """run_training.py:396-403: a random flip (none / up-down / left-right) and a random multiple of 90 degrees per
    sample, drawn from numpy's global generator in the reference's order (flip, rotation, flip, rotation, ...: a caller's
    np.random.seed reproduces the reference's augmentation, ops.augment_codes) -- applied by ONE kernel instead of the
    O(B) loop."""
    flips, rots = ops.augment_codes(len(batch))
    flips = torch.from_numpy(flips).to(batch.device)
    rots = torch.from_numpy(rots).to(batch.device)
    return ops.augment(batch.contiguous(), flips, rots)


def run_one_batch(model, batch, train_loss, model_kwargs=None, optimizer=None, transform=None, training=True,
                  grad_weight=1.0):
    """run_training.py:377-417.  `optimizer` may be a torch optimizer (autograd path), a FusedTrainer or a GraphedTrainer.
    grad_weight scales this rank's gradient before the data-parallel mean (FusedTrainer._allreduce); 1 in one process."""
    model_kwargs = model_kwargs or {}
    if transform is not None:
        batch = _augment(batch)
    if isinstance(optimizer, (FusedTrainer, GraphedTrainer)) and training:
        # (train_with_loader hands the labels of the extra losses through model_kwargs, run_training.py:596-599)
        kw = {"grad_weight": grad_weight, "labels": model_kwargs.get("labels")} if isinstance(optimizer, FusedTrainer) else {}
        vals = optimizer.step(batch, model_kwargs.get("batch_mask"), model_kwargs.get("time_matching_mat"), **kw)
        vals = vals.tolist()                                           # one device sync per step (reference: five)
        loss_dict = dict(zip(LOSS_KEYS, vals))
        loss_dict["time_matching_loss"] = vals[4] if len(vals) > 4 else 0.
        loss_dict = _in_model_order(model, loss_dict)
        if getattr(optimizer, "_extra", False):                        # vae.py:469: one entry per extra loss, after total_loss
            loss_dict.update({k: float(v) for k, v in optimizer.last_extra_losses.items()})
    else:
        _, loss_dict = model(batch, **model_kwargs)
        if training:
            loss_dict['total_loss'].backward()
            params = [p for p in model.parameters() if p.requires_grad]
            if grad_weight != 1.0:
                for p in params:
                    if p.grad is not None:
                        p.grad.mul_(float(grad_weight))
            D.allreduce_grads_(params)                          # data parallel: one flat bucket (no-op in one process)
            optimizer.step()
            model.zero_grad()
    for key, loss in loss_dict.items():
        train_loss.setdefault(key, []).append(float(loss))
    return model, train_loss


def _step_without_data(model, optimizer):
    """A rank whose shard of a ragged global batch is empty still joins the gradient exchange (with zeros) and the step."""
    if isinstance(optimizer, FusedTrainer):
        optimizer.step_without_data()
        return
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        p.grad = torch.zeros_like(p)
    D.allreduce_grads_(params)
    optimizer.step()
    model.zero_grad()


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


class _EpochLosses:
    """Epoch value of every entry of the loss dict.

    One process: the reference's aggregation, run_training.py:538-543 -- sum(per-batch values) / number of batches.
    Data parallel (world > 1): every rank sees only its shard of a batch, so the per-key sums of (local loss x local
    samples) and the sample count are exchanged ONCE per phase and every rank holds the same sample-weighted means (and
    takes the same early-stopping decision).  With a ragged last batch that differs slightly from the mean of batch
    values, and for the pairwise time-matching term (a sum over the LOCAL pairs, vq_vae.py:331) it is a different
    quantity altogether: documented deviation, there is no multi-device behaviour in the reference to match.
    The exchanged vector has the same length on every rank: the key list is rank 0's (rank 0 holds the first shard of
    every batch, dist.shard_range, so it has data whenever any rank has); a rank that had no data in the whole phase
    contributes zeros."""

    def __init__(self, device, world=1):
        self.values, self.sums, self.count, self.device, self.world = {}, {}, 0.0, device, world

    def add(self, batch_losses, n):
        for key, values in batch_losses.items():
            v = float(values[-1])
            self.values.setdefault(key, []).append(v)
            self.sums[key] = self.sums.get(key, 0.0) + v * n
        self.count += n

    def means(self):
        if self.world == 1:
            return {k: sum(v) / len(v) for k, v in self.values.items()}
        keys = D.broadcast_object(list(self.sums))
        tot = D.allreduce_sum_host([self.sums.get(k, 0.0) for k in keys] + [self.count], device=self.device)
        return {k: v / max(tot[-1], 1.0) for k, v in zip(keys, tot[:-1])}


def _make_optimizer(model, lr, fused):
    from .vq_vae import VQ_VAE
    from .vq_vae import VQ_VAE_z32
    # (a model with caller-supplied extra losses, vae.py:463-469, runs arbitrary torch code per step: the autograd path)
    fusable = isinstance(model, (VQ_VAE, VQ_VAE_z32)) and getattr(model, "extra_loss", None) is None
    if D.world_size() > 1 and not (fused and fused != "graph" and fusable):
        # FusedTrainer broadcasts its flat buffer itself; any other module: same replica everywhere before the first step
        for t in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t.data, src=0)
    if fused == "graph" and D.world_size() == 1:
        return GraphedTrainer(model, lr=lr)               # any module: the autograd step as a replayed HIP graph
    if fused and fusable:
        return FusedTrainer(model, lr=lr)
    return torch.optim.Adam(model.parameters(), lr=lr, betas=(.9, .999))


class _LossLog:
    """Per-batch loss values of one phase, kept on the device: a step's values are copied into a row (stream-ordered,
    the captured step overwrites its output tensor on the next replay) and the whole phase is read back ONCE, instead of
    the reference's float(loss) per key and step (run_training.py:409-414), which stalls the host on every step."""

    def __init__(self, device, n_batches, width=8, model=None):
        self.buf = torch.zeros((max(n_batches, 1), width), device=device)
        self.meta = []                      # (keys, number of values, samples) per logged batch
        self.model = model

    def add(self, keys, vals, n):
        k = vals.numel()
        self.buf[len(self.meta), :k].copy_(vals.detach().reshape(-1), non_blocking=True)
        self.meta.append((keys, k, n))

    def rows(self):
        """[(dict key -> float, samples)] in batch order; the one device synchronisation of the phase."""
        host = self.buf[:max(len(self.meta), 1)].tolist()
        out = []
        for (keys, k, n), row in zip(self.meta, host):
            d = dict(zip(keys, row[:k]))
            d.setdefault("time_matching_loss", 0.)
            out.append((_in_model_order(self.model, d) if self.model is not None else d, n))
        return out


_FUSED_KEYS = LOSS_KEYS + ("time_matching_loss",)


def loss_key_order(model):
    """Key order of the loss dict the model's forward returns -- what run_one_batch's `for key, loss in
    train_loss_dict.items()` (run_training.py:409) and with it the order of the epoch's writer.add_scalar rows follow:
    vq_vae.py:333-338 (VQ_VAE: ..., total_loss, perplexity) and vae.py:337-342, 456-470 (VQ_VAE_z16 / VQ_VAE_z32: ...,
    perplexity, total_loss).  The fused steps return a flat vector of values; their rows are put back in this order."""
    z16 = getattr(model, "_z16_loss", False) or type(model).__name__ == "VQ_VAE_z32"
    tail = ("perplexity", "total_loss") if z16 else ("total_loss", "perplexity")
    return ("recon_loss", "commitment_loss", "time_matching_loss") + tail


def _in_model_order(model, d):
    order = loss_key_order(model)
    out = {k: d[k] for k in order if k in d}
    out.update((k, v) for k, v in d.items() if k not in out)        # (extra_loss entries keep their place at the end)
    return out


def _device_step(model, optimizer, x, kw, training, grad_weight):
    """One batch that is already on the device -> (loss keys, device tensor of their values); nothing here waits for the
    GPU.  Same arithmetic as run_one_batch for every kind of optimizer."""
    mask, tm = kw.get("batch_mask"), kw.get("time_matching_mat")
    if isinstance(optimizer, FusedTrainer):
        vals = (optimizer.step(x, mask, tm, grad_weight=grad_weight) if training else optimizer.evaluate(x, mask, tm))
        return _FUSED_KEYS[:vals.numel()], vals
    if isinstance(optimizer, GraphedTrainer) and training:
        vals = optimizer.step(x, mask, tm)
        return _FUSED_KEYS[:vals.numel()], vals
    with torch.enable_grad() if training else torch.no_grad():
        _, loss_dict = model(x, **kw)
        if training:
            loss_dict['total_loss'].backward()
            params = [p for p in model.parameters() if p.requires_grad]
            if grad_weight != 1.0:
                for p in params:
                    if p.grad is not None:
                        p.grad.mul_(float(grad_weight))
            D.allreduce_grads_(params)
            optimizer.step()
            model.zero_grad()
    keys = tuple(k for k, v in loss_dict.items() if torch.is_tensor(v))      # (a float entry is the constant 0. of :382)
    return keys, torch.stack([loss_dict[k].detach().reshape(()).float() for k in keys])


def train(model, dataset, output_dir, relation_mat=None, mask=None, n_epochs=10, lr=0.001, batch_size=16,
          device='cuda:0', shuffle_data=False, transform=None, val_split_ratio=0.15, patience=20,
          get_relation_tensor=None, get_mask=None, writer=None, fused=True, feed="auto", stats=None, probe=None):
    """The training loop of run_training.py:455-551 -- Adam, a contiguous validation block at a random start, epoch and
    batch loops, TensorBoard-style scalars, EarlyStopping checkpoint of the state_dict to <output_dir>/model.pt -- made
    data parallel (one process per GPU, torch.distributed initialised by the launcher):

      * one process: split, shuffles, augmentation codes and epoch losses exactly as the reference draws / aggregates them
        (numpy's global generator in the reference's order, mean of the per-batch values);
      * world > 1: the validation split and every shuffle come from ONE seed drawn on rank 0, so all ranks walk the same
        batches;
      * each global batch of `batch_size` samples is cut into contiguous per-rank shards (dist.shard_range); a rank
        weights its gradient by n_local * world / n_global before the single all-reduce, so the averaged gradient is the
        global-batch mean loss's (BatchNorm statistics and the pairwise time-matching term stay rank-local: standard
        data-parallel semantics, the reference has no multi-device behaviour to match);
      * epoch losses are exchanged once per epoch, the early-stopping decision is therefore the same everywhere;
      * rank 0 alone writes model.pt (atomically), the others wait at a barrier.

    feed (dynamorph_amd.feed): where a batch comes from.  "auto": the dataset, its masks and the CSR relation matrix are
    uploaded once and stay in HBM when they fit ("resident": a batch is one gather + augment launch into the captured
    step's input buffer, losses are read back once per phase, the host never waits inside an epoch), else "stream"
    (pinned staging, copy stream, two device slots: PCIe-bound); "sync": the reference's loop as it is (host gather,
    synchronous copy, float() per step) -- taken automatically on the CPU, for dataset objects that only support
    dataset[ids], and when the caller brings its own get_relation_tensor / get_mask.  All feeds produce the same batches.
    stats: a dict that receives {"feed", "phase_seconds": {phase: [per epoch]} (device time between the phase's first and
    last launch; host wall time on the CPU), "phase_samples": {phase: n}, "epoch_seconds": [wall clock per epoch],
    "step_losses": {phase: [per epoch: [loss dict of every batch, in order]]}}.
    probe: callable(phase, epoch, ids, x, kwargs) called with every batch right before its step -- the sample ids of this
    rank's shard, the (augmented) batch and the model kwargs exactly as the step is about to read them (the tests hold
    them against what the reference's loop hands its model, tests/golden/g11_train_loop.npz); it must copy what it keeps.

    `dataset` is a TensorDataset-like object indexable with a list of ids (dataset[ids][0] -> host tensor)."""
    assert val_split_ratio is None or 0 < val_split_ratio < 1
    if patience is not None:
        assert val_split_ratio is not None
    custom_hooks = get_relation_tensor is not None or get_mask is not None
    if get_relation_tensor is None and relation_mat is not None:
        get_relation_tensor = globals()["get_relation_tensor"]
    if get_mask is None and mask is not None:
        get_mask = globals()["get_mask"]
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    rank, world = D.get_rank(), D.world_size()
    optimizer = _make_optimizer(model, lr, fused)
    model.zero_grad()

    if feed not in ("auto", "resident", "stream", "sync"):
        raise ValueError(f"train: unknown feed {feed!r}")
    feeder = None
    if feed != "sync":
        from . import feed as F
        src = F.dataset_tensor(dataset)
        usable = (dev.type == "cuda" and not custom_hooks and src is not None and src.dim() == 4 and src.shape[2] == src.shape[3]
                  and (mask is None or F.dataset_tensor(mask) is not None))
        if usable:
            feeder = F.Feed(dataset, dev, mode=feed, mask=mask, relation_mat=relation_mat, batch_size=batch_size,
                            trainer=optimizer if isinstance(optimizer, FusedTrainer) else None)
        elif feed != "auto":
            raise ValueError(f"train: feed={feed!r} needs a CUDA device, a tensor-backed dataset and the default "
                             "get_relation_tensor / get_mask")
    if stats is not None:
        stats.update(feed=feeder.mode if feeder else "sync", phase_seconds={"train": [], "val": []}, phase_samples={},
                     step_losses={"train": [], "val": []})

    # the reference's loop takes dataset[ids][0]; a bare tensor / ndarray (what upload_zscored returns) is indexed directly
    bare = None
    if torch.is_tensor(dataset) or isinstance(dataset, np.ndarray):
        bare = torch.as_tensor(dataset)
    n_samples = len(dataset)
    # (an int array instead of the reference's list: the same draws shuffle it into the same order -- numpy's shuffle is
    # the same Fisher-Yates walk for both -- and slicing a phase into batches costs nothing)
    sample_ids = np.arange(n_samples, dtype=np.int64)
    split = int(np.floor(val_split_ratio * n_samples))
    if world == 1:
        # the reference's draws from numpy's global generator, in the reference's order (run_training.py:490-493, 536):
        # a caller's np.random.seed reproduces the reference's split and shuffles
        order = np.random
    else:
        seed = D.broadcast_object(int(np.random.randint(0, 2 ** 31 - 1)))
        order = np.random.RandomState(seed)                 # split and shuffles: the same stream on every rank
        np.random.seed((seed + 7919 * rank) % (2 ** 32))    # augmentation draws: a stream of its own per rank
    split_start = int(order.randint(0, n_samples - split))
    if shuffle_data:
        order.shuffle(sample_ids)
    phases = {"train": np.concatenate([sample_ids[:split_start], sample_ids[split_start + split:]]),
              "val": sample_ids[split_start: split_start + split].copy()}

    os.makedirs(output_dir, exist_ok=True)
    early_stopping = EarlyStopping(patience=patience, verbose=(rank == 0), path=os.path.join(output_dir, 'model.pt'))
    early_stopping.writes = rank == 0
    say = print if rank == 0 else (lambda *a, **k: None)
    import time
    timed = stats is not None and dev.type == "cuda"
    for epoch in range(n_epochs):
        say('start epoch %d' % epoch)
        t_epoch = time.perf_counter()
        epoch_means, logs, marks = {}, {}, {}
        for phase, ids in phases.items():
            t_phase = time.perf_counter()
            if timed:
                marks[phase] = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
                marks[phase][0].record()
            training = phase == "train"
            losses = _EpochLosses(dev if dev.type == "cuda" else None, world)
            # this rank's shard of every global batch and the weight of its gradient in the data-parallel mean
            plan = []
            for start in range(0, len(ids), batch_size):
                ids_batch = ids[start:start + batch_size]
                lo, hi = D.shard_range(len(ids_batch), rank, world)
                plan.append((ids_batch[lo:hi], D.shard_weight(len(ids_batch), rank, world)))
            if feeder is not None:
                log = _LossLog(dev, len(plan), model=model)
                batches = feeder.phase([p[0] for p in plan], transform, fused=isinstance(optimizer, FusedTrainer))
                for ids_local, weight in plan:
                    if not len(ids_local):
                        if training:
                            _step_without_data(model, optimizer)
                        continue
                    n, x, kw = next(batches)
                    if probe is not None:
                        probe(phase, epoch, ids_local, x, kw)
                    keys, vals = _device_step(model, optimizer, x, kw, training, weight)
                    log.add(keys, vals, n)
                for _ in batches:                                   # (runs the generator to its end)
                    pass
                logs[phase] = (log, losses)                         # read back after BOTH phases are enqueued
            else:
                per_step = []
                for ids_local, weight in plan:
                    if not len(ids_local):
                        if training:
                            _step_without_data(model, optimizer)
                        continue
                    ids_local = ids_local.tolist()                  # (the reference indexes with lists)
                    if bare is not None:
                        batch = bare[ids_local].to(dev)             # a bare tensor / ndarray: dataset[ids][0] would be ONE sample
                    else:
                        batch = dataset[ids_local][0].to(dev)
                    kw = {'time_matching_mat': get_relation_tensor(relation_mat, ids_local, device=dev) if get_relation_tensor else None,
                          'batch_mask': get_mask(mask, ids_local, device=dev) if get_mask else None}
                    last = {}
                    if transform is not None:
                        batch = _augment(batch)                     # (run_one_batch's first statement, run_training.py:396)
                    if probe is not None:
                        probe(phase, epoch, ids_local, batch, kw)
                    run_one_batch(model, batch, last, optimizer=optimizer, model_kwargs=kw, transform=None,
                                  training=training, grad_weight=weight)
                    losses.add(last, len(ids_local))
                    if stats is not None:
                        per_step.append({k: v[-1] for k, v in last.items()})
                epoch_means[phase] = losses.means()
                if stats is not None:
                    stats["step_losses"][phase].append(per_step)
            if stats is not None:
                if timed:
                    marks[phase][1].record()
                else:
                    stats["phase_seconds"][phase].append(time.perf_counter() - t_phase)
                stats["phase_samples"][phase] = sum(len(p[0]) for p in plan)
        # the epoch's one device synchronisation: the loss rows of both phases (the validation pass was enqueued behind the
        # training steps without waiting for them)
        for phase, (log, losses) in logs.items():
            rows = log.rows()
            for row, n in rows:
                losses.add({k: [v] for k, v in row.items()}, n)
            epoch_means[phase] = losses.means()
            if stats is not None:
                stats["step_losses"][phase].append([row for row, _ in rows])
        if timed:
            torch.cuda.synchronize(dev)
            for phase, (e0, e1) in marks.items():                   # device time between the phase's first and last launch
                stats["phase_seconds"][phase].append(e0.elapsed_time(e1) * 1e-3)
        if shuffle_data:
            order.shuffle(phases["train"])
        if writer is not None and rank == 0:
            for phase, prefix in (("train", 'Loss/'), ("val", 'Val loss/')):
                for key, value in epoch_means[phase].items():
                    writer.add_scalar(prefix + key, value, epoch)
        early_stopping(epoch_means["val"]['total_loss'], model)
        D.barrier()                                          # model.pt is complete before any rank moves on
        if stats is not None:
            stats.setdefault("epoch_seconds", []).append(time.perf_counter() - t_epoch)      # wall clock, checkpoint included
        if early_stopping.early_stop:
            say("Early stopping")
            break
        say('epoch %d' % epoch)
        for phase, label in (("train", 'train: '), ("val", 'validation: ')):
            say(label, ''.join(['{}:{:0.4f}  '.format(key, loss) for key, loss in epoch_means[phase].items()]))
    if feeder is not None:
        feeder.close()
    return model
