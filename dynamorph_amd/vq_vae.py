"""Drop-in nn.Module surface of the reference's VQ-VAE, computed by the HIP kernels.

Mirrors (constructor signatures, attribute / submodule names, state-dict keys, return values):
  VectorQuantizer  HiddenStateExtractor/vq_vae.py:25-116  (== vae.py:12-103)
  ResidualBlock    HiddenStateExtractor/vq_vae.py:180-225
  VQ_VAE           HiddenStateExtractor/vq_vae.py:228-342
  VQ_VAE_z16       HiddenStateExtractor/vae.py:216-346
so `getattr(vae, network)(**kw).to(device)`, `load_state_dict(torch.load(model.pt))`,
`model.enc(x)`, `model.vq(z)`, `model(batch, time_matching_mat=..., batch_mask=...)`,
`total_loss.backward()` and `optimizer.step()` (run_training.py:885-910, 404-408;
pipeline/patch_VAE.py:425-452) work unchanged.

The nn.Conv2d / BatchNorm2d / ConvTranspose2d / Embedding children are parameter containers only
(that is what keeps the 68 state-dict keys); their ATen forward is never called.  All compute goes
through dynamorph_amd.engine -> libdynamorph_hip.so.  Inputs must live on the GPU: there is no CPU
fallback (a CPU tensor raises).
"""
import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from . import ops

CHANNEL_VAR = np.array([1., 1.])


def _require_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: input is on {t.device}; the dynamorph_amd HIP path only runs on the GPU "
                           "(no CPU fallback). Move the model and the batch to cuda first.")


def _prep(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _GradBag:
    """G(param) -> the tensor a backward kernel writes that parameter's gradient into."""

    def __init__(self):
        self.store = {}

    def __call__(self, p):
        g = self.store.get(id(p))
        if g is None:
            g = torch.empty_like(p)
            self.store[id(p)] = g
        return g

    def grads_for(self, params, needs):
        return tuple(self.store.get(id(p)) if need else None for p, need in zip(params, needs))


# =============================================================================== autograd
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layers, per_sample, *params):
        z, cx = E.encoder_forward(layers, x, per_sample=per_sample)
        ctx.layers, ctx.cx, ctx.params = layers, cx, params
        return z

    @staticmethod
    def backward(ctx, g_z):
        G = _GradBag()
        dx = E.encoder_backward(ctx.layers, ctx.cx, g_z, G, want_dx=ctx.needs_input_grad[0])
        ctx.cx = None
        return (dx, None, None) + G.grads_for(ctx.params, ctx.needs_input_grad[3:])


class _ResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, res_layers, *params):
        out, saved = E.residual_forward(res_layers, h)
        ctx.res_layers, ctx.saved, ctx.params = res_layers, saved, params
        return out

    @staticmethod
    def backward(ctx, g):
        G = _GradBag()
        pending = []
        g_in, _ = E.residual_backward(ctx.res_layers, ctx.saved, g.contiguous(), G, None, pending=pending)
        ops.reduce_slabs_multi(pending)
        ctx.saved = None
        return (g_in, None) + G.grads_for(ctx.params, ctx.needs_input_grad[2:])


class _VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, codebook, commitment_cost):
        out, idx, scalars = E.vq_forward(codebook, z, commitment_cost)
        ctx.save_for_backward(z, codebook, idx)
        ctx.cc = commitment_cost
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return out, scalars[0], scalars[1], idx

    @staticmethod
    def backward(ctx, g_out, g_loss, g_perp, _g_idx):
        z, codebook, idx = ctx.saved_tensors
        if g_out is not None:
            g_out = g_out.contiguous()
        if g_loss is None:
            g_loss = torch.zeros(1, device=z.device)
        dz, dw = ops.vq_backward(z, codebook.detach(), idx, g_out, g_loss.reshape(1).contiguous(), ctx.cc,
                                 want_dz=ctx.needs_input_grad[0])
        return dz, (dw if ctx.needs_input_grad[1] else None), None


def _recon_input_grad(cx, channel_var, gscale):
    """d(recon_loss) / d(inputs) of vq_vae.py:320-322 -- mean(((decoded - inputs) * mask)^2 / channel_var) -- for callers that
    differentiate w.r.t. the patches (saliency maps; the reference's training never does): 2 (inputs - decoded) mask^2 /
    (N channel_var), times the upstream gradient.  Plain elementwise torch on the device tensors the forward kept."""
    if gscale is None or cx.x is None or getattr(cx, "dec", None) is None:
        return None
    x, dec = cx.x, cx.dec
    g = (x - dec) * (2.0 / x.numel())
    if cx.mask is not None:
        g = g * (cx.mask * cx.mask)
    if channel_var is not None:
        g = g / channel_var.detach().reshape(1, -1, 1, 1)
    return g * gscale.reshape(())


class _DecoderFn(torch.autograd.Function):
    """decoded (and, when x is given, the masked reconstruction loss of vq_vae.py:320-322)."""

    @staticmethod
    def forward(ctx, zq, x, mask, layers, *params):
        dec, cx = E.decoder_forward(layers, zq, x, mask)
        ctx.layers, ctx.cx, ctx.params = layers, cx, params
        ctx.set_materialize_grads(False)
        if x is None:
            return dec, None
        B, NIN, H, W = x.shape
        zeros = torch.zeros(2, device=x.device)
        recon = ops.loss_finalize(cx.loss_slabs, B * NIN * H * W, zeros, 1.0, 0.0)[0]
        return dec, recon

    @staticmethod
    def backward(ctx, g_dec, g_recon):
        G = _GradBag()
        gscale = g_recon.reshape(1).contiguous() if g_recon is not None else None
        if g_dec is not None:
            g_dec = g_dec.contiguous()
        if gscale is None and g_dec is None:
            return (None,) * (4 + len(ctx.params))
        g_x = _recon_input_grad(ctx.cx, ctx.layers.channel_var, gscale) if ctx.needs_input_grad[1] else None
        g_zq = E.decoder_backward(ctx.layers, ctx.cx, gscale, g_dec, G, want_gz=ctx.needs_input_grad[0])
        ctx.cx = None
        return (g_zq, g_x, None, None) + G.grads_for(ctx.params, ctx.needs_input_grad[4:])


# ================================================================================ modules
class _PairMSDFn(torch.autograd.Function):
    """sim[i][j] = mean((z[i] - z[j])**2)  (vq_vae.py:327-329) on the HIP kernels dm_pair_msd(_backward)."""

    @staticmethod
    def forward(ctx, zf):
        _require_gpu(zf, "time-matching loss")
        z = zf.detach().contiguous().float()
        ctx.save_for_backward(z)
        return ops.pair_msd(z)

    @staticmethod
    def backward(ctx, g_sim):
        (z,) = ctx.saved_tensors
        return ops.pair_msd_backward(z, g_sim.contiguous().float())


class _TimeMatchingFn(torch.autograd.Function):
    """The time-matching loss as ONE op: Gram matrix on the MFMA, the loss form of the model family (mode 0:
    vq_vae.py:330-331; mode 1: vae.py:327-336) in the kernel's epilogue, the backward a second MFMA product
    (dm_time_matching_forward / _backward).  Returns the scalar loss."""

    @staticmethod
    def forward(ctx, zf, tm, mode, w_a, w_t, w_n, margin):
        _require_gpu(zf, "time-matching loss")
        z = zf.detach().contiguous().float()
        loss, S = ops.time_matching_forward(z, tm.detach().contiguous().float(), mode, w_a, w_t, w_n, margin)
        ctx.save_for_backward(z, S)
        ctx.tm_state = getattr(S, "_dm_tm_state", None)     # (saved tensors come back as new objects: attributes do not)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        z, S = ctx.saved_tensors
        if ctx.tm_state is not None:
            S._dm_tm_state = ctx.tm_state
        return (ops.time_matching_backward(z, S, g.reshape(1).contiguous().float()),) + (None,) * 6


def time_matching_loss(latents, time_matching_mat, z16_form, w_a=0.0, w_t=0.0, w_n=0.0, margin=0.0):
    """latents (B, ...) -> scalar loss of the pairwise term; the reference's expressions when the MFMA kernels do not tile
    the latent length (n % 32 != 0): distances from dm_pair_msd, weights / hinge / reduction in torch."""
    zf = latents.reshape((latents.shape[0], -1))
    assert (zf.shape[0], zf.shape[0]) == tuple(time_matching_mat.shape)
    if ops.time_matching_supported(zf.shape[0], zf.shape[1]):
        return _TimeMatchingFn.apply(zf, time_matching_mat, 1 if z16_form else 0, float(w_a), float(w_t), float(w_n), float(margin))
    sim_mat = _PairMSDFn.apply(zf)
    if not z16_form:
        return (sim_mat * time_matching_mat).sum()                          # vq_vae.py:331
    wts = time_matching_mat.clone()                                          # vae.py:327-335
    wts[time_matching_mat == 2] = w_a
    wts[time_matching_mat == 1] = w_t
    wts[time_matching_mat == 0] = w_n
    val = sim_mat * wts
    val = torch.where(time_matching_mat == 0, torch.clamp(val + margin, min=0), val)
    return val.mean()


class VectorQuantizer(nn.Module):
    """Vector quantizer of "Neural Discrete Representation Learning" (reference vq_vae.py:25-116)."""

    def __init__(self, embedding_dim=128, num_embeddings=128, commitment_cost=0.25, device='cuda:0'):
        super(VectorQuantizer, self).__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.device = device
        self.w = nn.Embedding(num_embeddings, embedding_dim)

    def forward(self, inputs):
        """inputs (B, D, H, W) -> (output_quantized, loss, perplexity)   [vq_vae.py:52-84]"""
        _require_gpu(inputs, "VectorQuantizer.forward")
        out, loss, perplexity, _ = _VQFn.apply(_prep(inputs), self.w.weight, float(self.commitment_cost))
        assert out.shape == inputs.shape
        return out, loss, perplexity

    @property
    def embeddings(self):
        return self.w.weight

    def encode_inputs(self, inputs):
        """Index tensor (B, H, W) int64 of the nearest embedding vectors   [vq_vae.py:90-103]"""
        _require_gpu(inputs, "VectorQuantizer.encode_inputs")
        idx, _, _, _ = ops.vq_forward(_prep(inputs.detach()), self.w.weight.detach(), want_out=False)
        return idx

    def decode_inputs(self, encoding_indices):
        """Quantized encodings (B, D, H, W) assembled from indices   [vq_vae.py:105-116]"""
        _require_gpu(encoding_indices, "VectorQuantizer.decode_inputs")
        return ops.vq_decode(encoding_indices.contiguous(), self.w.weight.detach())


def _held(base):
    """A layer of the reference's nn.Sequential kept as a PARAMETER CONTAINER: same class hierarchy, constructor, parameter
    and buffer names (the state-dict contract), but calling it directly raises -- the arithmetic of the path runs in the HIP
    pipeline of the parent (`model.enc`, `model.dec`, `ResidualBlock`), and a child called on its own would silently be
    ATen / MIOpen, i.e. a fallback."""
    class Held(base):
        def forward(self, *args, **kwargs):
            raise RuntimeError(f"{base.__name__} inside dynamorph_amd is a parameter container of the HIP pipeline: call the "
                               "enclosing module (model.enc / model.dec / ResidualBlock); there is no ATen fallback")
    Held.__name__ = base.__name__              # what a reader of repr(model) / type(layer).__name__ sees: the reference's
    Held.__qualname__ = "_" + base.__name__     # where pickle finds the class again: the module-level names below
    return Held


_Conv2d, _ConvTranspose2d, _BatchNorm2d, _ReLU = (_held(nn.Conv2d), _held(nn.ConvTranspose2d), _held(nn.BatchNorm2d),
                                                  _held(nn.ReLU))


def _res_layer(num_hiddens, num_residual_hiddens):
    return nn.Sequential(
        _ReLU(),
        _Conv2d(num_hiddens, num_residual_hiddens, 3, padding=1),
        _BatchNorm2d(num_residual_hiddens),
        _ReLU(),
        _Conv2d(num_residual_hiddens, num_hiddens, 1),
        _BatchNorm2d(num_hiddens))


class ResidualBlock(nn.Module):
    """output = output + layers[i](output)   (reference vq_vae.py:180-225)."""

    def __init__(self, num_hiddens=128, num_residual_hiddens=512, num_residual_layers=2):
        super(ResidualBlock, self).__init__()
        self.num_hiddens = num_hiddens
        self.num_residual_layers = num_residual_layers
        self.num_residual_hiddens = num_residual_hiddens
        self.layers = nn.ModuleList([_res_layer(num_hiddens, num_residual_hiddens)
                                     for _ in range(num_residual_layers)])

    def _handles(self):
        return [(l[1], l[2], l[4], l[5]) for l in self.layers]

    def forward(self, x):
        _require_gpu(x, "ResidualBlock.forward")
        hs = self._handles()
        params = [p for ca, bna, cb, bnb in hs for p in (ca.weight, ca.bias, bna.weight, bna.bias,
                                                         cb.weight, cb.bias, bnb.weight, bnb.bias)]
        return _ResidualFn.apply(_prep(x), hs, *params)


class _HipEncoder(nn.Sequential):
    """`model.enc`: the nn.Sequential of the reference (same child indices), computed as one HIP pipeline."""

    per_sample_stats = False

    def forward(self, x):
        _require_gpu(x, "VQ_VAE.enc")
        layers = E.Layers(enc=self)
        return _EncoderFn.apply(_prep(x), layers, self.per_sample_stats, *layers.encoder_params())


class _HipDecoder(nn.Sequential):
    """`model.dec`."""

    def forward(self, z):
        _require_gpu(z, "VQ_VAE.dec")
        layers = E.Layers(dec=self)
        dec, _ = _DecoderFn.apply(_prep(z), None, None, layers, *layers.decoder_params())
        return dec


class VQ_VAE(nn.Module):
    """Vector-Quantized VAE with a 16 x 16 x num_hiddens latent (reference vq_vae.py:228-342)."""

    _z16_loss = False

    def __init__(self,
                 num_inputs=2,
                 num_hiddens=16,
                 num_residual_hiddens=32,
                 num_residual_layers=2,
                 num_embeddings=64,
                 commitment_cost=0.25,
                 channel_var=CHANNEL_VAR,
                 weight_recon=1.,
                 weight_commitment=1.,
                 weight_matching=0.005,
                 device="cuda:0",
                 w_a=1.1,
                 w_t=0.1,
                 w_n=-0.5,
                 margin=0.5,
                 **kwargs):
        # callers pass gpu=True (pipeline/patch_VAE.py:431) or alpha/gpu (plot_scripts/recon_loss.py:18);
        # the reference forwards them to nn.Module.__init__ and crashes on current torch -- swallow them.
        for k in ("gpu", "alpha", "extra_loss"):
            kwargs.pop(k, None)
        super(VQ_VAE, self).__init__(**kwargs)
        self.num_inputs = num_inputs
        self.num_hiddens = num_hiddens
        self.num_residual_layers = num_residual_layers
        self.num_residual_hiddens = num_residual_hiddens
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.channel_var = nn.Parameter(
            torch.from_numpy(np.asarray(channel_var, dtype=np.float64)).float().reshape((1, num_inputs, 1, 1)),
            requires_grad=False)
        self.weight_recon = weight_recon
        self.weight_commitment = weight_commitment
        self.weight_matching = weight_matching
        self.w_a, self.w_t, self.w_n, self.margin = w_a, w_t, w_n, margin
        nh = num_hiddens
        self.enc = _HipEncoder(
            _Conv2d(num_inputs, nh // 2, 1),
            _Conv2d(nh // 2, nh // 2, 4, stride=2, padding=1),
            _BatchNorm2d(nh // 2),
            _ReLU(),
            _Conv2d(nh // 2, nh, 4, stride=2, padding=1),
            _BatchNorm2d(nh),
            _ReLU(),
            _Conv2d(nh, nh, 4, stride=2, padding=1),
            _BatchNorm2d(nh),
            _ReLU(),
            _Conv2d(nh, nh, 3, padding=1),
            _BatchNorm2d(nh),
            ResidualBlock(nh, num_residual_hiddens, num_residual_layers))
        self.vq = VectorQuantizer(nh, num_embeddings, commitment_cost=commitment_cost, device=device)
        self.dec = _HipDecoder(
            _ConvTranspose2d(nh, nh // 2, 4, stride=2, padding=1),
            _ReLU(),
            _ConvTranspose2d(nh // 2, nh // 4, 4, stride=2, padding=1),
            _ReLU(),
            _ConvTranspose2d(nh // 4, nh // 4, 4, stride=2, padding=1),
            _ReLU(),
            _Conv2d(nh // 4, num_inputs, 1))

    # ---- the pairwise term (vq_vae.py:324-332; weighted / hinge form of VQ_VAE_z16, vae.py:322-336): one fused op ----
    def _time_matching(self, z_before, time_matching_mat):
        return time_matching_loss(z_before, time_matching_mat, self._z16_loss, getattr(self, "w_a", 0.0),
                                  getattr(self, "w_t", 0.0), getattr(self, "w_n", 0.0), getattr(self, "margin", 0.0))

    def forward(self, inputs, time_matching_mat=None, batch_mask=None):
        """inputs (B, C, H, W) -> (decoded, loss dict)   [vq_vae.py:300-338]"""
        _require_gpu(inputs, "VQ_VAE.forward")
        x = _prep(inputs)
        layers = E.Layers(self)
        z_before = _EncoderFn.apply(x, layers, self.enc.per_sample_stats, *layers.encoder_params())
        z_after, c_loss, perplexity, _ = _VQFn.apply(z_before, self.vq.w.weight, float(self.commitment_cost))
        mask = _prep(batch_mask) if batch_mask is not None else None
        decoded, recon_loss = _DecoderFn.apply(z_after, x, mask, layers, *layers.decoder_params())
        total_loss = self.weight_recon * recon_loss + self.weight_commitment * c_loss
        time_matching_loss = 0.
        if time_matching_mat is not None:
            time_matching_loss = self._time_matching(z_before, time_matching_mat)
            total_loss = total_loss + self.weight_matching * time_matching_loss
        if self._z16_loss:
            return decoded, {'recon_loss': recon_loss, 'commitment_loss': c_loss,
                             'time_matching_loss': time_matching_loss, 'perplexity': perplexity,
                             'total_loss': total_loss}
        return decoded, {'recon_loss': recon_loss, 'commitment_loss': c_loss,
                         'time_matching_loss': time_matching_loss, 'total_loss': total_loss,
                         'perplexity': perplexity}

    def predict(self, inputs):
        """Prediction fn, same as forward pass."""
        return self.forward(inputs)


class VQ_VAE_z16(VQ_VAE):
    """vae.py:216-346: same network and state dict as VQ_VAE; weighted hinge time-matching loss and
    'perplexity' listed before 'total_loss'."""
    _z16_loss = True


# ================================================================================ VQ_VAE_z32
class _Z32StemFn(torch.autograd.Function):
    """enc[0:5] of VQ_VAE_z32: Conv(4,2,1) BN ReLU Conv(4,2,1) BN."""

    @staticmethod
    def forward(ctx, x, mods, *params):
        h, cx = E.z32_stem_forward(*mods, x)
        ctx.mods, ctx.cx, ctx.params = mods, cx, params
        return h

    @staticmethod
    def backward(ctx, g_h):
        G = _GradBag()
        dx = E.z32_stem_backward(*ctx.mods, ctx.cx, g_h, G, want_dx=ctx.needs_input_grad[0])
        ctx.cx = None
        return (dx, None) + G.grads_for(ctx.params, ctx.needs_input_grad[2:])


class _Z32TailFn(torch.autograd.Function):
    """dec[1:5] of VQ_VAE_z32: ConvT BN ReLU ConvT (+ the masked reconstruction loss when x is given)."""

    @staticmethod
    def forward(ctx, r, x, mask, channel_var, mods, *params):
        dec, cx = E.z32_tail_forward(*mods, r, x, mask, channel_var)
        ctx.mods, ctx.cx, ctx.params, ctx.cvar = mods, cx, params, channel_var
        ctx.set_materialize_grads(False)
        if x is None:
            return dec, None
        B, NIN, H, W = x.shape
        recon = ops.loss_finalize(cx.loss_slabs, B * NIN * H * W, torch.zeros(2, device=x.device), 1.0, 0.0)[0]
        return dec, recon

    @staticmethod
    def backward(ctx, g_dec, g_recon):
        gscale = g_recon.reshape(1).contiguous() if g_recon is not None else None
        if g_dec is not None:
            g_dec = g_dec.contiguous()
        if gscale is None and g_dec is None:
            return (None,) * (5 + len(ctx.params))
        G = _GradBag()
        g_x = _recon_input_grad(ctx.cx, ctx.cvar, gscale) if ctx.needs_input_grad[1] else None
        g_r = E.z32_tail_backward(*ctx.mods, ctx.cx, gscale, g_dec, G, want_gr=ctx.needs_input_grad[0])
        ctx.cx = None
        return (g_r, g_x, None, None, None) + G.grads_for(ctx.params, ctx.needs_input_grad[5:])


class _Z32Encoder(nn.Sequential):
    """`model.enc` of VQ_VAE_z32 (same child indices as the reference's nn.Sequential)."""

    def forward(self, x):
        _require_gpu(x, "VQ_VAE_z32.enc")
        mods = (self[0], self[1], self[3], self[4])
        params = [p for m in mods for p in (m.weight, m.bias)]
        h = _Z32StemFn.apply(_prep(x), mods, *params)
        return self[5](h)


class _Z32Decoder(nn.Sequential):
    """`model.dec` of VQ_VAE_z32."""

    def _tail(self, r, x, mask, channel_var):
        mods = (self[1], self[2], self[4])
        params = [p for m in mods for p in (m.weight, m.bias)]
        return _Z32TailFn.apply(r, x, mask, channel_var, mods, *params)

    def forward(self, z):
        _require_gpu(z, "VQ_VAE_z32.dec")
        # (no loss here, so the channel variances are not needed: `model.dec` holds no reference to its parent)
        dec, _ = self._tail(self[0](_prep(z)), None, None, None)
        return dec


class VQ_VAE_z32(nn.Module):
    """Vector-Quantized VAE with a 32 x 32 x num_hiddens latent (reference vae.py:348-474): two stride-2 convs and a
    residual stack in the encoder, a residual stack, BatchNorm and two transposed convs in the decoder, the
    weighted-hinge time-matching loss on z_after.  The default widths (num_hiddens 16, num_residual_hiddens 32) run on the
    register-resident kernels, any other width (config_example.yml: 64 / 64 / 512 codes) on the implicit-GEMM kernels of
    csrc/conv_wide.hip."""

    def __init__(self, num_inputs=2, num_hiddens=16, num_residual_hiddens=32, num_residual_layers=2, num_embeddings=64,
                 commitment_cost=0.25, channel_var=np.ones(2), weight_matching=0.005, w_a=1.1, w_t=0.1, w_n=-0.5,
                 margin=0.5, extra_loss=None, device="cuda:0", **kwargs):
        kwargs.pop("gpu", None)
        alpha = kwargs.pop("alpha", None)
        super(VQ_VAE_z32, self).__init__(**kwargs)
        if extra_loss is not None and not (isinstance(extra_loss, dict) and all(callable(f) for f in extra_loss.values())):
            raise TypeError("extra_loss: None or {loss name: callable(labels, z_after_flat) -> (loss, frac_pos)}  [vae.py:382-383]")
        # vae.py:467 multiplies every extra loss by self.alpha, which the reference's constructor never sets (its docstring
        # lists `alpha`, its signature does not: the attribute has to be assigned by the caller).  Here `alpha=` is accepted
        # and stored; without it the attribute is missing exactly as in the reference.
        if alpha is not None:
            self.alpha = alpha
        self.num_inputs = num_inputs
        self.num_hiddens = num_hiddens
        self.num_residual_layers = num_residual_layers
        self.num_residual_hiddens = num_residual_hiddens
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.channel_var = nn.Parameter(
            torch.from_numpy(np.asarray(channel_var, dtype=np.float64)).float().reshape((1, num_inputs, 1, 1)),
            requires_grad=False)
        self.weight_matching = weight_matching
        self.w_a, self.w_t, self.w_n, self.margin = w_a, w_t, w_n, margin
        nh = num_hiddens
        self.enc = _Z32Encoder(
            _Conv2d(num_inputs, nh // 2, 4, stride=2, padding=1),
            _BatchNorm2d(nh // 2),
            _ReLU(),
            _Conv2d(nh // 2, nh, 4, stride=2, padding=1),
            _BatchNorm2d(nh),
            ResidualBlock(nh, num_residual_hiddens, num_residual_layers))
        self.vq = VectorQuantizer(nh, num_embeddings, commitment_cost=commitment_cost, device=device)
        self.dec = _Z32Decoder(
            ResidualBlock(nh, num_residual_hiddens, num_residual_layers),
            _ConvTranspose2d(nh, nh // 2, 4, stride=2, padding=1),
            _BatchNorm2d(nh // 2),
            _ReLU(),
            _ConvTranspose2d(nh // 2, num_inputs, 4, stride=2, padding=1))
        self.extra_loss = extra_loss

    def forward(self, inputs, labels=None, time_matching_mat=None, batch_mask=None):
        """inputs (B, C, H, W) -> (decoded, loss dict)   [vae.py:430-470].  extra_loss (vae.py:463-469): every
        `loss_fn(labels, z_after.reshape(B, -1)) -> (loss, frac_pos)` is the caller's torch code on the device latents; its
        gradient reaches the encoder through the quantiser's straight-through backward like the time-matching term's."""
        _require_gpu(inputs, "VQ_VAE_z32.forward")
        x = _prep(inputs)
        z_before = self.enc(x)
        z_after, c_loss, perplexity = self.vq(z_before)
        mask = _prep(batch_mask) if batch_mask is not None else None
        decoded, recon_loss = self.dec._tail(self.dec[0](z_after), x, mask, self.channel_var)
        total_loss = recon_loss + c_loss
        if time_matching_mat is not None:
            tml = time_matching_loss(z_after, time_matching_mat, True, self.w_a, self.w_t, self.w_n, self.margin)   # vae.py:441-455
            total_loss = total_loss + tml * self.weight_matching
        else:
            tml = 0
        loss_dict = {'recon_loss': recon_loss, 'commitment_loss': c_loss,
                     'time_matching_loss': tml, 'perplexity': perplexity,
                     'total_loss': total_loss}
        if self.extra_loss is not None:                                   # vae.py:463-469
            z_after_ = z_after.reshape((z_after.shape[0], -1))
            for loss_name, loss_fn in self.extra_loss.items():
                extra_loss, frac_pos = loss_fn(labels, z_after_)
                total_loss = total_loss + extra_loss * self.alpha
                loss_dict['total_loss'] = total_loss
                loss_dict[loss_name] = extra_loss
        return decoded, loss_dict

    def predict(self, inputs):
        """Prediction fn, same as forward pass."""
        return self.forward(inputs)
