"""Kernel orchestration for the VQ-VAE path: which HIP kernel runs on which tensor, in what order.

Every function here only enqueues kernels from libdynamorph_hip.so on the current stream (via
dynamorph_amd.ops); there is no ATen math and no host synchronisation, so a whole training step
can be captured into a HIP graph.

Layer map (reference HiddenStateExtractor/vq_vae.py:276-298, ResidualBlock :203-224):
  enc.0 (1x1) o enc.1 (4x4/s2)  -> ONE 4x4/s2 conv on (x, 1): the composite weight is rebuilt
                                    from the stored parameters every call (dm_e1_compose); the
                                    ones channel carries enc.0's bias through the zero padding.
  enc.2/3, enc.5/6, enc.8/9      -> BatchNorm(+ReLU) folded into the NEXT conv's operand load;
                                    the producing conv's epilogue emits the (sum, sum^2) slabs.
  enc.10, enc.11, enc.12         -> 3x3 conv, BN, residual stack (3x3 -> BN -> ReLU -> 1x1 -> BN, + skip)
  dec.0/2/4 (ConvTranspose 4/2/1)-> 3x3-neighbourhood conv with N = 4 phases x Cout + pixel shuffle
  dec.6 (1x1) + recon loss       -> fused VALU head kernel

Backward mirrors it: each conv layer = one data-gradient kernel (whose epilogue applies the ReLU
mask, joins the residual branch and emits the BatchNorm-backward reductions) + one weight-gradient
kernel; BatchNorm backward itself is an AFFINE2 operand load.  Biases of convs that feed a
train-mode BatchNorm have an identically zero gradient (the batch mean removes them) and are
written as exact zeros.
"""
from types import SimpleNamespace

import torch

from . import ops
from .ops import (DM_LOAD_AFFINE, DM_LOAD_AFFINE2, DM_LOAD_AFFINE_RELU, DM_LOAD_IDENT, DM_LOAD_RELU, Op,
                  weight_view)


# DM_FUSED_BACKWARD=0 in the environment: the two-kernel backward of enc.4 (A/B measurements)
import os as _os
LATENT_TAIL = _os.environ.get("DM_LATENT_TAIL", "1") != "0"
# ... starting one layer earlier, at enc.7: built and tested, but measured slower (C2 0.354 vs 0.320 ms: the 46 KB input
# tile of the strided convolution is staged twice per patch with its loads exposed, and the instantiation spills) -- opt-in
LATENT_TAIL_E7 = _os.environ.get("DM_LATENT_TAIL_E7", "0") == "1"
FUSED_BACKWARD = _os.environ.get("DM_FUSED_BACKWARD", "1") != "0"


class Layers:
    """Parameter handles resolved from a VQ_VAE-shaped nn.Module (state-dict names of the reference), or from one of its
    halves: `model.enc` / `model.dec` called on their own build Layers(enc=self) / Layers(dec=self) -- no back-reference
    to the parent, so copy.deepcopy(model) and pickling give an independent, working module."""

    def __init__(self, model=None, enc=None, dec=None):
        if model is not None:
            enc, dec = model.enc, model.dec
        self.codebook = model.vq.w if model is not None else None
        self.channel_var = model.channel_var if model is not None else None
        if enc is not None:
            self.enc0, self.enc1, self.bn1 = enc[0], enc[1], enc[2]
            self.enc4, self.bn2 = enc[4], enc[5]
            self.enc7, self.bn3 = enc[7], enc[8]
            self.enc10, self.bn4 = enc[10], enc[11]
            self.res = [(l[1], l[2], l[4], l[5]) for l in enc[12].layers]
            self.nin = self.enc0.weight.shape[1]
            self.nh = self.enc10.weight.shape[0]
            self.nrh = self.res[0][0].weight.shape[0] if self.res else 0
        if dec is not None:
            self.dec0, self.dec2, self.dec4, self.dec6 = dec[0], dec[2], dec[4], dec[6]
            if enc is None:
                self.nin, self.nh = self.dec6.weight.shape[0], self.dec0.weight.shape[0]

    def encoder_params(self):
        ps = [self.enc0.weight, self.enc0.bias, self.enc1.weight, self.enc1.bias, self.bn1.weight, self.bn1.bias,
              self.enc4.weight, self.enc4.bias, self.bn2.weight, self.bn2.bias,
              self.enc7.weight, self.enc7.bias, self.bn3.weight, self.bn3.bias,
              self.enc10.weight, self.enc10.bias, self.bn4.weight, self.bn4.bias]
        for ca, bna, cb, bnb in self.res:
            ps += [ca.weight, ca.bias, bna.weight, bna.bias, cb.weight, cb.bias, bnb.weight, bnb.bias]
        return ps

    def decoder_params(self):
        return [self.dec0.weight, self.dec0.bias, self.dec2.weight, self.dec2.bias,
                self.dec4.weight, self.dec4.bias, self.dec6.weight, self.dec6.bias]


# ------------------------------------------------------------------------- BatchNorm glue
def _bn_coef(stats, bn, count, per_sample, nbatch, defer=None):
    """Batch statistics -> (coef, saved).  eval() mode (never used by the reference path) takes the
    running statistics instead; its coefficients are three tiny device ops, no HIP kernel needed.
    defer: list collecting the running-statistics updates of the per-sample path (ops.bn_running_replay)."""
    if not bn.training:
        invstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = bn.weight.detach() * invstd
        coef = torch.stack([scale, torch.zeros_like(scale), bn.bias.detach() - bn.running_mean * scale,
                            torch.zeros_like(scale)], 1).contiguous()
        if per_sample:
            return coef, None
        return coef, torch.stack([bn.running_mean, invstd], 1).contiguous()     # what _bn_backward takes in eval() mode
    momentum = 0.1 if bn.momentum is None else bn.momentum
    spg = stats.shape[0] // nbatch if per_sample else 1
    return ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                           bn.num_batches_tracked, momentum, bn.eps, per_sample=per_sample, slabs_per_group=spg,
                           defer=defer if per_sample else None)


def _w(p):
    return p.detach()


def _bn_backward(stats, count, bn, saved, G):
    """native_batch_norm_backward's reductions for `bn` from the (sum dy, sum dy * a) slabs -> the AFFINE2 coefficients of
    da; eval() mode (fixed statistics: count 0) keeps only da = gamma * invstd * dy."""
    return ops.bn_backward_finalize(stats, count if bn.training else 0, _w(bn.weight), saved, G(bn.weight), G(bn.bias))


# ------------------------------------------------------------------------------- encoder
_side_streams = {}


def _side_stream(device):
    """One helper stream per device for work that is off the critical path (the running-statistics replay)."""
    key = torch.device(device).index
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def e1_operands(L):
    """(weff, bias_border) of the composite enc.0 o enc.1 convolution.  A function of the stored parameters only: the
    inference path builds them once per encode_patches call (weights do not change there), the training path every step."""
    return ops.e1_compose_border(_w(L.enc0.weight), _w(L.enc0.bias), _w(L.enc1.weight), _w(L.enc1.bias))


def encoder_forward(L, x, per_sample=False, e1=None, join=True, latents_only=False, defer_last_join=0):
    """x (B,NIN,H,W) -> z_before (B,nh,H/8,W/8).  per_sample=True normalises every BatchNorm with
    that sample's own statistics = pipeline/patch_VAE.py:445-452 (batch-of-one calls in train mode).
    e1: e1_operands(L) computed by the caller (inference: once per call).
    latents_only (per_sample only): the caller wants z and the running statistics, nothing to differentiate through --
    the 16 x 16 part of the encoder (enc.10 .. enc.12) then is ONE launch that keeps a patch on its CU
    (ops.latent_tail_forward; DM_LATENT_TAIL=0 in the environment keeps the layer-by-layer kernels).
    defer_last_join = K (training step only): when the quantiser can do it (ops.vq_forward_join_supported for K codes), the
    LAST residual join is left to the VectorQuantizer kernel's load path: the return value is then (None, cx) and
    cx.pending_join = (rb, h_in, coef) goes to vq_forward_joined, which also produces z.
    join=False (per_sample only): the running-statistics replay -- a side effect no kernel of the path reads -- is left
    running on a helper stream beside whatever the caller launches next (the VectorQuantizer); the caller MUST call
    cx.join() before it hands the stream back (a HIP-graph capture cannot end with unjoined work)."""
    B, NIN, H, W = x.shape
    nh, nrh, c1 = L.nh, L.nrh, L.nh // 2
    ps = per_sample
    cx = SimpleNamespace(x=x, per_sample=ps, B=B, H=H, W=W, res=[], join=lambda: None)
    defer = [] if ps else None       # per-sample path: running statistics of all eight layers in one launch at the end

    # forward: K = 16*NIN; the ones channel of the composite is folded into a per-position bias table (bias_border)
    weff, border = e1 if e1 is not None else e1_operands(L)
    H1, W1 = H // 2, W // 2
    a1, st = ops.conv4x4s2(Op(x), weight_view(weff, (NIN + 1) * 16, 16, 4, 1), B, NIN, c1, H, W,
                           want_stats=True, bias_border=border, per_tile=ps)
    coef1, saved1 = _bn_coef(st, L.bn1, H1 * W1 * (1 if ps else B), ps, B, defer)

    H2, W2 = H1 // 2, W1 // 2
    a2, st = ops.conv4x4s2(Op(a1, DM_LOAD_AFFINE_RELU, coef1, per_sample=ps), weight_view(_w(L.enc4.weight), c1 * 16, 16, 4, 1),
                           B, c1, nh, H1, W1, want_stats=True, bias=_w(L.enc4.bias), per_tile=ps)
    coef2, saved2 = _bn_coef(st, L.bn2, H2 * W2 * (1 if ps else B), ps, B, defer)

    H3, W3 = H2 // 2, W2 // 2
    n3 = H3 * W3 * (1 if ps else B)
    fuse_tail = (ps and latents_only and LATENT_TAIL and L.bn3.training and L.bn4.training
                 and all(bna.training and bnb.training for _, bna, _, bnb in L.res)
                 and ops.latent_tail_supported(nh, nrh, H3, W3, len(L.res)))
    mom = lambda bn: 0.1 if bn.momentum is None else bn.momentum
    res_args = lambda: [(_w(ca.weight), _w(ca.bias), _w(bna.weight), _w(bna.bias), bna.eps, _w(cb.weight), _w(cb.bias),
                         _w(bnb.weight), _w(bnb.bias), bnb.eps) for ca, bna, cb, bnb in L.res]

    def defer_tail(st4, sts):
        defer.append((st4, 1, n3, L.bn4.running_mean, L.bn4.running_var, L.bn4.num_batches_tracked, mom(L.bn4)))
        for (sa, sb), (_, bna, _, bnb) in zip(sts, L.res):
            defer.append((sa, 1, n3, bna.running_mean, bna.running_var, bna.num_batches_tracked, mom(bna)))
            defer.append((sb, 1, n3, bnb.running_mean, bnb.running_var, bnb.num_batches_tracked, mom(bnb)))
    if fuse_tail and LATENT_TAIL_E7:
        # enc.7 .. enc.12 of a patch in one workgroup, from a2 (csrc/latent_tail.hip, E7 form)
        z, st4, sts, st3 = ops.latent_tail_forward(
            None, None, _w(L.enc10.weight), _w(L.enc10.bias), _w(L.bn4.weight), _w(L.bn4.bias), L.bn4.eps, res_args(),
            enc7=(a2, coef2, _w(L.enc7.weight), _w(L.enc7.bias), _w(L.bn3.weight), _w(L.bn3.bias), L.bn3.eps))
        defer.append((st3, 1, n3, L.bn3.running_mean, L.bn3.running_var, L.bn3.num_batches_tracked, mom(L.bn3)))
        defer_tail(st4, sts)
        cx.__dict__.update(a1=a1, a2=a2, coef1=coef1, coef2=coef2, saved1=None, dims=(H1, W1, H2, W2, H3, W3))
        _replay(cx, x, defer, join)
        return z, cx
    a3, st = ops.conv4x4s2(Op(a2, DM_LOAD_AFFINE_RELU, coef2, per_sample=ps), weight_view(_w(L.enc7.weight), nh * 16, 16, 4, 1),
                           B, nh, nh, H2, W2, want_stats=True, bias=_w(L.enc7.bias), per_tile=ps)
    coef3, saved3 = _bn_coef(st, L.bn3, n3, ps, B, defer)

    if fuse_tail:
        z, st4, sts = ops.latent_tail_forward(a3, coef3, _w(L.enc10.weight), _w(L.enc10.bias), _w(L.bn4.weight), _w(L.bn4.bias),
                                              L.bn4.eps, res_args())
        defer_tail(st4, sts)
        cx.__dict__.update(a1=a1, a2=a2, a3=a3, coef1=coef1, coef2=coef2, coef3=coef3, saved1=None, dims=(H1, W1, H2, W2, H3, W3))
        _replay(cx, x, defer, join)
        return z, cx
    a4, st = ops.conv3x3(Op(a3, DM_LOAD_AFFINE_RELU, coef3, per_sample=ps), weight_view(_w(L.enc10.weight), nh * 9, 9, 3, 1),
                         B, nh, nh, H3, W3, taps=9, want_stats=True, bias=_w(L.enc10.bias), per_tile=ps)
    coef4, saved4 = _bn_coef(st, L.bn4, n3, ps, B, defer)
    h = ops.apply(Op(a4, DM_LOAD_AFFINE, coef4, per_sample=ps), B, nh, H3, W3)

    cx.__dict__.update(a1=a1, a2=a2, a3=a3, a4=a4, coef1=coef1, coef2=coef2, coef3=coef3, coef4=coef4,
                       saved1=saved1, saved2=saved2, saved3=saved3, saved4=saved4, dims=(H1, W1, H2, W2, H3, W3))
    fuse = bool(defer_last_join) and not ps and len(L.res) > 0 and L.res[-1][3].training and \
        ops.vq_forward_join_supported(nh, int(defer_last_join), H3, W3)
    z, cx.res = residual_forward(L.res, h, ps, defer, defer_last_join=fuse)
    cx.pending_join = (cx.res[-1].rb, cx.res[-1].h_in, cx.res[-1].coefb) if fuse else None
    _replay(cx, x, defer, join)
    return z, cx


def _replay(cx, x, defer, join):
    """The deferred running-statistics updates of the per-sample path, one launch on the helper stream."""
    if defer:
        cur = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for seg in defer:
                seg[0].record_stream(side)                 # the statistics slabs are read on the helper stream
            ops.bn_running_replay(defer)
        cx.join = lambda: cur.wait_stream(side)
        if join:
            cx.join()


def residual_forward(res_layers, h, per_sample=False, defer=None, defer_last_join=False):
    """ResidualBlock.forward (vq_vae.py:212-225) on a materialised h (B,nh,H,W).
    defer: the caller's list of postponed running-statistics updates (per-sample path); None: flushed here.
    defer_last_join: the last layer's `h + BN(rb)` is NOT launched (the caller fuses it into its consumer); returns None
    for the output then."""
    B, nh, H, W = h.shape
    n = H * W * (1 if per_sample else B)
    saved = []
    own = defer is None and per_sample
    if own:
        defer = []
    for ca, bna, cb, bnb in res_layers:
        nrh = ca.weight.shape[0]
        ra, st = ops.conv3x3(Op(h, DM_LOAD_RELU), weight_view(_w(ca.weight), nh * 9, 9, 3, 1), B, nh, nrh, H, W, taps=9,
                             want_stats=True, bias=_w(ca.bias), per_tile=per_sample)
        coefa, saveda = _bn_coef(st, bna, n, per_sample, B, defer)
        rb, st = ops.conv3x3(Op(ra, DM_LOAD_AFFINE_RELU, coefa, per_sample=per_sample), weight_view(_w(cb.weight), nrh, 1, 0, 0),
                             B, nrh, nh, H, W, taps=1, want_stats=True, bias=_w(cb.bias), per_tile=per_sample)
        coefb, savedb = _bn_coef(st, bnb, n, per_sample, B, defer)
        last = defer_last_join and ca is res_layers[-1][0]
        hn = None if last else ops.apply(Op(rb, DM_LOAD_AFFINE, coefb, per_sample=per_sample), B, nh, H, W, resid=h)
        saved.append(SimpleNamespace(h_in=h, ra=ra, rb=rb, coefa=coefa, saveda=saveda, coefb=coefb, savedb=savedb))
        h = hn
    if own and defer:
        ops.bn_running_replay(defer)
    return h, saved


def _fed_bias(gbias, bn, coef_bwd, G, zero=True):
    """Gradient of a convolution bias that feeds the BatchNorm `bn`: identically zero in train mode (the batch mean removes a
    constant; zero=False: the caller's buffer holds the zero already); in eval() mode the sum of the BatchNorm's input
    gradient over the positions, gamma / sqrt(running_var + eps) * dbeta."""
    if bn.training:
        if zero:
            gbias.zero_()
    else:
        torch.mul(coef_bwd[:, 0], G(bn.bias), out=gbias)


def _zero(t, do=True):
    if do:
        t.zero_()


def residual_backward(res_layers, saved, g_h, G, q_below, pending=None, zero_fed_biases=True):
    """Gradient of the residual stack.  g_h: gradient w.r.t. its output.  q_below: the raw conv output
    whose BatchNorm produced the stack's input (None if there is none); when given, the returned stats
    are the (sum g, sum g*q_below) slabs that BatchNorm's backward needs.  Returns (g_in, stats)."""
    if not saved:
        return g_h, (ops.channel_stats(g_h, q_below) if q_below is not None else None)
    B, nh, H, W = g_h.shape
    cnt = B * H * W
    stats = ops.channel_stats(g_h, saved[-1].rb)
    for i in range(len(saved) - 1, -1, -1):
        ca, bna, cb, bnb = res_layers[i]
        s = saved[i]
        nrh = ca.weight.shape[0]
        if s.savedb is None or s.coefb.dim() != 2:
            raise NotImplementedError("backward needs batch-statistics BatchNorm (train mode, per_sample=False)")
        cb_bwd = _bn_backward(stats, cnt, bnb, s.savedb, G)
        da_rb = Op(g_h, DM_LOAD_AFFINE2, cb_bwd, p1=s.rb)
        _fed_bias(G(cb.bias), bnb, cb_bwd, G, zero_fed_biases)
        if FUSED_BACKWARD and s.coefa.dim() == 2 and ops.conv1x1_bwd_fused_supported(nh, nrh, H, W):
            # the 1x1 convolution's data and weight gradient from ONE staging of (g_h, rb, ra) -- csrc/conv1x1_bwd.hip
            dy_ra, st = ops.conv1x1_bwd_fused(da_rb, s.ra, s.coefa, _w(cb.weight), G(cb.weight), B, nh, nrh, H, W, pending=pending)
        else:
            ops.wgrad(da_rb, Op(s.ra, DM_LOAD_AFFINE_RELU, s.coefa), G(cb.weight), B, nh, nrh, H, W, 1, pending=pending)
            dy_ra, st = ops.conv3x3(da_rb, weight_view(_w(cb.weight), 1, nrh, 0, 0), B, nh, nrh, H, W, taps=1, want_stats=True,
                                    like=g_h, mask=Op(s.ra, DM_LOAD_AFFINE, s.coefa), stat_q=s.ra)
        ca_bwd = _bn_backward(st, cnt, bna, s.saveda, G)
        da_ra = Op(dy_ra, DM_LOAD_AFFINE2, ca_bwd, p1=s.ra)
        _fed_bias(G(ca.bias), bna, ca_bwd, G, zero_fed_biases)
        q = saved[i - 1].rb if i > 0 else q_below
        if FUSED_BACKWARD and ops.conv3x3_bwd_fused_supported(nrh, nh, H, W):
            # the 3x3 convolution's data and weight gradient from ONE staging of the patch -- csrc/conv3x3_bwd.hip
            g_h, stats = ops.conv3x3_bwd_fused(da_ra, s.h_in, None, _w(ca.weight), G(ca.weight), B, nrh, resid=g_h, q=q,
                                               want_stats=q is not None, pending=pending)
        else:
            ops.wgrad(da_ra, Op(s.h_in, DM_LOAD_RELU), G(ca.weight), B, nrh, nh, H, W, 3, pending=pending)
            g_h, stats = ops.conv3x3(da_ra, weight_view(_w(ca.weight), 9, nh * 9, -3, -1, off=8), B, nrh, nh, H, W, taps=9,
                                     want_stats=q is not None, like=g_h, mask=Op(s.h_in), resid=g_h, stat_q=q)
    return g_h, stats


def encoder_backward(L, cx, g_z, G, zero_fed_biases=True, pending_extra=(), want_dx=False):
    """Accumulates nothing: every parameter gradient G(p) is overwritten.  want_dx: also returns the gradient w.r.t. the input
    patches (the reference's training never asks for it; autograd there would answer) -- the data gradient of the composite
    enc.0 o enc.1 convolution, a transposed convolution with the composite weights' image channels.
    zero_fed_biases=False skips writing the (identically zero) gradients of the conv biases that feed a
    BatchNorm -- for callers whose gradient buffer is zero there already (FusedTrainer)."""
    pending = list(pending_extra)                    # (slabs, dst) pairs that ride along in the one slab reduction
    if cx.per_sample and cx.B > 1:
        raise NotImplementedError("backward through per-sample BatchNorm statistics with B > 1")
    B, x = cx.B, cx.x
    NIN, nh, c1 = L.nin, L.nh, L.nh // 2
    H1, W1, H2, W2, H3, W3 = cx.dims
    g_z = g_z.contiguous()

    g_h, stats = residual_backward(L.res, cx.res, g_z, G, cx.a4, pending=pending, zero_fed_biases=zero_fed_biases)
    cnt3 = B * H3 * W3
    c4b = _bn_backward(stats, cnt3, L.bn4, cx.saved4, G)
    da4 = Op(g_h, DM_LOAD_AFFINE2, c4b, p1=cx.a4)
    _fed_bias(G(L.enc10.bias), L.bn4, c4b, G, zero_fed_biases)
    if FUSED_BACKWARD and cx.coef3.dim() == 2 and ops.conv3x3_bwd_fused_supported(nh, nh, H3, W3):
        # enc.10: data and weight gradient from ONE staging of the patch -- csrc/conv3x3_bwd.hip
        dy3, st = ops.conv3x3_bwd_fused(da4, cx.a3, cx.coef3, _w(L.enc10.weight), G(L.enc10.weight), B, nh, q=cx.a3,
                                        pending=pending)
    else:
        ops.wgrad(da4, Op(cx.a3, DM_LOAD_AFFINE_RELU, cx.coef3), G(L.enc10.weight), B, nh, nh, H3, W3, 3, pending=pending)
        dy3, st = ops.conv3x3(da4, weight_view(_w(L.enc10.weight), 9, nh * 9, -3, -1, off=8), B, nh, nh, H3, W3, taps=9,
                              want_stats=True, like=g_h, mask=Op(cx.a3, DM_LOAD_AFFINE, cx.coef3), stat_q=cx.a3)

    c3b = _bn_backward(st, cnt3, L.bn3, cx.saved3, G)
    da3 = Op(dy3, DM_LOAD_AFFINE2, c3b, p1=cx.a3)
    _fed_bias(G(L.enc7.bias), L.bn3, c3b, G, zero_fed_biases)
    if FUSED_BACKWARD and cx.coef2.dim() == 2 and ops.conv4x4s2_bwd_fused_supported(nh, nh, H3, W3):
        # enc.7: data and weight gradient from ONE staging of the patch -- csrc/conv4x4s2_patch.hip
        dy2, st = ops.conv4x4s2_bwd_fused(da3, cx.a2, cx.coef2, _w(L.enc7.weight), G(L.enc7.weight), B, pending=pending)
    else:
        ops.wgrad(da3, Op(cx.a2, DM_LOAD_AFFINE_RELU, cx.coef2), G(L.enc7.weight), B, nh, nh, H3, W3, 4, pending=pending)
        dy2, st = ops.conv3x3(da3, weight_view(_w(L.enc7.weight), 16, nh * 16, 4, 1), B, nh, 4 * nh, H3, W3, taps=9,
                              pixel_shuffle=True, want_stats=True, like=g_h, mask=Op(cx.a2, DM_LOAD_AFFINE, cx.coef2),
                              stat_q=cx.a2)

    c2b = _bn_backward(st, B * H2 * W2, L.bn2, cx.saved2, G)
    da2 = Op(dy2, DM_LOAD_AFFINE2, c2b, p1=cx.a2)
    _fed_bias(G(L.enc4.bias), L.bn2, c2b, G, zero_fed_biases)
    if FUSED_BACKWARD and ops.conv_bwd_s2_fused_supported(nh, c1, H2, W2):
        # enc.4: data gradient + weight gradient from ONE staging of (dy2, a2, a1) -- csrc/conv_mfma.hip, kernel D
        dy1, st = ops.conv_bwd_s2_fused(da2, Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), weight_view(_w(L.enc4.weight), 16, c1 * 16, 4, 1),
                                        G(L.enc4.weight), B, nh, c1, H2, W2, mask=Op(cx.a1, DM_LOAD_AFFINE, cx.coef1),
                                        stat_q=cx.a1, pending=pending)
    else:
        ops.wgrad(da2, Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), G(L.enc4.weight), B, nh, c1, H2, W2, 4, pending=pending)
        dy1, st = ops.conv3x3(da2, weight_view(_w(L.enc4.weight), 16, c1 * 16, 4, 1), B, nh, 4 * c1, H2, W2, taps=9,
                              pixel_shuffle=True, want_stats=True, like=g_h, mask=Op(cx.a1, DM_LOAD_AFFINE, cx.coef1),
                              stat_q=cx.a1)

    c1b = _bn_backward(st, B * H1 * W1, L.bn1, cx.saved1, G)
    da1 = Op(dy1, DM_LOAD_AFFINE2, c1b, p1=cx.a1)
    dweff = torch.empty((c1, NIN + 1, 4, 4), device=x.device, dtype=torch.float32)
    ops.wgrad(da1, Op(x, ones=True), dweff, B, c1, NIN + 1, H1, W1, 4, pending=pending)
    ops.reduce_slabs_multi(pending)                  # all encoder weight gradients in one launch
    ops.e1_chain(dweff, _w(L.enc0.weight), _w(L.enc0.bias), _w(L.enc1.weight),
                 G(L.enc0.weight), G(L.enc0.bias), G(L.enc1.weight))
    _fed_bias(G(L.enc1.bias), L.bn1, c1b, G, zero_fed_biases)
    if want_dx:
        # a1 = conv(x, Weff[:, :NIN]) + border bias: dx = ConvTranspose(da1, Weff[:, :NIN]) (phase-decomposed kernel family;
        # the ones channel of the composite carries no gradient to x)
        weff, _ = e1_operands(L)
        dx, _ = ops.conv3x3(da1, weight_view(weff, 16, (NIN + 1) * 16, 4, 1), B, c1, 4 * NIN, H1, W1, taps=9, pixel_shuffle=True)
        return dx
    return None


# ------------------------------------------------------------------------------------ VQ
def vq_forward(codebook, z, commitment_cost, want_out=True, defer_scalars=False, want_scalars=True):
    """defer_scalars (training pass): no scalar launches here; the third return value is the state
    ops.vq_loss_finalize needs to produce them together with the reconstruction loss at the end of the step.
    want_scalars=False (inference latents, patch_VAE.py:445-452 discards loss and perplexity): no counter reduction and no
    scalar launch at all; the third return value is None."""
    B, D, H, W = z.shape
    if not want_scalars:
        idx, out, _, _ = ops.vq_forward(z, _w(codebook), want_out=want_out, want_hist=False)
        return out, idx, None
    if defer_scalars:
        idx, out, slabs, ws = ops.vq_forward(z, _w(codebook), want_out=want_out, want_hist=False)
        return out, idx, SimpleNamespace(slabs=slabs, ws=ws, K=codebook.shape[0], D=D, positions=B * H * W, cc=commitment_cost)
    idx, out, slabs, hist = ops.vq_forward(z, _w(codebook), want_out=want_out)
    scalars = ops.vq_finalize(slabs, hist, B * H * W, D, commitment_cost)     # (loss, perplexity, mse)
    return out, idx, scalars


def vq_forward_joined(codebook, pending_join, commitment_cost):
    """encoder_forward(..., defer_last_join=K) left the last residual join to the quantiser: one launch forms z, quantises
    it and writes both.  Returns (z, out, idx, state for ops.vq_loss_finalize) -- vq_forward(defer_scalars=True)'s contract
    plus the latents."""
    rb, h_in, coef = pending_join
    B, D, H, W = rb.shape
    idx, out, slabs, ws, z = ops.vq_forward_join(rb, h_in, coef, _w(codebook))
    return z, out, idx, SimpleNamespace(slabs=slabs, ws=ws, K=codebook.shape[0], D=D, positions=B * H * W, cc=commitment_cost)


# -------------------------------------------------------------------------------- decoder
def decoder_forward(L, zq, x=None, mask=None, defer_tail=False):
    """defer_tail (training pass only): when the fused tail is available, leave dec.4/dec.6 and the reconstruction
    loss to decoder_backward, which then runs them together with their backward in one kernel
    (ops.dec_tail_train); `decoded` is not produced and cx.loss_slabs is filled by decoder_backward."""
    B, nh, H3, W3 = zq.shape
    c1, c2 = nh // 2, nh // 4
    d0, _ = ops.conv3x3(Op(zq), weight_view(_w(L.dec0.weight), 16, c1 * 16, 4, 1), B, nh, 4 * c1, H3, W3, taps=9,
                        pixel_shuffle=True, bias=_w(L.dec0.bias), relu=True)
    d2, _ = ops.conv3x3(Op(d0), weight_view(_w(L.dec2.weight), 16, c2 * 16, 4, 1), B, c1, 4 * c2, 2 * H3, 2 * W3, taps=9,
                        pixel_shuffle=True, bias=_w(L.dec2.bias), relu=True)
    # (model.dec called on its own has no loss and hence no channel variances)
    var = _w(L.channel_var).reshape(-1) if L.channel_var is not None else torch.ones(L.dec6.weight.shape[0], device=zq.device)
    fused = ops.dec_tail_supported(c2, L.dec6.weight.shape[0], 4 * H3, 4 * W3)
    if fused and defer_tail and x is not None:
        cx = SimpleNamespace(zq=zq, d0=d0, d2=d2, d4=None, dec=None, x=x, mask=mask, loss_slabs=None, deferred=True)
        return None, cx
    if fused:
        # dec.4 + ReLU + dec.6 (+ loss) in one kernel; the 4 x 128 x 128 tensor d4 is never stored
        d4 = None
        dec, slabs = ops.dec_tail_forward(d2, _w(L.dec4.weight), _w(L.dec4.bias), _w(L.dec6.weight), _w(L.dec6.bias),
                                          x, mask, var)
    else:
        d4 = _dec4_forward(L, d2)
        if ops.head_supported(c2, L.dec6.weight.shape[0]):
            dec, slabs = ops.head_forward(d4, _w(L.dec6.weight), _w(L.dec6.bias), x, mask, var)
        else:
            # widths without a fused head: dec.6 as a plain 1x1 convolution, the loss as its own pass
            nin = L.dec6.weight.shape[0]
            dec, _ = ops.conv3x3(Op(d4), weight_view(_w(L.dec6.weight), c2, 1, 0, 0), B, c2, nin, 8 * H3, 8 * W3, taps=1,
                                 bias=_w(L.dec6.bias))
            slabs = ops.recon_loss(dec, x, mask, var) if x is not None else None
    cx = SimpleNamespace(zq=zq, d0=d0, d2=d2, d4=d4, dec=dec, x=x, mask=mask, loss_slabs=slabs, deferred=False)
    return dec, cx


def _dec4_forward(L, d2):
    B, c2, H, W = d2.shape
    d4, _ = ops.conv3x3(Op(d2), weight_view(_w(L.dec4.weight), 16, c2 * 16, 4, 1), B, c2, 4 * c2, H, W, taps=9,
                        pixel_shuffle=True, bias=_w(L.dec4.bias), relu=True)
    return d4


def _head_backward_unfused(L, cx, d4, var, gscale, gdec_ext, G, pending):
    """dec.6 backward for widths the fused head is not built for: loss gradient, bias sums, the 1x1 weight gradient
    and the input gradient (masked by dec.4's ReLU, its channel sums = dec.4's bias gradient) as separate launches."""
    B, c2, H, W = d4.shape
    NIN = L.dec6.weight.shape[0]
    if gscale is not None:
        g, part = ops.recon_loss_backward(cx.dec, cx.x, cx.mask, var, gscale)
        if gdec_ext is not None:
            g, part = g + gdec_ext, None
    else:
        g, part = gdec_ext.contiguous(), None
    ops.sum_slabs(part if part is not None else ops.channel_stats(g), G(L.dec6.bias))
    ops.wgrad(Op(g), Op(d4), G(L.dec6.weight), B, NIN, c2, H, W, 1, pending=pending)
    g4, st = ops.conv3x3(Op(g), weight_view(_w(L.dec6.weight), 1, c2, 0, 0), B, NIN, c2, H, W, taps=1, want_stats=True,
                         mask=Op(d4))
    ops.sum_slabs(st, G(L.dec4.bias))
    return g4


def decoder_backward(L, cx, gscale, gdec_ext, G, want_gz=True, pending=None):
    """gscale: 1-element device tensor = d(total)/d(recon_loss) (None: no loss term);
    gdec_ext: upstream gradient w.r.t. decoded (None: none).
    pending: the caller's list of (slabs, dst) pairs for ONE slab reduction of the whole backward pass -- the decoder's pairs
    are appended and NOT reduced here (FusedTrainer: they ride in the encoder's launch); None: reduced here."""
    zq = cx.zq
    B, nh, H3, W3 = zq.shape
    c1, c2 = nh // 2, nh // 4
    NIN = L.dec6.weight.shape[0]
    var = _w(L.channel_var).reshape(-1) if L.channel_var is not None else torch.ones(NIN, device=zq.device)
    own_pending = pending is None
    if own_pending:
        pending = []
    if cx.deferred:
        if gdec_ext is not None or gscale is None:
            raise ValueError("decoder_backward: a deferred tail takes the reconstruction-loss gradient only")
        g2, part, wsl, cx.loss_slabs = ops.dec_tail_train(cx.d2, _w(L.dec4.weight), _w(L.dec4.bias), _w(L.dec6.weight),
                                                          _w(L.dec6.bias), cx.x, cx.mask, var, gscale)
        ops.pend_stats(pending, part, [G(L.dec6.weight), G(L.dec6.bias), G(L.dec4.bias), G(L.dec2.bias)])
        pending.append((wsl, G(L.dec4.weight)))
    elif cx.d4 is None and gdec_ext is None and gscale is not None:
        # fused tail: recompute d4 from d2, g4 lives only in LDS
        g2, part, wsl = ops.dec_tail_backward(cx.d2, _w(L.dec4.weight), _w(L.dec4.bias), _w(L.dec6.weight), cx.dec, cx.x,
                                              cx.mask, var, gscale)
        ops.pend_stats(pending, part, [G(L.dec6.weight), G(L.dec6.bias), G(L.dec4.bias), G(L.dec2.bias)])
        pending.append((wsl, G(L.dec4.weight)))
    else:
        d4 = cx.d4 if cx.d4 is not None else _dec4_forward(L, cx.d2)
        if ops.head_supported(c2, NIN):
            g4, part = ops.head_backward(cx.dec, cx.x, cx.mask, var, d4, _w(L.dec6.weight), gscale, gdec_ext)
            ops.pend_stats(pending, part, [G(L.dec6.weight), G(L.dec6.bias), G(L.dec4.bias)])
        else:
            g4 = _head_backward_unfused(L, cx, d4, var, gscale, gdec_ext, G, pending)
        ops.wgrad(Op(cx.d2), Op(g4), G(L.dec4.weight), B, c2, c2, 4 * H3, 4 * W3, 4, pending=pending)
        g2, st = ops.conv4x4s2(Op(g4), weight_view(_w(L.dec4.weight), c2 * 16, 16, 4, 1), B, c2, c2, 8 * H3, 8 * W3,
                               want_stats=True, mask=Op(cx.d2))
        ops.pend_stats(pending, st, [G(L.dec2.bias)])
    g2 = g2.contiguous()
    if FUSED_BACKWARD and ops.convT_bwd_fused_supported(c1, c2, 2 * H3, 2 * W3):
        # dec.2: input and weight gradient from ONE staging of (d0, g2) -- csrc/convT_bwd.hip
        g0, st = ops.convT_bwd_fused(cx.d0, g2, _w(L.dec2.weight), G(L.dec2.weight), mask_relu=True, want_stats=True,
                                     pending=pending)
    else:
        ops.wgrad(Op(cx.d0), Op(g2), G(L.dec2.weight), B, c1, c2, 2 * H3, 2 * W3, 4, pending=pending)
        g0, st = ops.conv4x4s2(Op(g2), weight_view(_w(L.dec2.weight), c2 * 16, 16, 4, 1), B, c2, c1, 4 * H3, 4 * W3,
                               want_stats=True, mask=Op(cx.d0))
    ops.pend_stats(pending, st, [G(L.dec0.bias)])         # bias gradients ride in the one slab reduction below
    g_zq = None
    if want_gz and FUSED_BACKWARD and ops.convT_bwd_fused_supported(nh, c1, H3, W3):
        g_zq, _ = ops.convT_bwd_fused(zq.contiguous(), g0, _w(L.dec0.weight), G(L.dec0.weight), pending=pending)     # dec.0 likewise
    else:
        ops.wgrad(Op(zq), Op(g0), G(L.dec0.weight), B, nh, c1, H3, W3, 4, pending=pending)
    if own_pending:
        ops.reduce_slabs_multi(pending)              # all decoder weight gradients in one launch
    if not want_gz:
        return None
    if g_zq is None:
        g_zq, _ = ops.conv4x4s2(Op(g0), weight_view(_w(L.dec0.weight), c1 * 16, 16, 4, 1), B, c1, nh, 2 * H3, 2 * W3)
    return g_zq


# ======================================================================== VQ_VAE_z32 (vae.py:348-474)
# 32x32 latent: enc = Conv(4,2,1) BN ReLU Conv(4,2,1) BN ResidualBlock; dec = ResidualBlock ConvT BN ReLU ConvT.
# The residual stacks are the ResidualBlock module itself (residual_forward/backward); these four functions are the
# two conv "stems" around them.  Same conventions as above: raw conv outputs + statistics slabs, BatchNorm applied by
# the consumer's operand load, BatchNorm backward as an AFFINE2 operand.
def z32_stem_forward(conv0, bn0, conv1, bn1, x, per_sample=False):
    """x (B,NIN,H,W) -> h = BN(conv1(relu(BN(conv0(x))))) (B,nh,H/4,W/4), materialised for the residual stack.
    per_sample=True: every BatchNorm uses that sample's own statistics (process_VAE's batch-of-one calls, batched)."""
    B, NIN, H, W = x.shape
    ps = per_sample
    c1, nh = conv0.weight.shape[0], conv1.weight.shape[0]
    H1, W1, H2, W2 = H // 2, W // 2, H // 4, W // 4
    a1, st = ops.conv4x4s2(Op(x), weight_view(_w(conv0.weight), NIN * 16, 16, 4, 1), B, NIN, c1, H, W,
                           want_stats=True, bias=_w(conv0.bias), per_tile=ps)
    coef1, saved1 = _bn_coef(st, bn0, H1 * W1 * (1 if ps else B), ps, B)
    a2, st = ops.conv4x4s2(Op(a1, DM_LOAD_AFFINE_RELU, coef1, per_sample=ps), weight_view(_w(conv1.weight), c1 * 16, 16, 4, 1),
                           B, c1, nh, H1, W1, want_stats=True, bias=_w(conv1.bias), per_tile=ps)
    coef2, saved2 = _bn_coef(st, bn1, H2 * W2 * (1 if ps else B), ps, B)
    h = ops.apply(Op(a2, DM_LOAD_AFFINE, coef2, per_sample=ps), B, nh, H2, W2)
    cx = SimpleNamespace(x=x, a1=a1, a2=a2, coef1=coef1, coef2=coef2, saved1=saved1, saved2=saved2, per_sample=ps,
                         dims=(B, NIN, c1, nh, H1, W1, H2, W2))
    return h, cx


def z32_stem_backward(conv0, bn0, conv1, bn1, cx, g_h, G, stats=None, pending=None, zero_fed_biases=True, want_dx=False):
    """stats: the (sum g, sum g*a2) slabs when the caller's last kernel already produced them (residual_backward with
    q_below = cx.a2); pending: the caller's list for ONE slab reduction of the whole backward pass (None: reduced here);
    zero_fed_biases=False: the gradient buffer is zero already (FusedTrainer) -- the identically zero gradients of the
    conv biases that feed a BatchNorm are not written."""
    if getattr(cx, "per_sample", False) and cx.dims[0] > 1:
        raise NotImplementedError("backward through per-sample BatchNorm statistics with B > 1")
    B, NIN, c1, nh, H1, W1, H2, W2 = cx.dims
    g_h = g_h.contiguous()
    own = pending is None
    if own:
        pending = []
    if stats is None:
        stats = ops.channel_stats(g_h, cx.a2)
    c2b = _bn_backward(stats, B * H2 * W2, bn1, cx.saved2, G)
    da2 = Op(g_h, DM_LOAD_AFFINE2, c2b, p1=cx.a2)
    ops.wgrad(da2, Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), G(conv1.weight), B, nh, c1, H2, W2, 4, pending=pending)
    _fed_bias(G(conv1.bias), bn1, c2b, G, zero_fed_biases)
    dy1, st = ops.conv3x3(da2, weight_view(_w(conv1.weight), 16, c1 * 16, 4, 1), B, nh, 4 * c1, H2, W2, taps=9,
                          pixel_shuffle=True, want_stats=True, like=g_h, mask=Op(cx.a1, DM_LOAD_AFFINE, cx.coef1),
                          stat_q=cx.a1)
    c1b = _bn_backward(st, B * H1 * W1, bn0, cx.saved1, G)
    da1 = Op(dy1, DM_LOAD_AFFINE2, c1b, p1=cx.a1)
    ops.wgrad(da1, Op(cx.x), G(conv0.weight), B, c1, NIN, H1, W1, 4, pending=pending)
    _fed_bias(G(conv0.bias), bn0, c1b, G, zero_fed_biases)
    if own:
        ops.reduce_slabs_multi(pending)
    if want_dx:          # the gradient w.r.t. the input patches: conv0's data gradient (see encoder_backward)
        dx, _ = ops.conv3x3(da1, weight_view(_w(conv0.weight), 16, NIN * 16, 4, 1), B, c1, 4 * NIN, H1, W1, taps=9, pixel_shuffle=True)
        return dx
    return None


def z32_tail_forward(up0, bn, up1, r, x, mask, channel_var):
    """r (B,nh,H2,W2) -> decoded (B,NIN,4*H2,4*W2) (+ loss slabs when x is given)."""
    B, nh, H2, W2 = r.shape
    c1, NIN = up0.weight.shape[1], up1.weight.shape[1]
    d1, st = ops.conv3x3(Op(r), weight_view(_w(up0.weight), 16, c1 * 16, 4, 1), B, nh, 4 * c1, H2, W2, taps=9,
                         pixel_shuffle=True, want_stats=True, bias=_w(up0.bias))
    coefd, savedd = _bn_coef(st, bn, B * 4 * H2 * W2, False, B)
    dec, _ = ops.conv3x3(Op(d1, DM_LOAD_AFFINE_RELU, coefd), weight_view(_w(up1.weight), 16, NIN * 16, 4, 1), B, c1,
                         4 * NIN, 2 * H2, 2 * W2, taps=9, pixel_shuffle=True, bias=_w(up1.bias))
    var = _w(channel_var).reshape(-1) if channel_var is not None else torch.ones(NIN, device=r.device)
    slabs = ops.recon_loss(dec, x, mask, var) if x is not None else None
    cx = SimpleNamespace(r=r, d1=d1, coefd=coefd, savedd=savedd, dec=dec, x=x, mask=mask, var=var, loss_slabs=slabs,
                         dims=(B, nh, c1, NIN, H2, W2))
    return dec, cx


def z32_tail_backward(up0, bn, up1, cx, gscale, gdec_ext, G, want_gr=True, pending=None, zero_fed_biases=True):
    """gscale: 1-element device tensor d(total)/d(recon_loss) or None; gdec_ext: upstream gradient w.r.t. decoded or None.
    pending / zero_fed_biases: as in z32_stem_backward."""
    B, nh, c1, NIN, H2, W2 = cx.dims
    own = pending is None
    if own:
        pending = []
    if gscale is not None:
        g, part = ops.recon_loss_backward(cx.dec, cx.x, cx.mask, cx.var, gscale)
        if gdec_ext is not None:
            g = g + gdec_ext
            part = None
    else:
        g, part = gdec_ext.contiguous(), None
    # the last layer's bias gradient (channel sums of g) rides in the slab reduction
    ops.pend_stats(pending, part if part is not None else ops.channel_stats(g), [G(up1.bias)])
    ops.wgrad(Op(cx.d1, DM_LOAD_AFFINE_RELU, cx.coefd), Op(g), G(up1.weight), B, c1, NIN, 2 * H2, 2 * W2, 4,
              pending=pending)
    dy, st = ops.conv4x4s2(Op(g), weight_view(_w(up1.weight), NIN * 16, 16, 4, 1), B, NIN, c1, 4 * H2, 4 * W2,
                           want_stats=True, mask=Op(cx.d1, DM_LOAD_AFFINE, cx.coefd), stat_q=cx.d1)
    cdb = _bn_backward(st, B * 4 * H2 * W2, bn, cx.savedd, G)
    # da = BatchNorm backward of dy.  The weight-gradient kernels take that transform on their S operand; as the T operand only
    # where the one-pass kernel prefetches both tensors (the example widths) -- elsewhere da is materialised here
    da = Op(dy, DM_LOAD_AFFINE2, cdb, p1=cx.d1)
    if not ops.wgrad_t_affine2_supported(nh, c1, H2, W2, 4):
        da = Op(ops.apply(da, B, c1, 2 * H2, 2 * W2))
    ops.wgrad(Op(cx.r), da, G(up0.weight), B, nh, c1, H2, W2, 4, pending=pending)
    _fed_bias(G(up0.bias), bn, cdb, G, zero_fed_biases)
    g_r = None
    if want_gr:
        g_r, _ = ops.conv4x4s2(da, weight_view(_w(up0.weight), c1 * 16, 16, 4, 1), B, c1, nh, 2 * H2, 2 * W2)
    if own:
        ops.reduce_slabs_multi(pending)
    return g_r
