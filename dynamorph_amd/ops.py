"""Tensor-level wrappers over the C ABI: they take CUDA (ROCm) torch tensors, pass raw
device pointers and the current HIP stream, and allocate outputs with PyTorch.

Nothing here computes on the host and nothing falls back to ATen: a tensor that is not a
contiguous fp32 device tensor is an error.
"""
import ctypes as C
import functools

import torch

from . import _lib as L
from ._lib import DM_LOAD_AFFINE, DM_LOAD_AFFINE2, DM_LOAD_AFFINE_RELU, DM_LOAD_IDENT, DM_LOAD_RELU  # noqa: F401


def _stream():
    """The stream of the device the call's tensors live on (recorded by _ptr), not of whatever device is current; the
    launch itself runs under a device guard (_lib._guarded)."""
    dev = L.call_device.index
    return torch.cuda.current_stream(dev).cuda_stream


def _op(fn):
    """Every tensor-level wrapper below is one "call": the device recorded by _ptr is cleared when it starts and when it
    ends (also on an exception), so a half-assembled launch can never leak its device into the next call, and every pointer
    marshalled for the launch -- before or after host-only queries -- is checked against the same device."""
    @functools.wraps(fn)
    def call(*args, **kw):
        L.call_device.index = None
        try:
            return fn(*args, **kw)
        finally:
            L.call_device.index = None
    return call


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("dynamorph_amd: tensor is not on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise ValueError(f"dynamorph_amd: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("dynamorph_amd: tensor must be contiguous")
    _note_device(t.device.index)
    return t.data_ptr()


def _note_device(dev):
    """First operand of a call: its device becomes the call's; every later one must live there too."""
    if L.call_device.index is None:
        L.call_device.index = dev
    elif L.call_device.index != dev:
        raise ValueError(f"dynamorph_amd: operands of one call live on different devices "
                         f"(cuda:{L.call_device.index} and cuda:{dev})")


class Op:
    """A tensor together with its on-load transform (dm_operand)."""
    __slots__ = ("p0", "p1", "coef", "mode", "ones", "bstride", "_keep")

    def __init__(self, p0, mode=DM_LOAD_IDENT, coef=None, p1=None, ones=False, per_sample=False):
        self.p0, self.p1, self.coef, self.mode, self.ones = p0, p1, coef, mode, ones
        # per-sample coefficients are (B, C, 4); eval-mode BatchNorm hands the shared (C, 4) table to the per-sample path too
        self.bstride = coef.shape[-2] * 4 if (per_sample and coef is not None and coef.dim() == 3) else 0

    def struct(self):
        return L.Operand(_ptr(self.p0), _ptr(self.p1), _ptr(self.coef), self.bstride, self.mode, 1 if self.ones else 0)


def _null_operand():
    return L.Operand(None, None, None, 0, DM_LOAD_IDENT, 0)


class WView:
    """dm_weight_view plus a reference to the tensor it points into (a bare struct would dangle as soon
    as a temporary such as `w.to(device)` is collected and the caching allocator reuses its block)."""
    __slots__ = ("struct", "_keep", "_scratch")

    def __init__(self, w, sn, sc, sky, skx, off=0):
        if not (w.is_cuda and w.dtype == torch.float32):
            raise ValueError("dynamorph_amd: weights must be fp32 device tensors (the HIP path has no CPU fallback)")
        if not w.is_contiguous():
            # the strides below address a contiguous (n, c, ky, kx) tensor: a channels_last / transposed / sliced parameter
            # would be read with the wrong layout
            raise ValueError("dynamorph_amd: weights must be contiguous")
        self._keep = w
        self._scratch = None
        self.struct = L.WeightView(w.data_ptr(), off, sn, sc, sky, skx, None, 0)

    def ref(self):
        """The struct for a launch; records / checks the weights' device like every other operand of the call."""
        _note_device(self._keep.device.index)
        return C.byref(self.struct)

    def with_scratch(self, nfloats):
        """Attach `nfloats` of device scratch (dm_weight_view.scratch; what dm_conv*_scratch_floats asked for)."""
        if nfloats > 0 and (self._scratch is None or self._scratch.numel() < nfloats):
            self._scratch = torch.empty(nfloats, device=self._keep.device, dtype=torch.float32)
            self.struct.scratch = self._scratch.data_ptr()
            self.struct.scratch_floats = nfloats
        return self


def weight_view(w, sn, sc, sky, skx, off=0):
    return WView(w, sn, sc, sky, skx, off)


def epilogue(bias=None, relu=False, mask=None, resid=None, stat_q=None, stats=None, per_tile=False, bias_border=None):
    m = mask.struct() if mask is not None else _null_operand()
    return L.Epilogue(_ptr(bias), _ptr(bias_border), 1 if relu else 0, 1 if per_tile else 0, m, _ptr(resid),
                      _ptr(stat_q), _ptr(stats, torch.float64))


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


# ----------------------------------------------------------------------------- VQ
@_op
def vq_forward(z, codebook, want_idx=True, want_out=True, variant=L.DM_VQ_AUTO, want_rechecked=False, want_hist=True):
    """Returns (idx int64 (B,H,W), out (B,D,H,W), sse_slabs, hist).
    want_hist=False: no counter reduction is launched; the fourth return value is then the WORKSPACE holding the counter
    replicas, for vq_loss_finalize.
    variant: L.DM_VQ_AUTO (default) / DM_VQ_EXACT / DM_VQ_MFMA -- same results, different kernels (include/dynamorph_hip.h).
    want_rechecked: a fifth return value, the 1-element int32 device tensor counting the positions the MFMA kernel
    re-evaluated exactly (0 after the exact kernel)."""
    lib = L.load()
    if want_rechecked and not want_hist:
        raise ValueError("vq_forward: the re-check count is summed by the counter reduction (want_hist=True)")
    B, D, H, W = z.shape
    K = codebook.shape[0]
    nb = lib.dm_vq_num_blocks(B * H * W)
    idx = _new((B, H, W), z, torch.int64) if want_idx else None
    out = torch.empty_like(z) if want_out else None
    slabs = _new((nb,), z, torch.float64)
    hist = _new((K,), z, torch.int32) if want_hist else None         # cleared by dm_vq_forward itself
    wsb = lib.dm_vq_workspace_bytes(K, D)
    ws = _new((wsb // 4,), z)
    L.check(lib.dm_vq_forward_variant(_ptr(z), _ptr(codebook), _ptr(idx, torch.int64), _ptr(out),
                                      _ptr(slabs, torch.float64), _ptr(hist, torch.int32), B, D, K, H, W, _ptr(ws), wsb,
                                      variant, _stream()), "dm_vq_forward")
    if want_rechecked:
        return idx, out, slabs, hist, ws[:1].view(torch.int32)
    return idx, out, slabs, (hist if want_hist else ws)


def vq_forward_join_supported(D, K, H, W):
    return bool(L.load().dm_vq_forward_join_supported(D, K, H, W))


@_op
def vq_forward_join(rb, h_in, coef, codebook, want_idx=True, want_out=True):
    """The last residual join and the VectorQuantizer in one launch (include/dynamorph_hip.h, dm_vq_forward_join):
    z = fma(coef[:, 0], rb, coef[:, 2]) + h_in, quantised like vq_forward(z, ..., want_hist=False).
    Returns (idx, out, sse_slabs, workspace, z)."""
    lib = L.load()
    B, D, H, W = rb.shape
    K = codebook.shape[0]
    if tuple(h_in.shape) != tuple(rb.shape) or tuple(coef.shape) != (D, 4):
        raise ValueError("dm_vq_forward_join: rb / h_in (B, D, H, W) and a shared (D, 4) coefficient table")
    nb = lib.dm_vq_num_blocks(B * H * W)
    z = torch.empty_like(rb)
    idx = _new((B, H, W), rb, torch.int64) if want_idx else None
    out = torch.empty_like(rb) if want_out else None
    slabs = _new((nb,), rb, torch.float64)
    wsb = lib.dm_vq_workspace_bytes(K, D)
    ws = _new((wsb // 4,), rb)
    L.check(lib.dm_vq_forward_join(_ptr(rb), _ptr(h_in), _ptr(coef), _ptr(z), _ptr(codebook), _ptr(idx, torch.int64), _ptr(out),
                                   _ptr(slabs, torch.float64), None, B, D, K, H, W, _ptr(ws), wsb, _stream()),
            "dm_vq_forward_join")
    return idx, out, slabs, ws, z


@_op
def vq_forward_repeat(z, codebook, repeats, variant=L.DM_VQ_AUTO, bufs=None, want_hist=True):
    """Measurement helper (include/dynamorph_hip.h, dm_vq_forward_repeat): one preparation, `repeats` launches of the
    distance / argmin kernel, one counter reduction, on preallocated buffers `bufs` (from a first call) so that nothing is
    allocated inside a timed region.  want_hist=False: hist = NULL, i.e. no counter reduction -- how the training step and
    the inference path call dm_vq_forward (their counters are read from the workspace by the step's one scalar launch, or
    not at all).  Returns bufs."""
    lib = L.load()
    B, D, H, W = z.shape
    K = codebook.shape[0]
    if bufs is None:
        wsb = lib.dm_vq_workspace_bytes(K, D)
        bufs = (_new((B, H, W), z, torch.int64), torch.empty_like(z), _new((lib.dm_vq_num_blocks(B * H * W),), z, torch.float64),
                _new((K,), z, torch.int32), _new((wsb // 4,), z), wsb)
    idx, out, slabs, hist, ws, wsb = bufs
    L.check(lib.dm_vq_forward_repeat(_ptr(z), _ptr(codebook), _ptr(idx, torch.int64), _ptr(out), _ptr(slabs, torch.float64),
                                     _ptr(hist if want_hist else None, torch.int32), B, D, K, H, W, _ptr(ws), wsb, variant,
                                     repeats, _stream()),
            "dm_vq_forward_repeat")
    return bufs


@_op
def vq_finalize(slabs, hist, positions, D, commitment_cost):
    lib = L.load()
    scalars = _new((3,), slabs)
    L.check(lib.dm_vq_finalize(_ptr(slabs, torch.float64), slabs.numel(), _ptr(hist, torch.int32), hist.numel(),
                               positions, D, commitment_cost, _ptr(scalars), _stream()), "dm_vq_finalize")
    return scalars


@_op
def vq_loss_finalize(sse_slabs, ws, K, D, positions, commitment_cost, loss_slabs, count, weight_recon, weight_commitment):
    """(recon, commitment, total, perplexity) in one launch: vq_finalize + loss_finalize on the counter replicas that
    vq_forward(..., want_hist=False) left in its workspace `ws`."""
    lib = L.load()
    out = _new((4,), sse_slabs)
    L.check(lib.dm_vq_loss_finalize(_ptr(sse_slabs, torch.float64), sse_slabs.numel(), _ptr(ws), K, D, positions,
                                    commitment_cost, _ptr(loss_slabs, torch.float64), loss_slabs.numel(), count,
                                    weight_recon, weight_commitment, _ptr(out), _stream()), "dm_vq_loss_finalize")
    return out


@_op
def vq_loss_finalize_tm(sse_slabs, ws, K, D, positions, commitment_cost, loss_slabs, count, weight_recon, weight_commitment,
                        tm_slabs, weight_matching):
    """vq_loss_finalize with the pairwise term: (recon, commitment, total + weight_matching * tm, perplexity, tm) in one
    launch from the partial losses time_matching_forward(..., want_slabs=True) left."""
    lib = L.load()
    out = _new((5,), sse_slabs)
    L.check(lib.dm_vq_loss_finalize_tm(_ptr(sse_slabs, torch.float64), sse_slabs.numel(), _ptr(ws), K, D, positions,
                                       commitment_cost, _ptr(loss_slabs, torch.float64), loss_slabs.numel(), count,
                                       weight_recon, weight_commitment, _ptr(tm_slabs, torch.float64), tm_slabs.shape[0],
                                       weight_matching, _ptr(out), _stream()), "dm_vq_loss_finalize_tm")
    return out


@_op
def vq_decode(idx, codebook):
    lib = L.load()
    B, H, W = idx.shape
    K, D = codebook.shape
    q = _new((B, D, H, W), codebook)
    L.check(lib.dm_vq_decode(_ptr(idx, torch.int64), _ptr(codebook), _ptr(q), B, D, K, H, W, _stream()), "dm_vq_decode")
    return q


@_op
def vq_backward(z, codebook, idx, g_out, g_loss, commitment_cost, dw=None, want_dz=True):
    """g_loss: 1-element device tensor (or None = 1).  dw given: accumulated into with float atomics (zero it first);
    dw None: a fresh tensor from the slab form (see vq_backward_slabs for when that form is ordered)."""
    lib = L.load()
    B, D, H, W = z.shape
    K = codebook.shape[0]
    if dw is None:
        dz, slabs = vq_backward_slabs(z, codebook, idx, g_out, g_loss, commitment_cost, want_dz=want_dz)
        return dz, reduce_slabs(slabs, torch.empty_like(codebook))
    dz = torch.empty_like(z) if want_dz else None
    L.check(lib.dm_vq_backward(_ptr(z), _ptr(codebook), _ptr(idx, torch.int64), _ptr(g_out), _ptr(g_loss),
                               commitment_cost, _ptr(dz), _ptr(dw), B, D, K, H, W, _stream()), "dm_vq_backward")
    return dz, dw


@_op
def vq_backward_slabs(z, codebook, idx, g_out, g_loss, commitment_cost, want_dz=True):
    """Like vq_backward, but the codebook gradient comes back as per-workgroup slabs (nslabs, K*D) for
    reduce_slabs / reduce_slabs_multi: no global float atomics, nothing to zero.  Codebooks of at most 64 codes with
    embedding_dim 16/32/64 and H*W % 64 == 0 (every reference configuration) accumulate as a one-hot product on the
    matrix cores in a fixed order: bit-reproducible.  Larger codebooks add inside a workgroup with LDS float atomics in
    arrival order, so the low bits of THEIR gradient may differ from run to run."""
    lib = L.load()
    B, D, H, W = z.shape
    K = codebook.shape[0]
    dz = torch.empty_like(z) if want_dz else None
    slabs = _new((lib.dm_vq_backward_num_slabs(B * H * W, K, D), K * D), z)
    L.check(lib.dm_vq_backward_slabs(_ptr(z), _ptr(codebook), _ptr(idx, torch.int64), _ptr(g_out), _ptr(g_loss),
                                     commitment_cost, _ptr(dz), _ptr(slabs), B, D, K, H, W, _stream()),
            "dm_vq_backward_slabs")
    return dz, slabs


# ---------------------------------------------------------------------- convolutions
@_op
def conv4x4s2(inp, wv, B, CIN, NOUT, H, W, ep=None, out=None, want_stats=False, like=None, **epkw):
    lib = L.load()
    like = like if like is not None else inp.p0
    if out is None:
        out = _new((B, NOUT, H // 2, W // 2), like)
    stats = None
    if want_stats:
        nb = lib.dm_conv4x4s2_num_blocks(B, CIN, NOUT, H, W, 1 if epkw.get("per_tile") else 0)
        if nb <= 0:
            raise ValueError(f"dm_conv4x4s2: shape {(B, CIN, NOUT, H, W)} not tileable")
        stats = _new((nb, NOUT, 2), like, torch.float64)
    e = epilogue(stats=stats, **epkw)
    o = inp.struct()
    fallback = inp.mode == L.DM_LOAD_AFFINE2 or (epkw.get("bias_border") is not None and
                                                  any(epkw.get(k) is not None for k in ("mask", "resid", "stat_q")))
    wv.with_scratch(lib.dm_conv4x4s2_scratch_floats(CIN, NOUT, H, W, 1 if fallback else 0))
    L.check(lib.dm_conv4x4s2(C.byref(o), wv.ref(), _ptr(out), C.byref(e), B, CIN, NOUT, H, W, _stream()),
            "dm_conv4x4s2")
    return out, stats


@_op
def conv3x3(inp, wv, B, CIN, NOUT, H, W, taps=9, pixel_shuffle=False, out=None, want_stats=False, like=None, **epkw):
    lib = L.load()
    like = like if like is not None else inp.p0
    co = NOUT // 4 if pixel_shuffle else NOUT
    if out is None:
        out = _new((B, co, 2 * H, 2 * W) if pixel_shuffle else (B, co, H, W), like)
    stats = None
    if want_stats:
        nb = lib.dm_conv3x3_num_blocks(B, CIN, NOUT, H, W, taps, 1 if pixel_shuffle else 0, 1 if epkw.get("per_tile") else 0)
        if nb <= 0:
            raise ValueError(f"dm_conv3x3: shape {(B, CIN, NOUT, H, W)} not tileable")
        stats = _new((nb, co, 2), like, torch.float64)
    e = epilogue(stats=stats, **epkw)
    o = inp.struct()
    wv.with_scratch(lib.dm_conv3x3_scratch_floats(CIN, NOUT, H, W, taps, 1 if pixel_shuffle else 0,
                                                  1 if epkw.get("per_tile") else 0))
    L.check(lib.dm_conv3x3(C.byref(o), wv.ref(), _ptr(out), C.byref(e), B, CIN, NOUT, H, W, taps,
                           1 if pixel_shuffle else 0, _stream()), "dm_conv3x3")
    return out, stats


@_op
def wgrad(S, T, dst, B, CS, CT, Hs, Ws, k, pending=None):
    """dst (CS*CT*k*k floats, any shape) <- sum over batch/positions; deterministic slab reduction.
    pending: a list -> the slabs are left unreduced and (slabs, dst) is appended for reduce_slabs_multi."""
    lib = L.load()
    nb = lib.dm_wgrad_num_blocks(B, CS, CT, Hs, Ws, k)
    if nb <= 0:
        raise ValueError(f"dm_wgrad: shape {(B, CS, CT, Hs, Ws, k)} not tileable")
    slabs = _new((nb, CS * CT * k * k), S.p0)
    s, t = S.struct(), T.struct()
    L.check(lib.dm_wgrad(C.byref(s), C.byref(t), _ptr(slabs), None if pending is not None else _ptr(dst), B, CS, CT, Hs,
                         Ws, k, _stream()), "dm_wgrad")
    if pending is not None:
        pending.append((slabs, dst))
    return dst


def wgrad_t_affine2_supported(CS, CT, Hs, Ws, k):
    """True where dm_wgrad takes T as an AFFINE2 operand (a BatchNorm backward folded into the load of the output gradient)."""
    return bool(L.load().dm_wgrad_t_affine2_supported(CS, CT, Hs, Ws, k))


def backward_precision(mode=None):
    """Arithmetic of the backward matrix products: "f32", the exact fp32 chain -- the only one built (the split-bf16 opt-in
    of earlier rounds is retired: include/dynamorph_hip.h, dm_backward_precision; asking for it raises).  mode None: query."""
    names = ("f32", "split-bf16")
    if mode is not None and mode not in names:
        raise ValueError(f"backward_precision: {mode!r} (one of {names})")
    prev = L.load().dm_backward_precision(-1 if mode is None else names.index(mode))
    L.check(min(prev, 0), "dm_backward_precision")
    return names[prev]


def conv_bwd_s2_fused_supported(CD, CX, H, W):
    return bool(L.load().dm_conv_bwd_s2_fused_supported(CD, CX, H, W))


@_op
def conv_bwd_s2_fused(dy, tin, wv, dst, B, CD, CX, H, W, mask, stat_q=None, pending=None, want_stats=True):
    """Backward of a Conv2d(CX -> CD, 4, 2, 1) in one kernel (include/dynamorph_hip.h, dm_conv_bwd_s2_fused): data
    gradient (B, CX, 2H, 2W), masked by `mask` (the layer input, AFFINE) with its (sum, sum * input) statistics slabs, and
    the weight-gradient slabs, queued for reduce_slabs_multi into `dst` (pending) or reduced at once (pending None).
    dy: the output gradient as an AFFINE2 operand (BatchNorm backward folded in) on the (H, W) grid; tin: the layer input as
    the forward read it (AFFINE_RELU).  Returns (dx, stats)."""
    lib = L.load()
    nb = lib.dm_conv_bwd_s2_fused_num_blocks(B, CD, CX, H, W)
    if nb <= 0:
        raise ValueError(f"dm_conv_bwd_s2_fused: shape {(B, CD, CX, H, W)} not built")
    dx = _new((B, CX, 2 * H, 2 * W), dy.p0)
    stats = _new((nb, CX, 2), dy.p0, torch.float64) if want_stats else None
    slabs = _new((nb, CD * CX * 16), dy.p0)
    e = epilogue(mask=mask, stat_q=stat_q, stats=stats)
    d, t = dy.struct(), tin.struct()
    L.check(lib.dm_conv_bwd_s2_fused(C.byref(d), C.byref(t), wv.ref(), _ptr(dx), C.byref(e), _ptr(slabs), B, CD, CX, H, W,
                                     _stream()), "dm_conv_bwd_s2_fused")
    if pending is not None:
        pending.append((slabs, dst))
    else:
        reduce_slabs(slabs, dst)
    return dx, stats


def conv1x1_bwd_fused_supported(CD, CX, H, W):
    return bool(L.load().dm_conv1x1_bwd_fused_supported(CD, CX, H, W))


@_op
def conv1x1_bwd_fused(dy, x, xcoef, w, dst, B, CD, CX, H, W, pending=None):
    """Data AND weight gradient of a 1x1 convolution (CX -> CD channels) that feeds a train-mode BatchNorm, one launch
    (include/dynamorph_hip.h, dm_conv1x1_bwd_fused).  dy: Op of the output gradient (AFFINE2 = BatchNorm backward folded in);
    x: the layer input raw, xcoef (CX, 4) its BatchNorm + ReLU coefficients; w (CD, CX, 1, 1); dst: the weight gradient.
    Returns (dx, stats (nslabs, CX, 2)); the weight slabs are reduced into dst here, or queued on `pending`."""
    lib = L.load()
    nb = lib.dm_conv1x1_bwd_fused_num_blocks(B, CD, CX, H, W)
    if nb <= 0:
        raise ValueError(f"dm_conv1x1_bwd_fused: shape {CX} -> {CD} channels on {H}x{W} not built")
    dx = _new((B, CX, H, W), x)
    stats = _new((nb, CX, 2), x, torch.float64)
    slabs = _new((nb, CD * CX), x)
    d = dy.struct()
    L.check(lib.dm_conv1x1_bwd_fused(C.byref(d), _ptr(x), _ptr(xcoef), _ptr(w), _ptr(dx), _ptr(stats, torch.float64),
                                     _ptr(slabs), B, CD, CX, H, W, _stream()), "dm_conv1x1_bwd_fused")
    if pending is not None:
        pending.append((slabs, dst))
    else:
        reduce_slabs(slabs, dst)
    return dx, stats


def conv3x3_bwd_fused_supported(CD, CX, H, W):
    return bool(L.load().dm_conv3x3_bwd_fused_supported(CD, CX, H, W))


@_op
def conv3x3_bwd_fused(dy, x, xcoef, w, dst, B, CD, resid=None, q=None, want_stats=True, pending=None):
    """Data AND weight gradient of a 3x3 convolution (16 -> CD channels, 16 x 16 latents) that feeds a train-mode BatchNorm,
    one launch (include/dynamorph_hip.h, dm_conv3x3_bwd_fused).  dy: Op of the output gradient; x: the layer input raw,
    xcoef (16, 4) its BatchNorm + ReLU coefficients (None: plain ReLU); resid: added to dx; q: second factor of the statistics.
    Returns (dx, stats (nslabs, 16, 2) or None); the weight slabs are reduced into dst here, or queued on `pending`."""
    lib = L.load()
    CX, H, W = x.shape[1], x.shape[2], x.shape[3]
    nb = lib.dm_conv3x3_bwd_fused_num_blocks(B, CD, CX, H, W)
    if nb <= 0:
        raise ValueError(f"dm_conv3x3_bwd_fused: shape {CX} -> {CD} channels on {H}x{W} not built")
    dx = torch.empty_like(x)
    stats = _new((nb, CX, 2), x, torch.float64) if want_stats else None
    slabs = _new((nb, CD * CX * 9), x)
    d = dy.struct()
    L.check(lib.dm_conv3x3_bwd_fused(C.byref(d), _ptr(x), _ptr(xcoef), _ptr(w), _ptr(resid), _ptr(q), _ptr(dx),
                                     _ptr(stats, torch.float64), _ptr(slabs), B, CD, CX, H, W, _stream()),
            "dm_conv3x3_bwd_fused")
    if pending is not None:
        pending.append((slabs, dst))
    else:
        reduce_slabs(slabs, dst)
    return dx, stats


def conv4x4s2_bwd_fused_supported(CD, CX, H, W):
    return bool(L.load().dm_conv4x4s2_bwd_fused_supported(CD, CX, H, W))


@_op
def conv4x4s2_bwd_fused(dy, x, xcoef, w, dst, B, pending=None):
    """Data AND weight gradient of Conv2d(16 -> 16, 4, 2, 1) (enc.7) that feeds a train-mode BatchNorm, one launch
    (include/dynamorph_hip.h, dm_conv4x4s2_bwd_fused).  dy: Op of the output gradient (B, 16, 16, 16); x (B, 16, 32, 32): the
    layer input raw, xcoef (16, 4) its BatchNorm + ReLU coefficients.  Returns (dx, stats (nslabs, 16, 2))."""
    lib = L.load()
    CX, CD = x.shape[1], w.shape[0]
    H, W = x.shape[2] // 2, x.shape[3] // 2
    nb = lib.dm_conv4x4s2_bwd_fused_num_blocks(B, CD, CX, H, W)
    if nb <= 0:
        raise ValueError(f"dm_conv4x4s2_bwd_fused: shape {CX} -> {CD} channels, {H}x{W} output grid not built")
    dx = torch.empty_like(x)
    stats = _new((nb, CX, 2), x, torch.float64)
    slabs = _new((nb, CD * CX * 16), x)
    d = dy.struct()
    L.check(lib.dm_conv4x4s2_bwd_fused(C.byref(d), _ptr(x), _ptr(xcoef), _ptr(w), _ptr(dx), _ptr(stats, torch.float64),
                                       _ptr(slabs), B, CD, CX, H, W, _stream()), "dm_conv4x4s2_bwd_fused")
    if pending is not None:
        pending.append((slabs, dst))
    else:
        reduce_slabs(slabs, dst)
    return dx, stats


def convT_bwd_fused_supported(CI, CO, H, W):
    return bool(L.load().dm_convt_bwd_fused_supported(CI, CO, H, W))


@_op
def convT_bwd_fused(S, G, w, dst, mask_relu=False, want_stats=False, pending=None):
    """Input AND weight gradient of a thin ConvTranspose2d(CI -> CO, 4, 2, 1), one launch (include/dynamorph_hip.h,
    dm_convt_bwd_fused).  S (B, CI, H, W): the layer input; G (B, CO, 2H, 2W): the output gradient; w (CI, CO, 4, 4);
    dst: the weight gradient.  Returns (gin, stats (nslabs, CI, 2) or None); the weight slabs are reduced into dst here, or
    queued on `pending`."""
    lib = L.load()
    B, CI, H, W = S.shape
    CO = w.shape[1]
    if tuple(G.shape) != (B, CO, 2 * H, 2 * W) or tuple(w.shape) != (CI, CO, 4, 4):
        raise ValueError("dm_convt_bwd_fused: shapes do not match")
    nb = lib.dm_convt_bwd_fused_num_blocks(B, CI, CO, H, W)
    if nb <= 0:
        raise ValueError(f"dm_convt_bwd_fused: ConvTranspose2d({CI} -> {CO}) on {H}x{W} not built")
    gin = torch.empty_like(S)
    stats = _new((nb, CI, 2), S, torch.float64) if want_stats else None
    slabs = _new((nb, CI * CO * 16), S)
    L.check(lib.dm_convt_bwd_fused(_ptr(S), _ptr(G), _ptr(w), _ptr(gin), _ptr(stats, torch.float64), _ptr(slabs),
                                   1 if mask_relu else 0, B, CI, CO, H, W, _stream()), "dm_convt_bwd_fused")
    if pending is not None:
        pending.append((slabs, dst))
    else:
        reduce_slabs(slabs, dst)
    return gin, stats


def pend_stats(pending, stats, dsts):
    """Queue the column sums of a statistics slab tensor (nslabs, N, 2) for reduce_slabs_multi: consecutive runs of
    columns go to the tensors `dsts` (what sum_slabs / sum_slabs_scatter would do in a launch of their own)."""
    col = 0
    for d in dsts:
        pending.append((stats, d, col))
        col += d.numel()
    if col != stats.shape[1]:
        raise ValueError("pend_stats: destinations must cover the slab's columns")


@_op
def reduce_slabs_multi(pending):
    """One launch for every (slabs, dst) pair collected by wgrad(..., pending=...) and every (stats, dst, column) run
    queued by pend_stats."""
    lib = L.load()
    while pending:
        chunk, pending[:] = pending[:32], pending[32:]
        segs = (L.ReduceSeg * len(chunk))()
        for i, item in enumerate(chunk):
            if len(item) == 3:      # (stats (nslabs, N, 2) float64, dst, first column): a run of columns of a statistics slab
                stats, dst, col = item
                if stats.dtype != torch.float64 or stats.dim() != 3 or col + dst.numel() > stats.shape[1]:
                    raise ValueError("reduce_slabs_multi: bad statistics segment")
                _ptr(stats, torch.float64)
                segs[i] = L.ReduceSeg(stats.data_ptr() + 16 * col, _ptr(dst), stats.shape[0], dst.numel(), stats.shape[1], 1)
            else:
                slabs, dst = item
                segs[i] = L.ReduceSeg(_ptr(slabs), _ptr(dst), slabs.shape[0], slabs.shape[1], 0, 0)
        L.check(lib.dm_reduce_slabs_multi(segs, len(chunk), _stream()), "dm_reduce_slabs_multi")


@_op
def sum_slabs_scatter(stats, dsts, scale=1.0):
    """stats (nslabs, N, 2) float64 -> consecutive runs of the N sums written straight into the tensors `dsts`."""
    lib = L.load()
    N = stats.shape[1]
    sc = L.Scatter()
    sc.nseg = len(dsts)
    end = 0
    for k, d in enumerate(dsts):
        end += d.numel()
        sc.end[k] = end
        sc.dst[k] = _ptr(d)
    if end != N:
        raise ValueError(f"dm_sum_slabs_scatter: destinations cover {end} of {N} entries")
    L.check(lib.dm_sum_slabs_scatter(_ptr(stats, torch.float64), stats.shape[0], N, scale, C.byref(sc), _stream()),
            "dm_sum_slabs_scatter")


# ------------------------------------------------------------------------ BatchNorm
@_op
def bn_finalize(stats, count_per_group, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                per_sample=False, slabs_per_group=1, defer=None):
    """defer (per_sample only): a list -> the running statistics are NOT touched by this launch; their update is appended
    to the list for bn_running_replay, which brings every deferred layer up to date in one launch."""
    lib = L.load()
    nslabs, Cn = stats.shape[0], stats.shape[1]
    if per_sample and defer is not None:
        defer.append((stats, slabs_per_group, count_per_group, running_mean, running_var, nbt, momentum))
        running_mean = running_var = nbt = None
    if per_sample:
        Bn = nslabs // slabs_per_group
        coef = _new((Bn, Cn, 4), gamma)
        saved = _new((Bn, Cn, 2), gamma)
    else:
        coef = _new((Cn, 4), gamma)
        saved = _new((Cn, 2), gamma)
    L.check(lib.dm_bn_finalize(_ptr(stats, torch.float64), nslabs, slabs_per_group, Cn, count_per_group, _ptr(gamma),
                               _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(nbt, torch.int64), momentum,
                               eps, _ptr(coef), _ptr(saved), 1 if per_sample else 0, _stream()), "dm_bn_finalize")
    return coef, saved


@_op
def bn_running_replay(deferred):
    """One launch for the running statistics of every layer bn_finalize(..., defer=deferred) skipped."""
    lib = L.load()
    while deferred:
        chunk, deferred[:] = deferred[:16], deferred[16:]
        segs = (L.ReplaySeg * len(chunk))()
        for i, (stats, spg, cnt, rm, rv, nbt, mom) in enumerate(chunk):
            segs[i] = L.ReplaySeg(_ptr(stats, torch.float64), stats.shape[0], spg, stats.shape[1], cnt, _ptr(rm), _ptr(rv),
                                  _ptr(nbt, torch.int64), mom)
        L.check(lib.dm_bn_running_replay(segs, len(chunk), _stream()), "dm_bn_running_replay")


def latent_tail_supported(c, cr, h, w, nres):
    return bool(L.load().dm_latent_tail_supported(c, cr, h, w, nres))


@_op
def latent_tail_forward(a3, coef3, w10, b10, gamma4, beta4, eps4, res, enc7=None):
    """enc.10 .. enc.12 of every patch with that patch's own BatchNorm statistics, one launch (csrc/latent_tail.hip).
    res: per residual layer (wa, ba, gamma_a, beta_a, eps_a, wb, bb, gamma_b, beta_b, eps_b).
    enc7 = (a2, coef2, w7, b7, gamma3, beta3, eps3): start at a2 (enc.4's raw output) instead -- enc.7 / enc.8 run in the
    kernel too, a3 / coef3 are None and a fourth return value holds the per-patch sums of enc.7's output.
    Returns (z, stats4, [(stats_a, stats_b), ...][, stats3]): the per-patch sums (B, C, 2) float64 for bn_running_replay."""
    lib = L.load()
    if enc7 is not None:
        a2 = enc7[0]
        B, Cn, H, W = a2.shape[0], a2.shape[1], a2.shape[2] // 2, a2.shape[3] // 2
        a3 = a2                                          # (allocation template only)
    else:
        B, Cn, H, W = a3.shape
    z = _new((B, Cn, H, W), a3)
    st4 = _new((B, Cn, 2), a3, torch.float64)
    args = L.LatentTailArgs()
    st3 = None
    if enc7 is not None:
        a2, coef2, w7, b7, g3, be3, eps3 = enc7
        st3 = _new((B, Cn, 2), a2, torch.float64)
        args.a2, args.coef2, args.w7, args.b7 = _ptr(a2), _ptr(coef2), _ptr(w7), _ptr(b7)
        args.gamma3, args.beta3, args.stats3, args.eps3 = _ptr(g3), _ptr(be3), _ptr(st3, torch.float64), eps3
    else:
        args.a3, args.coef3 = _ptr(a3), _ptr(coef3)
    args.w10, args.b10 = _ptr(w10), _ptr(b10)
    args.gamma4, args.beta4, args.stats4, args.z = _ptr(gamma4), _ptr(beta4), _ptr(st4, torch.float64), _ptr(z)
    args.eps4, args.B, args.C, args.H, args.W, args.nres = eps4, B, Cn, H, W, len(res)
    args.CR = res[0][0].shape[0] if res else 32
    sts = []
    for i, (wa, ba, ga, bea, epsa, wb, bb, gb, beb, epsb) in enumerate(res):
        sa = _new((B, wa.shape[0], 2), a3, torch.float64)
        sb = _new((B, wb.shape[0], 2), a3, torch.float64)
        r = args.res[i]
        r.wa, r.ba, r.gamma_a, r.beta_a, r.stats_a, r.eps_a = _ptr(wa), _ptr(ba), _ptr(ga), _ptr(bea), _ptr(sa, torch.float64), epsa
        r.wb, r.bb, r.gamma_b, r.beta_b, r.stats_b, r.eps_b = _ptr(wb), _ptr(bb), _ptr(gb), _ptr(beb), _ptr(sb, torch.float64), epsb
        sts.append((sa, sb))
    L.check(lib.dm_latent_tail_forward(C.byref(args), _stream()), "dm_latent_tail_forward")
    return (z, st4, sts, st3) if enc7 is not None else (z, st4, sts)


@_op
def bn_backward_finalize(stats, count, gamma, saved, dgamma, dbeta):
    lib = L.load()
    nslabs, Cn = stats.shape[0], stats.shape[1]
    coef_bwd = _new((Cn, 4), gamma)
    L.check(lib.dm_bn_backward_finalize(_ptr(stats, torch.float64), nslabs, Cn, count, _ptr(gamma), _ptr(saved),
                                        _ptr(dgamma), _ptr(dbeta), _ptr(coef_bwd), _stream()), "dm_bn_backward_finalize")
    return coef_bwd


@_op
def apply(inp, B, Cn, H, W, resid=None, out=None):
    lib = L.load()
    if out is None:
        out = _new((B, Cn, H, W), inp.p0)
    o = inp.struct()
    L.check(lib.dm_apply(C.byref(o), _ptr(resid), _ptr(out), B, Cn, H, W, _stream()), "dm_apply")
    return out


@_op
def channel_stats(p, q=None):
    lib = L.load()
    B, Cn, H, W = p.shape
    nb = lib.dm_channel_stats_num_blocks(B, Cn, H, W)
    stats = _new((nb, Cn, 2), p, torch.float64)
    L.check(lib.dm_channel_stats(_ptr(p), _ptr(q), _ptr(stats, torch.float64), B, Cn, H, W, _stream()), "dm_channel_stats")
    return stats


@_op
def sum_slabs(stats, dst, scale=1.0, n=None):
    """dst[i] = scale * sum_slabs stats[slab][i][0] for the first n = dst.numel() entries of the slab rows."""
    lib = L.load()
    N = stats.shape[1]
    L.check(lib.dm_sum_slabs(_ptr(stats, torch.float64), stats.shape[0], N, scale, _ptr(dst), _stream()), "dm_sum_slabs")
    return dst


# ------------------------------------------------------------------------------ head
@_op
def head_forward(d4, w6, b6, x, mask, channel_var):
    lib = L.load()
    B, C4, H, W = d4.shape
    NIN = w6.shape[0]
    dec = _new((B, NIN, H, W), d4)
    slabs = None
    if x is not None:
        slabs = _new((lib.dm_head_num_blocks(B, H, W),), d4, torch.float64)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_head_forward(_ptr(d4), _ptr(w6), _ptr(b6), _ptr(x), _ptr(mask), mc, _ptr(channel_var), _ptr(dec),
                                _ptr(slabs, torch.float64), B, C4, NIN, H, W, _stream()), "dm_head_forward")
    return dec, slabs


@_op
def head_backward(dec, x, mask, channel_var, d4, w6, gscale, gdec_ext=None):
    """Returns (g4, part) with part: (nblocks, NIN*C4 + NIN + C4, 2) float64 slabs."""
    lib = L.load()
    B, C4, H, W = d4.shape
    NIN = w6.shape[0]
    g4 = torch.empty_like(d4)
    part = _new((lib.dm_head_num_blocks(B, H, W), NIN * C4 + NIN + C4, 2), d4, torch.float64)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_head_backward(_ptr(dec), _ptr(x), _ptr(mask), mc, _ptr(channel_var), _ptr(d4), _ptr(w6),
                                 _ptr(gscale), _ptr(gdec_ext), _ptr(g4), _ptr(part, torch.float64), B, C4, NIN, H, W,
                                 _stream()), "dm_head_backward")
    return g4, part


def head_supported(c4, nin):
    return bool(L.load().dm_head_supported(c4, nin))


def dec_tail_supported(c2, nin, h2, w2):
    return bool(L.load().dm_dec_tail_supported(c2, nin, h2, w2))


@_op
def dec_tail_forward(d2, w4, b4, w6, b6, x, mask, channel_var):
    """Fused dec.4 + ReLU + dec.6 (+ masked reconstruction loss partials when x is given)."""
    lib = L.load()
    B, C2, H2, W2 = d2.shape
    NIN = w6.shape[0]
    dec = _new((B, NIN, 2 * H2, 2 * W2), d2)
    slabs = _new((lib.dm_dec_tail_num_blocks(B, H2, W2),), d2, torch.float64) if x is not None else None
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_dec_tail_forward(_ptr(d2), _ptr(w4), _ptr(b4), _ptr(w6), _ptr(b6), _ptr(x), _ptr(mask), mc,
                                    _ptr(channel_var), _ptr(dec), _ptr(slabs, torch.float64), B, C2, NIN, H2, W2,
                                    _stream()), "dm_dec_tail_forward")
    return dec, slabs


@_op
def dec_tail_backward(d2, w4, b4, w6, dec, x, mask, channel_var, gscale):
    """Returns (g2, part (nb, NIN*4+NIN+8, 2) float64, w_slabs (nb, 256) float32)."""
    lib = L.load()
    B, C2, H2, W2 = d2.shape
    NIN = w6.shape[0]
    nb = lib.dm_dec_tail_num_blocks(B, H2, W2)
    g2 = torch.empty_like(d2)
    part = _new((nb, NIN * C2 + NIN + 2 * C2, 2), d2, torch.float64)
    wsl = _new((nb, C2 * C2 * 16), d2)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_dec_tail_backward(_ptr(d2), _ptr(w4), _ptr(b4), _ptr(w6), _ptr(dec), _ptr(x), _ptr(mask), mc,
                                     _ptr(channel_var), _ptr(gscale), _ptr(g2), _ptr(part, torch.float64), _ptr(wsl),
                                     B, C2, NIN, H2, W2, _stream()), "dm_dec_tail_backward")
    return g2, part, wsl


@_op
def dec_tail_train(d2, w4, b4, w6, b6, x, mask, channel_var, gscale):
    """Forward loss + backward of the decoder tail in one kernel (no `decoded`).
    Returns (g2, part, w_slabs, loss_slabs)."""
    lib = L.load()
    B, C2, H2, W2 = d2.shape
    NIN = w6.shape[0]
    nb = lib.dm_dec_tail_num_blocks(B, H2, W2)
    g2 = torch.empty_like(d2)
    part = _new((nb, NIN * C2 + NIN + 2 * C2, 2), d2, torch.float64)
    wsl = _new((nb, C2 * C2 * 16), d2)
    loss = _new((nb,), d2, torch.float64)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_dec_tail_train(_ptr(d2), _ptr(w4), _ptr(b4), _ptr(w6), _ptr(b6), _ptr(x), _ptr(mask), mc,
                                  _ptr(channel_var), _ptr(gscale), _ptr(g2), _ptr(part, torch.float64), _ptr(wsl),
                                  _ptr(loss, torch.float64), B, C2, NIN, H2, W2, _stream()), "dm_dec_tail_train")
    return g2, part, wsl, loss


@_op
def reduce_slabs(slabs, dst):
    lib = L.load()
    L.check(lib.dm_reduce_slabs(_ptr(slabs), slabs.shape[0], slabs.shape[1], _ptr(dst), _stream()), "dm_reduce_slabs")
    return dst


@_op
def loss_finalize(loss_slabs, count, vq_scalars, weight_recon, weight_commitment):
    lib = L.load()
    out = _new((4,), vq_scalars)
    L.check(lib.dm_loss_finalize(_ptr(loss_slabs, torch.float64), loss_slabs.numel(), count, _ptr(vq_scalars),
                                 weight_recon, weight_commitment, _ptr(out), _stream()), "dm_loss_finalize")
    return out


# ------------------------------------------------------------------ plain reconstruction loss
@_op
def recon_loss(dec, x, mask, channel_var):
    """Partial sums (nblocks,) float64 of (dec*m - x*m)^2 / var for loss_finalize."""
    lib = L.load()
    B, NIN, H, W = dec.shape
    slabs = _new((lib.dm_recon_loss_num_blocks(B, NIN, H, W),), dec, torch.float64)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_recon_loss(_ptr(dec), _ptr(x), _ptr(mask), mc, _ptr(channel_var), _ptr(slabs, torch.float64),
                              B, NIN, H, W, _stream()), "dm_recon_loss")
    return slabs


@_op
def recon_loss_backward(dec, x, mask, channel_var, gscale):
    """(g_decoded, bias_slabs (nblocks, NIN, 2) float64)."""
    lib = L.load()
    B, NIN, H, W = dec.shape
    g = torch.empty_like(dec)
    part = _new((lib.dm_recon_loss_num_blocks(B, NIN, H, W), NIN, 2), dec, torch.float64)
    mc = mask.shape[1] if mask is not None else 0
    L.check(lib.dm_recon_loss_backward(_ptr(dec), _ptr(x), _ptr(mask), mc, _ptr(channel_var), _ptr(gscale), _ptr(g),
                                       _ptr(part, torch.float64), B, NIN, H, W, _stream()), "dm_recon_loss_backward")
    return g, part


# ------------------------------------------------------------------ time-matching loss
@_op
def pair_msd(z):
    """z (B, n) contiguous -> sim (B, B), sim[i][j] = mean((z[i] - z[j])**2)."""
    lib = L.load()
    B, n = z.shape
    sim = _new((B, B), z)
    L.check(lib.dm_pair_msd(_ptr(z), _ptr(sim), B, n, _stream()), "dm_pair_msd")
    return sim


@_op
def pair_msd_backward(z, g_sim):
    lib = L.load()
    B, n = z.shape
    dz = torch.empty_like(z)
    L.check(lib.dm_pair_msd_backward(_ptr(z), _ptr(g_sim), _ptr(dz), B, n, _stream()), "dm_pair_msd_backward")
    return dz


def time_matching_supported(B, n):
    return bool(L.load().dm_time_matching_supported(B, n))


@_op
def time_matching_forward(z, tm, mode, w_a=0.0, w_t=0.0, w_n=0.0, margin=0.0, want_slabs=False, allow_sparse=True):
    """The whole pairwise term on the MFMA (include/dynamorph_hip.h, dm_time_matching_forward).  z (B, n), tm (B, B) float32.
    Returns (loss: 1-element device tensor, S (2, B, B) = dloss/dsim + its transpose, far pairs / near pairs, for
    time_matching_backward).  want_slabs: the partial losses (nslabs, 1, 2) float64 instead of their sum (the training
    step's scalar launch adds them: vq_loss_finalize_tm)."""
    lib = L.load()
    B, n = z.shape
    wsf = lib.dm_time_matching_workspace_floats(B, n)
    ws = _new((wsf,), z)
    S = _new((2, B, B), z)
    nsl = lib.dm_time_matching_num_slabs(B)
    slabs = _new((nsl, 1, 2), z, torch.float64)
    # the state block the backward call reads (the pair count of mode 0's sparse form and the map of S's nonzero blocks:
    # include/dynamorph_hip.h) travels with S as an attribute; an S that lost it (a copy, a slice) takes the dense product,
    # which gives the same gradient
    if not allow_sparse:                                   # (tests / measurements: the stateless, dense form whatever tm holds)
        L.check(lib.dm_time_matching_forward(_ptr(z), _ptr(tm), B, n, mode, w_a, w_t, w_n, margin, _ptr(ws), wsf, _ptr(S),
                                             _ptr(slabs, torch.float64), _stream()), "dm_time_matching_forward")
        return (slabs, S) if want_slabs else (sum_slabs(slabs, _new((1,), z)), S)
    state = torch.empty(lib.dm_time_matching_state_ints(B), dtype=torch.int32, device=z.device)
    L.check(lib.dm_time_matching_forward_state(_ptr(z), _ptr(tm), B, n, mode, w_a, w_t, w_n, margin, _ptr(ws), wsf, _ptr(S),
                                               _ptr(slabs, torch.float64), _ptr(state, torch.int32), _stream()),
            "dm_time_matching_forward_state")
    S._dm_tm_state = state
    if want_slabs:
        return slabs, S
    loss = sum_slabs(slabs, _new((1,), z))
    return loss, S


@_op
def time_matching_backward(z, S, g_loss=None, scale=1.0, add=None):
    """dz = [add +] scale * g_loss[0] * d loss / d z  (g_loss: 1-element device tensor or None = 1; add: a gradient of the
    same latents, (B, n) or any shape with as many elements, summed in the kernel's store instead of an elementwise pass)."""
    lib = L.load()
    B, n = z.shape
    dz = torch.empty_like(z)
    if add is not None and add.numel() != z.numel():
        raise ValueError("dm_time_matching_backward_add: `add` must have the latents' size")
    state = getattr(S, "_dm_tm_state", None)
    if state is not None:
        L.check(lib.dm_time_matching_backward_state(_ptr(z), _ptr(S), _ptr(g_loss), scale, _ptr(add), _ptr(dz), B, n,
                                                    _ptr(state, torch.int32), _stream()), "dm_time_matching_backward_state")
        return dz
    if add is not None:
        L.check(lib.dm_time_matching_backward_add(_ptr(z), _ptr(S), _ptr(g_loss), scale, _ptr(add), _ptr(dz), B, n, _stream()),
                "dm_time_matching_backward_add")
        return dz
    L.check(lib.dm_time_matching_backward(_ptr(z), _ptr(S), _ptr(g_loss), scale, _ptr(dz), B, n, _stream()),
            "dm_time_matching_backward")
    return dz


# ------------------------------------------------------------- composition / optimizer
@_op
def e1_compose(w0, b0, w1):
    lib = L.load()
    C0, NIN = w0.shape[0], w0.shape[1]
    C1 = w1.shape[0]
    weff = _new((C1, NIN + 1, 4, 4), w1)
    L.check(lib.dm_e1_compose(_ptr(w0), _ptr(b0), _ptr(w1), _ptr(weff), NIN, C0, C1, _stream()), "dm_e1_compose")
    return weff


@_op
def e1_compose_border(w0, b0, w1, b1):
    """(weff, bias_border): bias_border (3, 3, C1) replaces the ones channel in the forward conv."""
    lib = L.load()
    C0, NIN = w0.shape[0], w0.shape[1]
    C1 = w1.shape[0]
    weff = _new((C1, NIN + 1, 4, 4), w1)
    table = _new((3, 3, C1), w1)
    L.check(lib.dm_e1_compose_border(_ptr(w0), _ptr(b0), _ptr(w1), _ptr(b1), _ptr(weff), _ptr(table), NIN, C0, C1,
                                     _stream()), "dm_e1_compose_border")
    return weff, table


@_op
def e1_chain(dweff, w0, b0, w1, dw0, db0, dw1):
    lib = L.load()
    C0, NIN = w0.shape[0], w0.shape[1]
    C1 = w1.shape[0]
    L.check(lib.dm_e1_chain(_ptr(dweff), _ptr(w0), _ptr(b0), _ptr(w1), _ptr(dw0), _ptr(db0), _ptr(dw1), NIN, C0, C1,
                            _stream()), "dm_e1_chain")


@_op
def adam(param, grad, m, v, lr, beta1, beta2, eps, step_dev):
    lib = L.load()
    L.check(lib.dm_adam(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), lr, beta1, beta2, eps,
                        _ptr(step_dev), _stream()), "dm_adam")


@_op
def adam_counted(param, grad, m, v, lr, beta1, beta2, eps, steps_done, steps_done_next, grad_scale=1.0):
    """grad_scale != 1: the step reads grad * grad_scale (data parallel: the all-reduced SUM x 1 / world, in the load)."""
    lib = L.load()
    if grad_scale == 1.0:
        L.check(lib.dm_adam_counted(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), lr, beta1, beta2, eps,
                                    _ptr(steps_done), _ptr(steps_done_next), _stream()), "dm_adam_counted")
    else:
        L.check(lib.dm_adam_counted_scaled(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), lr, beta1, beta2, eps,
                                           float(grad_scale), _ptr(steps_done), _ptr(steps_done_next), _stream()),
                "dm_adam_counted_scaled")


@_op
def zscore_channels(x, channel_mean, channel_std, out=None):
    """float32(zscore(x, channel_mean, channel_std)) of pipeline/train_utils.py:228-250 for an (N, C, H, W) float64 or
    float32 DEVICE tensor and GIVEN per-channel statistics (lists of Python floats, as the reference's config supplies them,
    or numpy scalars / arrays): bit-equal to the numpy expression followed by .astype(np.float32) -- the type numpy would
    compute in and the value of std + eps are worked out on the host with numpy itself."""
    import numpy as np
    lib = L.load()
    if x.dtype not in (torch.float64, torch.float32):
        raise ValueError("dm_zscore_channels: float64 or float32 input")
    x = x.contiguous()
    N, Cn, H, W = x.shape
    if len(channel_mean) != Cn or len(channel_std) != Cn:
        raise ValueError("dm_zscore_channels: one mean and one std per channel")
    np_dt = np.float64 if x.dtype == torch.float64 else np.float32
    mean, denom = np.empty(Cn), np.empty(Cn)
    dts, qts = set(), set()
    for c in range(Cn):
        diff = np.zeros(1, np_dt) - channel_mean[c]              # numpy's own promotion rules decide the types
        den = channel_std[c] + np.finfo(float).eps                # (the reference's expression: a float64 scalar)
        quot = diff / den
        dts.add(diff.dtype); qts.add(quot.dtype)
        mean[c] = float(np.asarray(channel_mean[c]).astype(diff.dtype))      # the scalars as the array operations see them
        denom[c] = float(np.asarray(den).astype(quot.dtype))
    if len(dts) != 1 or len(qts) != 1 or not dts <= {np.dtype(np.float32), np.dtype(np.float64)}:
        raise ValueError("dm_zscore_channels: statistics of mixed / unsupported types")
    diff_f64, quot_f64 = dts.pop() == np.float64, qts.pop() == np.float64
    md = torch.from_numpy(np.stack([mean, denom])).to(x.device)
    if out is None:
        out = torch.empty((N, Cn, H, W), dtype=torch.float32, device=x.device)
    L.check(lib.dm_zscore_channels(C.c_void_p(x.data_ptr()), 1 if x.dtype == torch.float64 else 0, 1 if diff_f64 else 0,
                                   1 if quot_f64 else 0, _ptr(out), C.c_void_p(md[0].data_ptr()), C.c_void_p(md[1].data_ptr()),
                                   N, Cn, H * W, _stream()), "dm_zscore_channels")
    return out


@_op
def zscore_patch(x):
    """x (N, C, H, W) float64 or float32 device tensor -> float32 z-scored patches (per patch and channel)."""
    lib = L.load()
    if x.dtype not in (torch.float64, torch.float32):
        raise ValueError("dm_zscore_patch: float64 or float32 input")
    x = x.contiguous()
    N, Cn, H, W = x.shape
    out = torch.empty((N, Cn, H, W), device=x.device, dtype=torch.float32)
    L.call_device.index = x.device.index
    L.check(lib.dm_zscore_patch(C.c_void_p(x.data_ptr()), 1 if x.dtype == torch.float64 else 0, _ptr(out), N * Cn, H * W,
                                _stream()), "dm_zscore_patch")
    return out


@_op
def augment(x, flip_code, rot_code):
    lib = L.load()
    B, Cn, H, W = x.shape
    if H != W:
        raise ValueError("dm_augment: square patches only")
    out = torch.empty_like(x)
    L.check(lib.dm_augment(_ptr(x), _ptr(out), _ptr(flip_code, torch.int32), _ptr(rot_code, torch.int32), B, Cn, H,
                           _stream()), "dm_augment")
    return out


# ----------------------------------------------------------------------------- feeding the step (csrc/feed.hip)
@_op
def gather_augment(src, ids, flip_code, rot_code, out, n=None):
    """out[:n] = rot90(flip(src[ids], flip), rot) (run_training.py:512 + :396-403) in one launch.  src (N, C, H, H) fp32
    in HBM, ids / codes int32 device vectors (None: the first n samples in order / no augmentation); `out` is written in
    place (the captured step's input buffer) and returned."""
    lib = L.load()
    N, Cn, H, W = src.shape
    if H != W:
        raise ValueError("dm_gather_augment: square patches only")
    n = int(n if n is not None else (ids.numel() if ids is not None else out.shape[0]))
    if out.shape[0] < n or tuple(out.shape[1:]) != (Cn, H, W):
        raise ValueError(f"dm_gather_augment: out {tuple(out.shape)} does not hold {n} x {(Cn, H, W)}")
    for t in (ids, flip_code, rot_code):
        if t is not None and t.numel() < n:
            raise ValueError("dm_gather_augment: fewer ids / codes than samples")
    L.check(lib.dm_gather_augment(_ptr(src), N, _ptr(ids, torch.int32), _ptr(flip_code, torch.int32),
                                  _ptr(rot_code, torch.int32), _ptr(out), n, Cn, H, _stream()), "dm_gather_augment")
    return out


@_op
def gather_rows(src, ids, out, n=None):
    """out[:n] = src[ids] for (N, ...) fp32 rows (the mask planes of a batch, run_training.py:371)."""
    lib = L.load()
    row = src[0].numel()
    n = int(n if n is not None else (ids.numel() if ids is not None else out.shape[0]))
    if out.shape[0] < n or out[0].numel() != row:
        raise ValueError("dm_gather_rows: out does not hold n rows of the source's size")
    if ids is not None and ids.numel() < n:
        raise ValueError("dm_gather_rows: fewer ids than rows")
    L.check(lib.dm_gather_rows(_ptr(src), src.shape[0], _ptr(ids, torch.int32), _ptr(out), n, row, _stream()),
            "dm_gather_rows")
    return out


@_op
def csr_block(indptr, indices, data, n, ids, pos, stamp, out):
    """out (B, B) = relation_mat[ids, :][:, ids].todense() (run_training.py:348-351) from CSR arrays in HBM."""
    lib = L.load()
    B = ids.numel()
    if tuple(out.shape) != (B, B) or pos.numel() < n or indptr.numel() != n + 1:
        raise ValueError("dm_csr_block: shapes do not match")
    if indices.numel() == 0:                      # an empty matrix: scipy hands over zero-length arrays (no pointer)
        out.zero_()
        return out
    L.check(lib.dm_csr_block(_ptr(indptr, torch.int64), _ptr(indices, torch.int32), _ptr(data), n, _ptr(ids, torch.int32),
                             B, _ptr(pos, torch.int64), int(stamp), _ptr(out), _stream()), "dm_csr_block")
    return out


def augment_codes(n):
    """(flip, rot) int32 numpy arrays for n samples, drawn from numpy's GLOBAL legacy generator exactly as the
    reference's per-sample loop draws them (run_training.py:399-402: choice([0,1,2]) then choice([0,1,2,3]), interleaved)
    -- a caller's np.random.seed reproduces the reference's augmentation -- without 2n Python-level draws: the
    generator's 32-bit words are drawn in one block, parsed by dm_augment_codes (host C), and the generator is left
    where the n interleaved calls would have left it."""
    import numpy as np
    lib = L.load()
    flip = np.empty(n, np.int32)
    rot = np.empty(n, np.int32)
    if n == 0:
        return flip, rot
    state = np.random.get_state()
    m = 3 * n + 64                                 # expected 2.33 words per sample; retried (doubling) if short
    while True:
        raw = np.random.randint(0, 2 ** 32, size=m, dtype=np.uint32)
        used = lib.dm_augment_codes(raw.ctypes.data, m, n, flip.ctypes.data, rot.ctypes.data)
        np.random.set_state(state)
        if used >= 0:
            break
        m *= 2
    np.random.randint(0, 2 ** 32, size=int(used), dtype=np.uint32)      # advance by exactly the words consumed
    return flip, rot
