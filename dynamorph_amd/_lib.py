"""ctypes binding of libdynamorph_hip.so (the C ABI declared in include/dynamorph_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails this module
raises, so a GPU box can never silently run something else than the HIP kernels.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DM_LIB_PATH: a diagnostic twin of the library (make -C csrc stamps) for experiments; the package default is the in-tree .so
LIB_PATH = os.environ.get("DM_LIB_PATH") or os.path.join(_HERE, "libdynamorph_hip.so")

DM_LOAD_IDENT, DM_LOAD_RELU, DM_LOAD_AFFINE, DM_LOAD_AFFINE_RELU, DM_LOAD_AFFINE2 = range(5)
DM_VQ_AUTO, DM_VQ_EXACT, DM_VQ_MFMA, DM_VQ_BF16 = range(4)

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Operand(C.Structure):
    _fields_ = [("p0", vp), ("p1", vp), ("coef", vp), ("coef_bstride", i64), ("mode", i32), ("ones_channel", i32)]


class WeightView(C.Structure):
    _fields_ = [("w", vp), ("off", i64), ("sn", i64), ("sc", i64), ("sky", i64), ("skx", i64), ("scratch", vp),
                ("scratch_floats", i64)]


class Epilogue(C.Structure):
    _fields_ = [("bias", vp), ("bias_border", vp), ("relu", i32), ("stats_per_tile", i32), ("mask", Operand),
                ("resid", vp), ("stat_q", vp), ("stats", vp)]


class Scatter(C.Structure):
    _fields_ = [("nseg", i32), ("end", i32 * 8), ("dst", vp * 8)]


class ReduceSeg(C.Structure):
    _fields_ = [("slabs", vp), ("dst", vp), ("nslabs", i32), ("E", i32), ("stride", i32), ("pairs_of_doubles", i32)]


class ReplaySeg(C.Structure):
    _fields_ = [("stats", vp), ("nslabs", i32), ("slabs_per_group", i32), ("C", i32), ("count_per_group", i64),
                ("running_mean", vp), ("running_var", vp), ("num_batches_tracked", vp), ("momentum", f32)]


class LatentTailRes(C.Structure):
    _fields_ = [("wa", vp), ("ba", vp), ("gamma_a", vp), ("beta_a", vp), ("stats_a", vp),
                ("wb", vp), ("bb", vp), ("gamma_b", vp), ("beta_b", vp), ("stats_b", vp), ("eps_a", f32), ("eps_b", f32)]


class LatentTailArgs(C.Structure):
    _fields_ = [("a3", vp), ("coef3", vp), ("w10", vp), ("b10", vp), ("gamma4", vp), ("beta4", vp), ("stats4", vp), ("z", vp),
                ("eps4", f32), ("B", i32), ("C", i32), ("CR", i32), ("H", i32), ("W", i32), ("nres", i32),
                ("res", LatentTailRes * 4),
                ("a2", vp), ("coef2", vp), ("w7", vp), ("b7", vp), ("gamma3", vp), ("beta3", vp), ("stats3", vp), ("eps3", f32)]


OP, WV, EP = C.POINTER(Operand), C.POINTER(WeightView), C.POINTER(Epilogue)

# name -> (restype, argtypes); every symbol include/dynamorph_hip.h declares
SIGNATURES = {
    "dm_last_error": (C.c_char_p, []),
    "dm_version": (C.c_int, []),
    "dm_vq_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dm_vq_num_blocks": (C.c_int, [i64]),
    "dm_vq_forward": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, C.c_size_t, vp]),
    "dm_vq_forward_variant": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, C.c_size_t, C.c_int, vp]),
    "dm_vq_forward_repeat": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, C.c_size_t, C.c_int, C.c_int, vp]),
    "dm_vq_forward_join_supported": (C.c_int, [C.c_int] * 4),
    "dm_vq_forward_join": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp, C.c_size_t, vp]),
    "dm_vq_decode": (C.c_int, [vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_vq_finalize": (C.c_int, [vp, C.c_int, vp, C.c_int, i64, C.c_int, f32, vp, vp]),
    "dm_vq_loss_finalize": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, i64, f32, vp, C.c_int, i64, f32, f32, vp, vp]),
    "dm_vq_loss_finalize_tm": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, i64, f32, vp, C.c_int, i64, f32, f32, vp, C.c_int, f32, vp, vp]),
    "dm_vq_backward": (C.c_int, [vp, vp, vp, vp, vp, f32, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_vq_backward_num_slabs": (C.c_int, [i64, C.c_int, C.c_int]),
    "dm_vq_backward_slabs": (C.c_int, [vp, vp, vp, vp, vp, f32, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_conv4x4s2": (C.c_int, [OP, WV, vp, EP] + [C.c_int] * 5 + [vp]),
    "dm_conv4x4s2_num_blocks": (C.c_int, [C.c_int] * 6),
    "dm_conv4x4s2_scratch_floats": (i64, [C.c_int] * 5),
    "dm_conv3x3": (C.c_int, [OP, WV, vp, EP] + [C.c_int] * 7 + [vp]),
    "dm_conv3x3_num_blocks": (C.c_int, [C.c_int] * 8),
    "dm_conv3x3_scratch_floats": (i64, [C.c_int] * 7),
    "dm_backward_precision": (C.c_int, [C.c_int]),
    "dm_conv_bwd_s2_fused_supported": (C.c_int, [C.c_int] * 4),
    "dm_conv_bwd_s2_fused_num_blocks": (C.c_int, [C.c_int] * 5),
    "dm_conv_bwd_s2_fused": (C.c_int, [OP, OP, WV, vp, EP, vp] + [C.c_int] * 5 + [vp]),
    "dm_conv1x1_bwd_fused_supported": (C.c_int, [C.c_int] * 4),
    "dm_conv1x1_bwd_fused_num_blocks": (C.c_int, [C.c_int] * 5),
    "dm_conv1x1_bwd_fused": (C.c_int, [OP, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_convt_bwd_fused_supported": (C.c_int, [C.c_int] * 4),
    "dm_convt_bwd_fused_num_blocks": (C.c_int, [C.c_int] * 5),
    "dm_convt_bwd_fused": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "dm_conv3x3_bwd_fused_supported": (C.c_int, [C.c_int] * 4),
    "dm_conv3x3_bwd_fused_num_blocks": (C.c_int, [C.c_int] * 5),
    "dm_conv3x3_bwd_fused": (C.c_int, [OP, vp, vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_conv4x4s2_bwd_fused_supported": (C.c_int, [C.c_int] * 4),
    "dm_conv4x4s2_bwd_fused_num_blocks": (C.c_int, [C.c_int] * 5),
    "dm_conv4x4s2_bwd_fused": (C.c_int, [OP, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_wgrad_num_blocks": (C.c_int, [C.c_int] * 6),
    "dm_wgrad_t_affine2_supported": (C.c_int, [C.c_int] * 5),
    "dm_wgrad": (C.c_int, [OP, OP, vp, vp] + [C.c_int] * 6 + [vp]),
    "dm_bn_finalize": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, vp, vp, f32, f32, vp, vp, C.c_int, vp]),
    "dm_bn_backward_finalize": (C.c_int, [vp, C.c_int, C.c_int, i64, vp, vp, vp, vp, vp, vp]),
    "dm_apply": (C.c_int, [OP, vp, vp] + [C.c_int] * 4 + [vp]),
    "dm_channel_stats_num_blocks": (C.c_int, [C.c_int] * 4),
    "dm_channel_stats": (C.c_int, [vp, vp, vp] + [C.c_int] * 4 + [vp]),
    "dm_sum_slabs": (C.c_int, [vp, C.c_int, C.c_int, f32, vp, vp]),
    "dm_sum_slabs_scatter": (C.c_int, [vp, C.c_int, C.c_int, f32, C.POINTER(Scatter), vp]),
    "dm_reduce_slabs_multi": (C.c_int, [C.POINTER(ReduceSeg), C.c_int, vp]),
    "dm_bn_running_replay": (C.c_int, [C.POINTER(ReplaySeg), C.c_int, vp]),
    "dm_latent_tail_supported": (C.c_int, [C.c_int] * 5),
    "dm_latent_tail_forward": (C.c_int, [C.POINTER(LatentTailArgs), vp]),
    "dm_head_supported": (C.c_int, [C.c_int] * 2),
    "dm_head_num_blocks": (C.c_int, [C.c_int] * 3),
    "dm_head_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_head_backward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_dec_tail_supported": (C.c_int, [C.c_int] * 4),
    "dm_dec_tail_num_blocks": (C.c_int, [C.c_int] * 3),
    "dm_dec_tail_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_dec_tail_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_dec_tail_train": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "dm_reduce_slabs": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "dm_loss_finalize": (C.c_int, [vp, C.c_int, i64, vp, f32, f32, vp, vp]),
    "dm_recon_loss_num_blocks": (C.c_int, [C.c_int] * 4),
    "dm_recon_loss": (C.c_int, [vp, vp, vp, C.c_int, vp, vp] + [C.c_int] * 4 + [vp]),
    "dm_recon_loss_backward": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, vp] + [C.c_int] * 4 + [vp]),
    "dm_pair_msd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "dm_pair_msd_backward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "dm_time_matching_supported": (C.c_int, [C.c_int, C.c_int]),
    "dm_time_matching_workspace_floats": (i64, [C.c_int, C.c_int]),
    "dm_time_matching_num_slabs": (C.c_int, [C.c_int]),
    "dm_time_matching_state_ints": (C.c_int, [C.c_int]),
    "dm_time_matching_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32, f32, f32, f32, vp, i64, vp, vp, vp]),
    "dm_time_matching_backward": (C.c_int, [vp, vp, vp, f32, vp, C.c_int, C.c_int, vp]),
    "dm_time_matching_backward_add": (C.c_int, [vp, vp, vp, f32, vp, vp, C.c_int, C.c_int, vp]),
    "dm_time_matching_forward_state": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32, f32, f32, f32, vp, i64, vp, vp, vp, vp]),
    "dm_time_matching_backward_state": (C.c_int, [vp, vp, vp, f32, vp, vp, C.c_int, C.c_int, vp, vp]),
    "dm_e1_compose": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dm_e1_compose_border": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dm_e1_chain": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dm_adam": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, vp]),
    "dm_adam_counted": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, vp, vp]),
    "dm_adam_counted_scaled": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, vp]),
    "dm_zscore_patch": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "dm_zscore_channels": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, C.c_int, i64, vp]),
    "dm_augment": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dm_gather_augment": (C.c_int, [vp, i64, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dm_gather_rows": (C.c_int, [vp, i64, vp, vp, C.c_int, i64, vp]),
    "dm_csr_block": (C.c_int, [vp, vp, vp, i64, vp, C.c_int, vp, i64, vp, vp]),
    "dm_augment_codes": (i64, [vp, i64, i64, vp, vp]),
    "dm_reorder_with_trajectories": (i64, [vp, i64, i64, vp, vp, vp, vp]),
}

_lib = None


class DynamorphHipError(RuntimeError):
    pass


class _CallDevice(threading.local):
    """Device of the tensors handed to the call being assembled (dynamorph_amd.ops._ptr records it; ops._op clears it at
    the start and at the end of every tensor-level wrapper, so an exception half-way through assembling a launch cannot
    leave a stale index behind)."""
    index = None


call_device = _CallDevice()

# entry points that only compute on the host (grid sizes, scratch sizes, capability queries): they touch no device, so
# they are bound without the device guard and do NOT consume the device recorded for the launch being assembled
HOST_ONLY_SUFFIXES = ("_num_blocks", "_num_slabs", "_scratch_floats", "_workspace_bytes", "_workspace_floats", "_supported", "_state_ints")
HOST_ONLY = ("dm_last_error", "dm_version", "dm_backward_precision", "dm_augment_codes", "dm_reorder_with_trajectories")


def is_host_only(name):
    return name in HOST_ONLY or name.endswith(HOST_ONLY_SUFFIXES)


def _guarded(fn):
    """The launching entry points take raw pointers and enqueue on whatever device is current, so the tensors' device is
    made current around the call (the reference hands non-zero gpu ids to its workers, run_VAE.py:78-85): a kernel
    launched on device 0's stream against device-N memory is a fault or a silent race.  The device is the one ops._ptr
    recorded for this call; it stays recorded until the wrapper returns (ops._op), so every pointer of a call is checked
    against it, whatever host-only queries run in between."""
    def call(*args):
        dev = call_device.index
        if dev is None:
            return fn(*args)
        import torch
        if dev == torch.cuda.current_device():
            return fn(*args)
        with torch.cuda.device(dev):
            return fn(*args)
    call.__name__ = getattr(fn, "__name__", "dm_call")
    return call


class _Library:
    """Attribute access to the C entry points; the launching ones sit behind the device guard."""

    def __init__(self, cdll):
        self._cdll = cdll
        for name in SIGNATURES:
            fn = getattr(cdll, name)
            setattr(self, name, fn if is_host_only(name) else _guarded(fn))


def load():
    """Load the library once; raise (never fall back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DynamorphHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dynamorph_amd/csrc`. There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = _Library(lib)
    return _lib


def check(rc, what):
    """Mirror the reference's only error convention: shape problems -> ValueError, the rest RuntimeError."""
    if rc == 0:
        return
    msg = load().dm_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg}")
    raise DynamorphHipError(f"{what}: HIP error {rc}: {msg}")
