"""`not gpu`: the C-ABI library loads and exports every symbol include/dynamorph_hip.h declares; host-side
argument checking works without a device (no compute call is made here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dynamorph_hip.h")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "dynamorph_amd", "csrc")])
    from dynamorph_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from dynamorph_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported by the library"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in dynamorph_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_struct_layouts_match_the_header():
    from dynamorph_amd import _lib
    assert ctypes.sizeof(_lib.Operand) == 40          # 3 pointers + int64 + 2 int32
    assert ctypes.sizeof(_lib.WeightView) == 64          # 6 words + scratch pointer + scratch size
    assert ctypes.sizeof(_lib.Epilogue) == 24 + 40 + 24   # bias, bias_border, relu, stats_per_tile | mask | 3 pointers
    assert _lib.Epilogue.mask.offset == 24 and _lib.Epilogue.stats.offset == 80


def test_version_and_host_only_queries(lib):
    assert lib.dm_vq_backward_num_slabs(524288, 64, 16) == 512 and lib.dm_vq_backward_num_slabs(262144, 4096, 16) == 128
    assert lib.dm_vq_backward_num_slabs(3000, 64, 16) == 3 and lib.dm_vq_backward_num_slabs(1 << 22, 65536, 16) == 32
    assert lib.dm_version() == 128          # 128: dm_wgrad_t_affine2_supported (T as an AFFINE2 operand of the one-pass weight gradient), dm_conv1x1_bwd_fused at 64 channels; 127: streaming kernels of the wide family (wide_stream.hip), one-pass 64-channel weight gradients, dm_bn_backward_finalize with count 0 (eval mode); 126: vq_cells_kernel (large codebooks), dm_adam_counted_scaled, split-bf16 gradient kernels retired; 125: dm_conv3x3_bwd_fused on 32 x 32 latents (CD = 32), the training decoder tail in 64-column tiles; 124: dm_zscore_channels; 123: dm_time_matching_forward_state / _backward_state; 122: dm_reorder_with_trajectories; 121: dm_conv4x4s2_bwd_fused; 120: dm_conv3x3_bwd_fused; 119: dm_convt_bwd_fused; 118: dm_conv1x1_bwd_fused; 117: dm_vq_loss_finalize_tm, dm_time_matching_backward_add; 116: dm_vq_forward_join; 115: dm_gather_augment, dm_gather_rows, dm_csr_block, dm_augment_codes; 114: latent tail from a2; 113: dm_latent_tail_forward; 112: the fused decoder tail takes any width that is a multiple of 4
    assert lib.dm_latent_tail_supported(16, 32, 16, 16, 2) == 1 and lib.dm_latent_tail_supported(16, 32, 32, 32, 2) == 0
    assert lib.dm_latent_tail_supported(64, 64, 16, 16, 2) == 0
    assert lib.dm_dec_tail_supported(4, 4, 128, 128) == 1 and lib.dm_dec_tail_supported(4, 2, 64, 64) == 1
    assert lib.dm_dec_tail_supported(4, 2, 64, 66) == 0 and lib.dm_dec_tail_supported(8, 2, 64, 64) == 0
    assert lib.dm_dec_tail_num_blocks(1, 8, 64) == 1 and lib.dm_dec_tail_num_blocks(1, 8, 128) == 3 and lib.dm_dec_tail_num_blocks(1, 8, 56) == 1
    # header + pair-interleaved codebook (exact kernel) + MFMA A operand + norms + lane-ordered rows (csrc/vq.hip)
    # header + pair-interleaved codebook (exact kernel) + MFMA A operand + norms + lane-ordered rows + counter replicas
    # ... + the bf16-split A operand (twice the f32 one)
    # (<= 64 codes: 1024 per-workgroup counter rows of 72 ints instead of the 64 replicas)
    assert lib.dm_vq_workspace_bytes(64, 16) == (32 + 64 * 16 + 64 * 16 + 2 * 64 * 16 + 64 + 64 * 16 + 1024 * 72) * 4
    # (64 < K <= 4096 at embedding_dim 16: + the permuted bf16 operand and norms of vq_cells_kernel, 16 + 1 floats per code)
    assert lib.dm_vq_workspace_bytes(4096, 16) == (32 + 4096 * 16 * 5 + 4096 + 16 * 4096 + 17 * 4096) * 4
    assert lib.dm_vq_workspace_bytes(100, 16) - lib.dm_vq_workspace_bytes(100, 8) > 17 * 128 * 4        # (K rounded up to 128)
    assert lib.dm_vq_num_blocks(524288) == 2048
    assert lib.dm_conv4x4s2_bwd_fused_supported(16, 16, 16, 16) == 1 and lib.dm_conv4x4s2_bwd_fused_supported(16, 8, 32, 32) == 0
    assert lib.dm_conv4x4s2_bwd_fused_num_blocks(2048, 16, 16, 16, 16) == 256 and lib.dm_conv4x4s2_bwd_fused_num_blocks(5, 16, 16, 16, 16) == 5
    assert lib.dm_conv3x3_bwd_fused_supported(16, 16, 16, 16) == 1 and lib.dm_conv3x3_bwd_fused_supported(32, 16, 16, 16) == 1
    assert lib.dm_conv3x3_bwd_fused_supported(32, 16, 32, 32) == 1 and lib.dm_conv3x3_bwd_fused_supported(64, 64, 16, 16) == 0
    assert lib.dm_conv3x3_bwd_fused_supported(32, 16, 32, 48) == 0 and lib.dm_conv3x3_bwd_fused_num_blocks(3, 32, 16, 32, 32) == 3
    assert lib.dm_conv3x3_bwd_fused_num_blocks(2048, 32, 16, 16, 16) == 256 and lib.dm_conv3x3_bwd_fused_num_blocks(2048, 16, 16, 16, 16) == 512
    assert lib.dm_convt_bwd_fused_supported(8, 4, 32, 32) == 1 and lib.dm_convt_bwd_fused_supported(16, 8, 16, 16) == 1
    assert lib.dm_convt_bwd_fused_supported(8, 4, 32, 16) == 0 and lib.dm_convt_bwd_fused_supported(4, 4, 64, 64) == 0
    assert lib.dm_convt_bwd_fused_num_blocks(2048, 8, 4, 32, 32) == 768 and lib.dm_convt_bwd_fused_num_blocks(2, 16, 8, 16, 16) == 4
    assert lib.dm_convt_bwd_fused_num_blocks(2048, 16, 8, 16, 16) == 512
    assert lib.dm_conv1x1_bwd_fused_supported(16, 32, 16, 16) == 1 and lib.dm_conv1x1_bwd_fused_supported(16, 32, 8, 8) == 0
    assert lib.dm_conv1x1_bwd_fused_supported(48, 48, 16, 16) == 0 and lib.dm_conv1x1_bwd_fused_num_blocks(2048, 16, 32, 16, 16) == 512
    # 64 -> 64 channels (wide_stream.hip): one statistics / weight slab per WAVE
    assert lib.dm_conv1x1_bwd_fused_supported(64, 64, 32, 32) == 1 and lib.dm_conv1x1_bwd_fused_num_blocks(768, 64, 64, 32, 32) == 2048
    assert lib.dm_conv1x1_bwd_fused_num_blocks(3, 64, 64, 32, 32) == 48
    assert lib.dm_wgrad_t_affine2_supported(64, 32, 32, 32, 4) == 1 and lib.dm_wgrad_t_affine2_supported(64, 64, 32, 32, 3) == 0
    assert lib.dm_wgrad_t_affine2_supported(16, 8, 32, 32, 4) == 0
    assert lib.dm_conv1x1_bwd_fused_num_blocks(3, 16, 32, 32, 32) == 12
    assert lib.dm_vq_forward_join_supported(16, 64, 16, 16) == 1 and lib.dm_vq_forward_join_supported(16, 4096, 32, 32) == 0
    assert lib.dm_vq_forward_join_supported(8, 64, 16, 16) == 0 and lib.dm_vq_forward_join_supported(16, 64, 10, 10) == 0
    assert lib.dm_conv4x4s2_num_blocks(2048, 3, 8, 128, 128, 1) == 2048 * 8   # one slab per tile (per-sample stats)
    assert lib.dm_conv4x4s2_num_blocks(2048, 3, 8, 128, 128, 0) == 768        # persistent grid
    assert lib.dm_conv4x4s2_num_blocks(1, 3, 8, 100, 100, 0) == 1             # not tileable by the MFMA path: generic kernel, one slab per sample
    assert lib.dm_conv4x4s2_num_blocks(1, 3, 8, 101, 100, 0) == -1            # odd height: no 4x4/s2 output grid
    assert lib.dm_conv3x3_num_blocks(4, 16, 16, 16, 16, 9, 0, 0) == 4
    assert lib.dm_conv3x3_num_blocks(4, 32, 16, 16, 16, 9, 0, 1) == 8         # 32 input channels: 8-row tiles
    assert lib.dm_wgrad_num_blocks(2048, 8, 3, 64, 64, 4) == 512     # persistent grid cap
    assert lib.dm_wgrad_num_blocks(1, 8, 3, 64, 64, 5) == -1
    assert lib.dm_head_num_blocks(2048, 128, 128) == 2048
    assert lib.dm_head_supported(4, 2) == 1 and lib.dm_head_supported(16, 4) == 1 and lib.dm_head_supported(12, 2) == 0
    # scratch for the re-laid weights: none for register-resident shapes, (passes x chunks x block) floats otherwise
    assert lib.dm_conv3x3_scratch_floats(16, 16, 16, 16, 9, 0, 0) == 0
    assert lib.dm_conv3x3_scratch_floats(64, 64, 32, 32, 9, 0, 0) == 8 * 6348          # 1 pass x 8 chunks of 8 channels
    assert lib.dm_conv3x3_scratch_floats(64, 64, 33, 32, 9, 0, 0) == 0                 # not tileable by 8 x 16: generic kernel
    assert lib.dm_conv4x4s2_scratch_floats(8, 16, 64, 64, 0) == 0 and lib.dm_conv4x4s2_scratch_floats(8, 16, 64, 64, 1) > 0
    assert lib.dm_conv4x4s2_scratch_floats(32, 64, 64, 64, 0) > 0
    assert lib.dm_channel_stats_num_blocks(20, 16, 16, 16) == 1 and lib.dm_channel_stats_num_blocks(2048, 16, 16, 16) == 64   # 32 samples per workgroup


def test_argument_errors_are_reported_before_any_launch(lib):
    """negative return + message, nothing is enqueued (no device needed)."""
    from dynamorph_amd import _lib
    rc = lib.dm_vq_forward(None, None, None, None, None, None, 1, 16, 64, 16, 16, None, 0, None)
    assert rc == -1 and b"NULL" in lib.dm_last_error()
    rc = lib.dm_vq_decode(None, None, None, 1, 16, 64, 16, 16, None)
    assert rc == -1
    op = _lib.Operand(None, None, None, 0, 0, 0)
    rc = lib.dm_apply(ctypes.byref(op), None, None, 1, 1, 4, 4, None)
    assert rc == -1 and b"p0 is NULL" in lib.dm_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "dm_apply")
    rc = lib.dm_adam(None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, None, None)
    assert rc == -1
    args = _lib.LatentTailArgs()
    args.B, args.C, args.CR, args.H, args.W, args.nres = 4, 16, 32, 16, 16, 2
    assert lib.dm_latent_tail_forward(ctypes.byref(args), None) == -1 and b"NULL" in lib.dm_last_error()
    args.C = 64
    assert lib.dm_latent_tail_forward(ctypes.byref(args), None) == -1 and b"built for 16 channels" in lib.dm_last_error()
    assert lib.dm_latent_tail_forward(None, None) == -1
    # round 4 entry points: argument errors before any launch
    op2 = _lib.Operand(None, None, None, 0, 0, 0)
    assert lib.dm_conv1x1_bwd_fused(ctypes.byref(op2), None, None, None, None, None, None, 1, 16, 32, 16, 16, None) == -1
    assert lib.dm_conv3x3_bwd_fused(ctypes.byref(op2), None, None, None, None, None, None, None, None, 1, 16, 16, 16, 16, None) == -1
    assert lib.dm_conv4x4s2_bwd_fused(ctypes.byref(op2), None, None, None, None, None, None, 1, 16, 16, 16, 16, None) == -1
    assert lib.dm_convt_bwd_fused(None, None, None, None, None, None, 0, 1, 8, 4, 32, 32, None) == -1
    assert lib.dm_gather_augment(None, 0, None, None, None, None, 1, 2, 128, None) == -1
    assert lib.dm_gather_rows(None, 0, None, None, 1, 16, None) == -1
    assert lib.dm_csr_block(None, None, None, 0, None, 1, None, 1, None, None) == -1
    assert lib.dm_vq_forward_join(None, None, None, None, None, None, None, None, None, 1, 16, 64, 16, 16, None, 0, None) == -1
    assert lib.dm_augment_codes(None, 0, 1, None, None) == -1
    import numpy as np
    raw = np.array([3, 3, 1, 2, 0, 3, 7, 7], dtype=np.uint32)        # words & 3: 3 3 1 2 | 0 3 | 3 3
    fl, ro = np.zeros(2, np.int32), np.zeros(2, np.int32)
    used = lib.dm_augment_codes(raw.ctypes.data, len(raw), 2, fl.ctypes.data, ro.ctypes.data)
    assert used == 6 and fl.tolist() == [1, 0] and ro.tolist() == [2, 3]
    assert lib.dm_augment_codes(raw.ctypes.data, 3, 2, fl.ctypes.data, ro.ctypes.data) == -1      # ran out of words
    assert lib.dm_time_matching_forward_state(None, None, 4, 32, 0, 0.0, 0.0, 0.0, 0.0, None, 0, None, None, None, None) == -1
    assert lib.dm_time_matching_backward_state(None, None, None, 1.0, None, None, 4, 32, None, None) == -1
    assert lib.dm_time_matching_state_ints(2048) == 4 + 32 * 64 and lib.dm_time_matching_state_ints(1) == 5
    assert lib.dm_time_matching_num_slabs(2048) == 4 * (32 * 33 // 2) and lib.dm_time_matching_num_slabs(64) == 16
    assert lib.dm_zscore_channels(None, 1, 1, 1, None, None, None, 1, 2, 16384, None) == -1
    assert lib.dm_reorder_with_trajectories(None, 0, 1, None, None, None, None) == -4
    # three samples, 0 <-> 2 adjacent: words & 3 = 3 (rejected), 1 -> second of {0, 1, 2} = 1 alone; then {0, 2}: word & 1 = 1
    # -> 2, followed by its trajectory [2, 0]
    raw = np.array([7, 5, 3], dtype=np.uint32)
    ptr, idx, order = np.array([0, 1, 1, 2], np.int64), np.array([2, 0], np.int64), np.zeros(3, np.int64)
    err = ctypes.c_int64(-1)
    assert lib.dm_reorder_with_trajectories(raw.ctypes.data, 3, 3, ptr.ctypes.data, idx.ctypes.data, order.ctypes.data, ctypes.byref(err)) == 3
    assert order.tolist() == [1, 2, 0]
    assert lib.dm_reorder_with_trajectories(raw.ctypes.data, 1, 3, ptr.ctypes.data, idx.ctypes.data, order.ctypes.data, ctypes.byref(err)) == -1
    rc = lib.dm_dec_tail_train(None, None, None, None, None, None, None, 0, None, None, None, None, None, None, 1, 4, 2, 64, 66, None)
    assert rc == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dynamorph_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DynamorphHipError):
        _lib.load()


def test_backward_precision_split_is_retired():
    """dm_backward_precision: the exact fp32 chain is the only arithmetic built; asking for the split-bf16 opt-in of earlier
    rounds is an error (host-only call, no GPU needed), the query answers "f32"."""
    from dynamorph_amd import ops
    assert ops.backward_precision() == "f32"
    assert ops.backward_precision("f32") == "f32"
    import pytest
    with pytest.raises(ValueError, match="not built"):
        ops.backward_precision("split-bf16")
    assert ops.backward_precision() == "f32"


def test_band_kernel_has_no_scratch_behind_its_counted_wait(tmp_path):
    """conv3x3_bwd_kernel<32, 512, BAND = true> (csrc/conv3x3_bwd.hip) orders its global -> LDS halo requests with a hand-counted
    `s_waitcnt vmcnt(RPW)`: the count assumes that the only vector-memory operations issued after the requests are the RPW
    stores of dx.  A register spill is a vector-memory operation on gfx9 and would break that count silently, so the build
    is checked: the instantiation has no scratch and no scratch instruction (hipcc cross-compiles here, no GPU needed)."""
    import re
    import subprocess
    src = os.path.join(ROOT, "dynamorph_amd", "csrc", "conv3x3_bwd.hip")
    asm = tmp_path / "conv3x3_bwd.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-S", "--cuda-device-only", "-o", str(asm), src], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    name = "_ZN12_GLOBAL__N_118conv3x3_bwd_kernelILi32ELi512ELb1EEE"
    start = text.index("\n" + name)
    body = text[start:text.index("s_endpgm", start)]
    assert "global_load_lds_dwordx4" in body                      # it IS the band form (the halo rows come in by LDS-DMA)
    assert "scratch_" not in body
    meta = text[text.index(".amdhsa_kernel " + name):]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta[:meta.index(".end_amdhsa_kernel")])
