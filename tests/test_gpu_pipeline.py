"""GPU: the callers on either side of the model (process_VAE I/O contract, run_training-style loop) and the
BASELINE.json stress configuration (4-channel 256x256 patches, 4096-entry codebook)."""
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import codes_gate, grad_gate, oracle_truth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def copy_of(module):
    import copy
    return copy.deepcopy(module)


def test_process_vae_pickle_contract(tmp_path, golden):
    """pipeline/patch_VAE.py:343-462: <well>_file_paths.pkl + <well>_static_patches.pkl + <weights>/model.pt in,
    <raw>/<model_name>/<well>_latent_space[_after].pkl out (protocol 4, (N,4096) float32)."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import process_VAE
    from oracle import vqvae_oracle as O
    raw, wdir = tmp_path / "raw", tmp_path / "weights" / "vqvae_test"
    raw.mkdir(); wdir.mkdir(parents=True)
    rng = np.random.RandomState(0)
    patches = rng.rand(5, 2, 1, 128, 128) * 1000 + 200        # (N,C,1,H,W) float64 as extract_patches writes them
    fs = [f"/data/C5-Site_0/{i}_0.h5" for i in range(5)]
    pickle.dump(fs, open(raw / "C5_file_paths.pkl", "wb"))
    pickle.dump(patches, open(raw / "C5_static_patches.pkl", "wb"))
    sd = {k: torch.from_numpy(v) for k, v in golden("g1_state_dict.npz").items()}
    torch.save(sd, wdir / "model.pt")
    cfg = SimpleNamespace(latent_encoding=SimpleNamespace(
        weights=str(wdir), channels=[0, 1], num_hiddens=16, num_residual_hiddens=32, num_embeddings=64,
        commitment_cost=0.25, network="VQ_VAE_z16", save_output=False, channel_mean=None, channel_std=None))
    process_VAE(str(raw), None, ["C5-Site_0"], cfg, gpu=0, batch_size=2)
    out = raw / "vqvae_test"
    zb = pickle.load(open(out / "C5_latent_space.pkl", "rb"))
    za = pickle.load(open(out / "C5_latent_space_after.pkl", "rb"))
    assert zb.shape == (5, 4096) and za.shape == (5, 4096) and zb.dtype == np.float32
    ref = O.OracleVQVAE(variant="z16")
    ref.load_state_dict(sd)
    x = torch.from_numpy(O.zscore_patch(np.squeeze(patches))).float()
    zb_ref, za_ref = O.encode_per_sample(ref, x)
    assert np.abs(zb - zb_ref.reshape(5, -1).numpy()).max() < 3e-4
    idx_ref = ref.vq.encode_inputs(zb_ref)
    m = dynamorph_amd.VQ_VAE_z16().to(DEV)
    m.load_state_dict(sd)
    idx = m.vq.encode_inputs(torch.from_numpy(zb).reshape(5, 16, 16, 16).to(DEV)).cpu()
    codes_gate(idx != idx_ref, zb_ref, ref.vq.w.weight.detach(), "process_VAE latents")
    # save_output (patch_VAE.py:464-489): 20 samples drawn with seed 0, reconstructed one by one through model(sample)[0]
    cfg.latent_encoding.save_output = True
    process_VAE(str(raw), None, ["C5-Site_0"], cfg, gpu=0, batch_size=2)
    np.random.seed(0)
    picks = sorted(set(int(i) for i in np.random.randint(0, 5, (20,))))
    for i in picks:
        rec = np.load(out / f"recon_{i}.npz")
        assert rec["sample"].shape == (2, 128, 128) and rec["output"].shape == (2, 128, 128)
        np.testing.assert_allclose(rec["sample"], x[i].numpy(), rtol=0, atol=0)
        with torch.no_grad():
            want = copy_of(ref)(x[i:i + 1])[0][0].numpy()
        assert np.abs(rec["output"] - want).max() < 5e-4
    cfg.latent_encoding.save_output = False
    with pytest.raises(ValueError, match="Error in loading model weights"):
        os.remove(wdir / "model.pt")
        process_VAE(str(raw), None, ["C5-Site_0"], cfg, gpu=0)


def test_stress_config_4ch_256px_k4096():
    """BASELINE.json configs[4] shapes at a small batch: forward, backward and indices vs the CPU oracle."""
    import dynamorph_amd
    from oracle import vqvae_oracle as O
    torch.manual_seed(3)
    kw = dict(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4))
    ref = O.OracleVQVAE(**kw)
    x = torch.randn(2, 4, 256, 256, generator=torch.Generator().manual_seed(9))
    m = dynamorph_amd.VQ_VAE(**kw).to(DEV)
    m.load_state_dict(ref.state_dict())
    ref.vq.chunk = 1                      # 268 MB of distances per patch at K = 4096 (float64: twice that)
    ld_r, g32, g64 = oracle_truth(ref, x)
    dec, ld = m(x.to(DEV))
    ld["total_loss"].backward()
    assert dec.shape == (2, 4, 256, 256)
    for k in ("recon_loss", "commitment_loss", "total_loss"):
        assert abs(float(ld[k]) - float(ld_r[k])) <= 2e-5, (k, float(ld[k]), float(ld_r[k]))
    ref2 = O.OracleVQVAE(**kw)
    ref2.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    zb = m.enc(x.to(DEV))
    assert zb.shape == (2, 16, 32, 32)
    zb_r = ref2.enc(x).detach()
    idx, idx_r = m.vq.encode_inputs(zb).cpu(), ref2.vq.encode_inputs(zb_r)
    codes_gate(idx != idx_r, zb_r, ref2.vq.w.weight.detach(), "stress configuration (K = 4096)")
    from test_gpu_model import BN_FED_BIASES
    grad_gate(m, g32, g64, skip=BN_FED_BIASES, floor=5e-4, what="stress configuration (K = 4096)")


def test_train_loop_mirror_runs_and_checkpoints(tmp_path):
    """run_training.py:455-551 loop: Adam, validation block, EarlyStopping checkpoint; loss must go down."""
    import dynamorph_amd
    from dynamorph_amd.train import train
    torch.manual_seed(0)
    np.random.seed(0)
    m = dynamorph_amd.VQ_VAE().to(DEV)
    data = torch.utils.data.TensorDataset(torch.randn(24, 2, 128, 128, generator=torch.Generator().manual_seed(1)))
    before = {k: v.clone() for k, v in m.state_dict().items()}

    class Scalars:                       # stands in for the SummaryWriter of run_training.py:501
        def __init__(self):
            self.rows = {}

        def add_scalar(self, key, value, epoch):
            self.rows.setdefault(key, []).append(float(value))
    w = Scalars()
    train(m, data, str(tmp_path), n_epochs=6, lr=2e-3, batch_size=8, device=DEV, transform=True,
          val_split_ratio=0.34, patience=10, writer=w)
    tr = w.rows["Loss/total_loss"]
    assert len(tr) == 6 and tr[-1] < tr[0], tr                     # the loss does go down
    assert len(w.rows["Val loss/total_loss"]) == 6
    ck = torch.load(tmp_path / "model.pt")
    assert list(ck.keys()) == list(before.keys())
    assert any(not torch.equal(ck[k].cpu(), before[k].cpu()) for k in ck if "weight" in k)
    assert int(m.enc[2].num_batches_tracked) > 0


@pytest.mark.parametrize("N,nin,hw", [(5, 1, 128), (3, 4, 128), (2, 3, 256)])
def test_encode_patches_per_sample_shape_sweep(N, nin, hw):
    """process_VAE semantics (batch-of-one enc -> vq calls, train-mode BatchNorm) for other channel counts and
    patch sizes: the batched per-sample-statistics encoder vs the oracle's loop."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    torch.manual_seed(300 + nin)
    kw = dict(num_inputs=nin, channel_var=np.ones(nin))
    ref = O.OracleVQVAE(**kw)
    m = dynamorph_amd.VQ_VAE(**kw).to("cuda:0")
    m.load_state_dict(ref.state_dict())
    x = torch.randn(N, nin, hw, hw, generator=torch.Generator().manual_seed(N))
    with torch.no_grad():
        zb_r, za_r = O.encode_per_sample(ref, x)
    zb, za = encode_patches(m, x, device="cuda:0", batch_size=4)
    zb_r = zb_r.reshape(N, -1).numpy()
    assert zb.shape == zb_r.shape
    np.testing.assert_allclose(zb, zb_r, rtol=2e-4, atol=2e-4)
    # codes: identical except at reference near-ties
    codes_gate((np.abs(za.reshape(za_r.shape) - za_r.numpy()) > 1e-3).any(axis=1), zb_r.reshape(za_r.shape), ref.vq.w.weight.detach(),
               "encode_patches")
    # the running statistics advanced by N batch-of-one calls
    assert int(m.enc[2].num_batches_tracked) == N
    np.testing.assert_allclose(m.enc[2].running_mean.cpu().numpy(), ref.enc[2].running_mean.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.enc[2].running_var.cpu().numpy(), ref.enc[2].running_var.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,nin", [(1, 2), (3, 2), (5, 4), (700, 2), (1024, 2)])
def test_latent_tail_kernel_against_the_layer_by_layer_path_and_the_oracle(B, nin):
    """enc.10 .. enc.12 of a patch inside one workgroup (csrc/latent_tail.hip, what encode_patches runs) vs the same layers
    as separate launches, and vs the oracle's batch-of-one loop: latents, and the running statistics / batch counters of
    ALL eight BatchNorm layers (the fused kernel writes the per-patch sums the replay kernel reads).  BatchNorm weights and
    biases are randomised so that gamma / beta / bias plumbing is exercised."""
    import copy
    import dynamorph_amd
    from dynamorph_amd import engine as E
    from oracle import vqvae_oracle as O
    torch.manual_seed(9000 + B)
    kw = dict(num_inputs=nin, channel_var=np.ones(nin))
    ref = O.OracleVQVAE(**kw)
    with torch.no_grad():
        for mod in ref.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.5, 0.5)
    m1 = dynamorph_amd.VQ_VAE(**kw).to(DEV)
    m1.load_state_dict(ref.state_dict())
    m2 = copy.deepcopy(m1)
    x = torch.randn(B, nin, 128, 128, generator=torch.Generator().manual_seed(B)).to(DEV)
    with torch.no_grad():
        z1, _ = E.encoder_forward(E.Layers(m1), x, per_sample=True, latents_only=True)
        z2, _ = E.encoder_forward(E.Layers(m2), x, per_sample=True)
    torch.cuda.synchronize()
    scale = z2.abs().max().item()
    assert (z1 - z2).abs().max().item() <= 2e-5 * scale, ((z1 - z2).abs().max().item(), scale)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        if "running" in k:
            assert torch.allclose(sd1[k], sd2[k], rtol=1e-5, atol=1e-7), k
        if "tracked" in k:
            assert int(sd1[k]) == int(sd2[k]) == B, k
    if B <= 5:
        with torch.no_grad():
            zb_r, _ = O.encode_per_sample(ref, x.cpu())
        assert (z1.cpu() - zb_r).abs().max().item() <= 3e-4 * max(1.0, zb_r.abs().max().item())
        sd_r = ref.state_dict()
        for k in sd1:
            if "running" in k:
                assert torch.allclose(sd1[k].cpu(), sd_r[k], rtol=2e-4, atol=2e-6), k


def test_encode_patches_constant_and_huge_patches():
    """Degenerate patches through the per-sample path (latent tail kernel included): a constant patch (every BatchNorm sees
    zero variance: 1 / sqrt(eps) scaling, variance clamped at 0), an all-zero patch, and one with values of 1e4 -- finite
    latents that match the oracle's batch-of-one calls."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    torch.manual_seed(77)
    ref = O.OracleVQVAE()
    m = dynamorph_amd.VQ_VAE().to(DEV)
    m.load_state_dict(ref.state_dict())
    x = torch.randn(5, 2, 128, 128, generator=torch.Generator().manual_seed(5))
    x[0] = 3.25
    x[1] = 0.0
    x[2] *= 1e4
    x[3, 0] = -1.5                      # one constant channel
    with torch.no_grad():
        zb_r, za_r = O.encode_per_sample(ref, x)
    zb, za = encode_patches(m, x, device=DEV, batch_size=8)
    zb_r = zb_r.reshape(5, -1).numpy()
    assert np.isfinite(zb).all() and np.isfinite(za).all()
    scale = np.abs(zb_r).max(axis=1, keepdims=True) + 1e-6
    assert (np.abs(zb - zb_r) / scale).max() <= 2e-3, (np.abs(zb - zb_r) / scale).max(axis=1)
    # the non-degenerate patches to the usual tolerance
    np.testing.assert_allclose(zb[3:], zb_r[3:], rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize("kw,N", [({}, 5), (dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512), 3)])
def test_encode_patches_z32_per_sample(kw, N):
    """process_VAE semantics for VQ_VAE_z32 (the network config_example.yml names), default and example widths: the batched
    per-sample-statistics encoder against the oracle's batch-of-one loop, BatchNorm running statistics included."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    torch.manual_seed(77 + N)
    ref = O.OracleVQVAEz32(**kw)
    m = dynamorph_amd.VQ_VAE_z32(**kw).to("cuda:0")
    m.load_state_dict(ref.state_dict())
    x = torch.randn(N, 2, 128, 128, generator=torch.Generator().manual_seed(N))
    with torch.no_grad():
        zb_r = torch.cat([ref.enc(x[i:i + 1]) for i in range(N)], 0)            # patch_VAE.py:445-452, train mode
        za_r = torch.cat([ref.vq(zb_r[i:i + 1])[0] for i in range(N)], 0)
    zb, za = encode_patches(m, x, device="cuda:0", batch_size=2)
    assert zb.shape == (N, zb_r[0].numel())
    np.testing.assert_allclose(zb, zb_r.reshape(N, -1).numpy(), rtol=3e-4, atol=3e-4)
    codes_gate((np.abs(za.reshape(za_r.shape) - za_r.numpy()) > 1e-3).any(axis=1), zb_r, ref.vq.w.weight.detach(), "encode_patches z32")
    assert int(m.enc[1].num_batches_tracked) == N
    np.testing.assert_allclose(m.enc[1].running_mean.cpu().numpy(), ref.enc[1].running_mean.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.enc[4].running_var.cpu().numpy(), ref.enc[4].running_var.numpy(), rtol=1e-4, atol=1e-6)


def test_encode_patches_wide_vq_vae_per_sample():
    """process_VAE semantics for VQ_VAE at the example widths (64 / 64 / 512): per-sample statistics through the
    implicit-GEMM kernels (border-bias first conv, statistics slabs grouped by sample)."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    kw = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)
    torch.manual_seed(64)
    ref = O.OracleVQVAE(**kw)
    m = dynamorph_amd.VQ_VAE(**kw).to("cuda:0")
    m.load_state_dict(ref.state_dict())
    N = 3
    x = torch.randn(N, 2, 128, 128, generator=torch.Generator().manual_seed(N))
    with torch.no_grad():
        zb_r, za_r = O.encode_per_sample(ref, x)
    zb, za = encode_patches(m, x, device="cuda:0", batch_size=2)
    np.testing.assert_allclose(zb, zb_r.reshape(N, -1).numpy(), rtol=3e-4, atol=3e-4)
    codes_gate((np.abs(za.reshape(za_r.shape) - za_r.numpy()) > 1e-3).any(axis=1), zb_r, ref.vq.w.weight.detach(), "encode_patches, wide VQ_VAE")
    assert int(m.enc[2].num_batches_tracked) == N
    np.testing.assert_allclose(m.enc[2].running_var.cpu().numpy(), ref.enc[2].running_var.numpy(), rtol=1e-4, atol=1e-6)


def test_encode_patches_sharded_single_process_is_encode_patches():
    import copy
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches, encode_patches_sharded
    torch.manual_seed(5)
    m1 = dynamorph_amd.VQ_VAE().to("cuda:0")
    m2 = copy.deepcopy(m1)
    x = torch.randn(5, 2, 128, 128, generator=torch.Generator().manual_seed(1))
    a = encode_patches(m1, x, device="cuda:0", batch_size=2)
    b = encode_patches_sharded(m2, x, device="cuda:0", batch_size=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("on_device", [False, True])
def test_encode_patches_pipeline_is_independent_of_the_batching(on_device):
    """The copy-in / encode / copy-out pipeline (two pinned staging buffers per direction, three streams): many ragged
    batches give bit for bit what one batch gives, in input order -- per-sample BatchNorm statistics keep patches
    independent; also with the z-score on the device (float64 staging)."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    torch.manual_seed(9)
    m = dynamorph_amd.VQ_VAE().to("cuda:0")
    x = torch.randn(23, 2, 128, 128, generator=torch.Generator().manual_seed(4))
    x = (x.double() * 300 + 1000) if on_device else x
    one = encode_patches(m, x, device="cuda:0", batch_size=64, zscore_on_device=on_device)
    for bs in (5, 1, 23, 11):
        many = encode_patches(m, x.numpy() if bs == 11 else x, device="cuda:0", batch_size=bs, zscore_on_device=on_device)
        assert many[0].shape == (23, 16 * 16 * 16) and many[0].dtype == np.float32
        assert np.array_equal(one[0], many[0]) and np.array_equal(one[1], many[1]), bs
    empty = encode_patches(m, x[:0], device="cuda:0")
    assert empty[0].shape[0] == 0
    # pinned input is sent without staging; results above DM_PINNED_RESULT_BYTES come back into pageable memory
    pinned_in = encode_patches(m, x.pin_memory(), device="cuda:0", batch_size=6, zscore_on_device=on_device)
    os.environ["DM_PINNED_RESULT_BYTES"] = "0"
    try:
        pageable_out = encode_patches(m, x, device="cuda:0", batch_size=6, zscore_on_device=on_device)
    finally:
        del os.environ["DM_PINNED_RESULT_BYTES"]
    for got in (pinned_in, pageable_out):
        assert np.array_equal(one[0], got[0]) and np.array_equal(one[1], got[1])


def test_encode_patches_in_eval_mode_uses_running_statistics():
    """model.eval() before encode_patches (never done by the reference path, but a caller may): BatchNorm takes its
    running statistics for every sample -- the coefficients are shared, not per sample."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    torch.manual_seed(5)
    ref = O.OracleVQVAE()
    x = torch.randn(6, 2, 128, 128, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref(x)                                   # one train-mode call: running statistics away from their initial values
    m = dynamorph_amd.VQ_VAE().to(DEV)
    m.load_state_dict(ref.state_dict())
    ref.eval(); m.eval()
    with torch.no_grad():
        zb_r = ref.enc(x)
        za_r = ref.vq(zb_r)[0]
    zb, za = encode_patches(m, x, device=DEV, batch_size=4)
    np.testing.assert_allclose(zb, zb_r.reshape(6, -1).numpy(), rtol=2e-4, atol=2e-4)
    codes_gate((np.abs(za.reshape(za_r.shape) - za_r.numpy()) > 1e-3).any(axis=1), zb_r, ref.vq.w.weight.detach(), "eval mode")
    assert int(m.enc[2].num_batches_tracked) == int(ref.enc[2].num_batches_tracked)      # eval: nothing advanced


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_encode_patches_on_a_device_that_is_not_current():
    """run_VAE.py:78-85 hands non-zero gpu ids to its workers: the kernels must run on the tensors' device whatever the
    current device is."""
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    torch.manual_seed(6)
    m0 = dynamorph_amd.VQ_VAE().to("cuda:0")
    m1 = dynamorph_amd.VQ_VAE().to("cuda:1")
    m1.load_state_dict(m0.state_dict())
    x = torch.randn(3, 2, 128, 128)
    torch.cuda.set_device(0)
    a = encode_patches(m0, x, device="cuda:0", batch_size=2)
    b = encode_patches(m1, x, device="cuda:1", batch_size=2)
    assert torch.cuda.current_device() == 0
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------------------------ the dataset side of train(): run_training.py:880
@pytest.mark.parametrize("dtype,stats", [("float64", "list"), ("float32", "list"), ("float32", "np64"), ("float64", "np32")])
def test_zscore_channels_is_numpy_bit_for_bit(dtype, stats):
    """pipeline/train_utils.py:228-250 with given statistics, then .astype(np.float32) (run_training.py:880): the device
    expression runs in the type numpy's promotion rules pick (float32 stays float32 against Python floats, goes to double
    against float64 scalars) and lands on the same bits."""
    from dynamorph_amd import ops
    from dynamorph_amd import train_utils as TU
    rng = np.random.default_rng(3)
    raw = (rng.normal(0.3, 0.2, (37, 2, 32, 32)) * np.array([1.0, 300.0]).reshape(1, 2, 1, 1)).astype(dtype)
    mean, std = [0.4, 37.125], [0.05, 11.3]
    if stats == "np64":
        mean, std = list(np.array(mean, np.float64)), list(np.array(std, np.float64))
    elif stats == "np32":
        mean, std = list(np.array(mean, np.float32)), list(np.array(std, np.float32))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        want = TU.zscore(raw, mean, std).astype(np.float32)
    got = ops.zscore_channels(torch.from_numpy(raw).to(DEV), mean, std).cpu().numpy()
    assert np.array_equal(got, want)


def test_upload_zscored_feeds_train_from_the_device(tmp_path):
    """feed.upload_zscored: the pickled (N, C, 1, H, W) float64 patches z-scored on the device equal the host expression of
    run_training.py:880 bit for bit, and train() takes the device tensor as its dataset (resident feed, used where it lies)
    with the same result as from the host copy."""
    import contextlib, copy, io
    import dynamorph_amd
    from dynamorph_amd import feed as F
    from dynamorph_amd import train_utils as TU
    from dynamorph_amd.train import train
    rng = np.random.default_rng(5)
    raw = rng.normal(0.5, 0.1, (24, 2, 1, 128, 128))
    mean, std = [0.49, 0.51], [0.11, 0.09]
    with contextlib.redirect_stdout(io.StringIO()):
        host = TU.zscore(np.squeeze(raw), mean, std).astype(np.float32)
    dev_t = F.upload_zscored(raw, mean, std, DEV, chunk_bytes=5 * 2 * 128 * 128 * 8)      # (several ragged chunks)
    assert dev_t.is_cuda and dev_t.dtype == torch.float32 and np.array_equal(dev_t.cpu().numpy(), host)
    torch.manual_seed(0)
    m0 = dynamorph_amd.VQ_VAE().to(DEV)
    out = {}
    for name, data in (("host", torch.utils.data.TensorDataset(torch.from_numpy(host))), ("device", torch.utils.data.TensorDataset(dev_t))):
        m = copy.deepcopy(m0)
        np.random.seed(9)
        st = {}
        with contextlib.redirect_stdout(io.StringIO()):
            train(m, data, str(tmp_path / name), n_epochs=2, lr=1e-3, batch_size=8, device=DEV, transform=True,
                  val_split_ratio=0.25, patience=5, feed="resident", stats=st)
        assert st["feed"] == "resident"
        out[name] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in out["host"].items():
        assert torch.equal(v, out["device"][k]), k


def test_device_dataset_is_resident_whatever_the_budget_and_bare_tensors_take_the_sync_loop(tmp_path, monkeypatch):
    """(1) feed='auto' with a dataset that already lives in HBM (upload_zscored's result) must not count it against
    the resident budget and end in the streaming path (which gathers on the host); (2) a BARE tensor as the dataset in the
    synchronous loop is indexed directly (dataset[ids][0] would be one sample of the batch); (3) feed='stream' with a device
    dataset is a clear error."""
    import contextlib, copy, io
    import dynamorph_amd
    from dynamorph_amd import feed as F
    from dynamorph_amd.train import train
    torch.manual_seed(3)
    host = torch.randn(24, 2, 128, 128)
    dev_t = host.to(DEV)
    monkeypatch.setenv("DM_RESIDENT_BYTES", "1024")                     # far below the dataset's 3 MB
    m0 = dynamorph_amd.VQ_VAE().to(DEV)
    out = {}
    for name, data, feed in (("auto-device-bare", dev_t, "auto"), ("sync-bare", host, "sync"),
                             ("sync-dataset", torch.utils.data.TensorDataset(host), "sync")):
        m = copy.deepcopy(m0)
        np.random.seed(9)
        st = {}
        with contextlib.redirect_stdout(io.StringIO()):
            train(m, data, str(tmp_path / name), n_epochs=2, lr=1e-3, batch_size=8, device=DEV, transform=None,
                  val_split_ratio=0.25, patience=5, feed=feed, stats=st)
        assert st["feed"] == ("resident" if feed == "auto" else "sync"), (name, st["feed"])
        out[name] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for k, v in out["sync-dataset"].items():
        assert torch.equal(v, out["sync-bare"][k]), k
        assert torch.equal(v, out["auto-device-bare"][k]), k
    with pytest.raises(ValueError, match="host memory"):
        F.Feed(dev_t, DEV, mode="stream", batch_size=8)
    with pytest.raises(ValueError, match="contiguous float32"):
        F.Feed(dev_t.double(), DEV, mode="auto", batch_size=8)
