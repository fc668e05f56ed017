"""GPU: the MODEL at BASELINE.json's batch sizes against the CPU oracle (missing 4) -- not only the
size-independent properties of test_gpu_model.py::test_round_trip_properties_full_size.

  C3  one FusedTrainer.step on 2048 patches  vs  oracle.train_step (run_training.py:504-532's body at that batch):
      losses <= 1e-5, codes over all 524 288 positions (conftest.codes_gate), every gradient on the float64 yardstick;
  C2  encode_patches on 1024 patches          vs  oracle.encode_per_sample (patch_VAE.py:445-452's loop).

The oracle evaluates the VectorQuantizer distances in chunks of 64 samples (OracleVQ.chunk; the reference's single
expression is 8.6 GB at this batch, 17 GB in float64) -- bit-identical, tests/test_oracle.py."""
import copy
import os
import gc

import numpy as np
import pytest
import torch

from conftest import codes_gate, grad_gate
from test_gpu_model import BN_FED_BIASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _host_memory_ok(need_gb):
    try:
        import psutil
        return psutil.virtual_memory().available >= need_gb * (1 << 30)
    except Exception:
        return True


@pytest.mark.timeout(900)
def test_c3_training_step_at_batch_2048_against_the_oracle():
    if not _host_memory_ok(24):
        pytest.skip("host has < 24 GB available for the float64 oracle run at B = 2048")
    import dynamorph_amd
    from dynamorph_amd.train import FusedTrainer
    from oracle import vqvae_oracle as O
    B = 2048
    torch.manual_seed(2048)
    ref = O.OracleVQVAE()
    ref.vq.chunk = 64
    x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234))
    m = dynamorph_amd.VQ_VAE().to(DEV)
    m.load_state_dict(ref.state_dict())
    xd = x.to(DEV)

    # codes of the untouched model on the whole batch, HIP encoder vs the oracle's (train-mode batch statistics)
    probe = copy.deepcopy(ref)
    with torch.no_grad():
        zb_r = probe.enc(x)
        idx_r = probe.vq.encode_inputs(zb_r)
        mp = copy.deepcopy(m)
        idx = mp.vq.encode_inputs(mp.enc(xd)).cpu()
    assert idx.shape == (B, 16, 16)
    codes_gate(idx != idx_r, zb_r, probe.vq.w.weight.detach(), "C3, B = 2048: all 524 288 positions")
    del probe, mp, zb_r
    gc.collect()
    # The gradients below are sums over 524 288 latent positions of incoherent terms (random initialisation): a sum of N such
    # terms has the size of sqrt(N) of them, so ONE position that takes another code moves every gradient by ~1/724 of its scale
    # -- far above any fp32 error.  Where codes_gate has admitted flips (reference near-ties only), the yardsticks are evaluated
    # with the HIP path's codes, so that gradients are compared under the same discrete choices.  (Round 5: the paired
    # first-convolution kernel's a1 is closer to float64 than kernel A's -- rms 3.3e-8 against 4.1e-8, the reference's own 9.6e-8,
    # tools/exp/e1_accuracy.py -- and turns 2 near-ties of this batch the other way; with the reference's codes in the yardstick
    # nine gradients sat 2-150 x beyond the reference's error, tools/exp/c3_grad_ratios.py.)
    nflip = int((idx != idx_r).sum())
    # (codes_gate has already restricted flips to the reference's own near-ties and to 1e-5 of the positions; at this batch
    #  rounds 5-6 see 0-2, and more than 4 would mean the encoder's latents drifted: the forced comparison must not hide that)
    assert nflip <= 4, nflip

    def yardsticks(model32):
        """(float64 gradients, fp32 gradients, fp32 losses, the stepped fp32 model) of `model32` on the batch; the float64
        graph is freed before the fp32 run starts."""
        m64 = copy.deepcopy(model32).double()
        _, l64 = m64(x.double())
        l64["total_loss"].backward()
        g64_ = {k: p.grad for k, p in m64.named_parameters() if p.grad is not None}
        del l64, m64
        gc.collect()
        opt_ = O.make_adam(model32, 1e-4)
        _, l32 = model32(x)
        l32["total_loss"].backward()
        g32_ = {k: p.grad.clone() for k, p in model32.named_parameters() if p.grad is not None}
        opt_.step()
        l32 = {k: float(v) for k, v in l32.items()}
        gc.collect()
        return g64_, g32_, l32

    unforced = None
    if nflip:
        # the comparison WITHOUT the forcing is kept visible (printed below, not asserted): the reference's own codes in
        # both yardsticks
        unforced = yardsticks(copy.deepcopy(ref))[:2]
        ref.vq.force_idx = idx.clone()
        print(f"{nflip} near-tie codes differ: the gated float64 and fp32 yardsticks use the HIP path's codes")
    g64, g32, ld_r = yardsticks(ref)

    tr = FusedTrainer(m, lr=1e-4, use_graph=True)
    vals = tr.step(xd).tolist()
    for i, k in enumerate(("recon_loss", "commitment_loss", "total_loss")):
        assert abs(vals[i] - ld_r[k]) <= 1e-5, (k, vals[i], ld_r[k])                 # the north star's tolerance
    assert abs(vals[3] - ld_r["perplexity"]) <= 1e-3 * ld_r["perplexity"]
    tr.expose_grads()
    # (each tensor's ratio is the quotient of two fp32 error maxima, 35 of them: 34 sit below 1.5; the documented outlier is
    #  enc.2.weight at 1.53 -- 6.1e-7 against the reference's 4.0e-7 on a gradient of scale 1.1e-3 -- with its own factor)
    grad_gate(m, g32, g64, skip=BN_FED_BIASES, factor=1.5, factor_for={"enc.2.weight": 2.0}, what="C3 step at B = 2048")
    if unforced is not None:
        grad_gate(m, unforced[1], unforced[0], skip=BN_FED_BIASES, enforce=False,
                  what=f"C3 step at B = 2048 against the reference's OWN codes ({nflip} of 524 288 differ)")
    # one Adam step at lr = 1e-4 moves every weight by at most lr; where the reference's own gradient is above its fp32
    # noise the step has the same direction
    sd_r = ref.state_dict()
    for k, v in m.state_dict().items():
        if k in BN_FED_BIASES or "tracked" in k:
            continue
        if "running" in k:
            assert (v.cpu() - sd_r[k]).abs().max().item() <= 1e-5 * max(1.0, sd_r[k].abs().max().item()), k
        else:
            assert (v.cpu() - sd_r[k]).abs().max().item() <= 2.0e-4 + 1e-9, k
    noisy = 0
    for k, p in ref.named_parameters():
        if not p.requires_grad or k in BN_FED_BIASES:
            continue
        clear = g32[k].abs() > 50 * (g32[k].double() - g64[k]).abs().max().float()   # well above the fp32 noise
        moved = (m.state_dict()[k].cpu() - sd_r[k]).abs()
        assert moved[clear].max().item() <= 2.5e-5 if clear.any() else True, k
        noisy += int((~clear).sum())
    print("entries whose reference gradient is inside its own fp32 noise:", noisy)


@pytest.mark.timeout(600)
def test_c2_latents_of_1024_patches_against_the_oracle_loop():
    import dynamorph_amd
    from dynamorph_amd.patch_vae import encode_patches
    from oracle import vqvae_oracle as O
    N = 1024
    torch.manual_seed(1024)
    ref = O.OracleVQVAE()
    m = dynamorph_amd.VQ_VAE().to(DEV)
    m.load_state_dict(ref.state_dict())
    x = torch.randn(N, 2, 128, 128, generator=torch.Generator().manual_seed(4321))
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        zb_r, za_r = O.encode_per_sample(ref, x)          # patch_VAE.py:445-452: 1024 batch-of-one calls, train-mode BN
        zb_64, _ = O.encode_per_sample(ref64, x.double())  # the float64 yardstick of the latents
    zb, za = encode_patches(m, x, device=DEV, batch_size=1024)
    assert zb.shape == (N, 4096) and za.shape == (N, 4096) and zb.dtype == np.float32
    # the latents on the float64 yardstick: as close to float64 as the reference's own fp32 latents are (x 1.5) -- until
    # round 5 a flat 3e-4 against the fp32 reference, ~300 x the error either of them has
    truth = zb_64.reshape(N, -1).numpy()
    e_ref = np.abs(zb_r.reshape(N, -1).numpy().astype(np.float64) - truth).max()
    e_hip = np.abs(zb.astype(np.float64) - truth).max()
    print(f"C2 latents against float64: HIP {e_hip:.2e}, fp32 reference {e_ref:.2e}")
    assert e_hip <= 1.5 * e_ref + 1e-7, (e_hip, e_ref)
    del ref64, zb_64, truth
    # quantised latents: equal wherever the code is the same; codes may differ only at the reference's own near-ties
    diff = (np.abs(za.reshape(N, 16, 16, 16) - za_r.numpy()) > 1e-3).any(axis=1)
    codes_gate(diff, zb_r, ref.vq.w.weight.detach(), "C2, N = 1024: all 262 144 positions")
    same = ~diff
    assert np.abs(za.reshape(N, 16, 16, 16) - za_r.numpy()).transpose(0, 2, 3, 1)[same].max() <= 1e-5
    # the side effect the loop has on the checkpointed buffers: 1024 momentum updates of the running statistics
    sd_r = ref.state_dict()
    for k, v in m.state_dict().items():
        if "running" in k:
            assert (v.cpu() - sd_r[k]).abs().max().item() <= 2e-4 * max(1.0, sd_r[k].abs().max().item()), k
        if "tracked" in k:
            assert int(v) == int(sd_r[k]) == N, k


@pytest.mark.timeout(900)
def test_c5_at_its_bench_batch_1024_self_consistency():
    """BASELINE.json configs[4] at the batch bench.py times it with (B = 1024: 1 M latent positions against 4096 codes) --
    too large for the CPU oracle (268 MB of distances per patch), so the size-independent properties:
      * the bf16-filter kernel's codes are bit-equal to the EXACT kernel's on the model's own latents (the reference
        initialises the codes uniformly in +-1/K, vq_vae.py:48: all 4096 sit in a cube of side 5e-4 -- the clustered case);
      * the code histogram sums to the number of positions; quantised = z + (q - z) of the chosen codes;
      * commitment loss = (1 + beta) * mse(q, z)   (vq_vae.py:74-76);
      * one FusedTrainer.step is finite and its captured-graph form equals the eager launch sequence bit for bit."""
    import dynamorph_amd
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    from dynamorph_amd._lib import DM_VQ_BF16, DM_VQ_EXACT
    from dynamorph_amd.train import FusedTrainer
    B, K = 1024, 4096
    torch.manual_seed(5)
    m = dynamorph_amd.VQ_VAE(num_inputs=4, num_embeddings=K, channel_var=np.ones(4)).to(DEV)
    x = torch.randn(B, 4, 256, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(77))
    with torch.no_grad():
        z, _ = E.encoder_forward(E.Layers(copy.deepcopy(m)), x)
    cb = m.vq.w.weight.detach()
    P = B * 32 * 32
    assert z.shape == (B, 16, 32, 32)
    idx_e, out_e, slabs_e, hist_e = ops.vq_forward(z, cb, variant=DM_VQ_EXACT)
    idx_b, out_b, slabs_b, hist_b, nre = ops.vq_forward(z, cb, variant=DM_VQ_BF16, want_rechecked=True)
    assert torch.equal(idx_b, idx_e) and torch.equal(out_b.view(torch.int32), out_e.view(torch.int32))
    assert torch.equal(hist_b, hist_e) and int(hist_b.sum()) == P
    assert 0 <= int(nre) <= P
    print(f"C5, B = 1024: {int(nre)} of {P} positions took the exact path ({100.0 * int(nre) / P:.2f} %)")
    q = ops.vq_decode(idx_b, cb)
    assert torch.equal(out_b, z + (q - z))
    mse = float(((q.double() - z.double()) ** 2).mean())
    sse = float(slabs_b.sum())
    assert abs(sse / (P * 16) - mse) <= 1e-6 * mse
    scal = ops.vq_finalize(slabs_b, hist_b, P, 16, 0.25).tolist()          # (loss, perplexity, mse)
    assert abs(scal[0] - 1.25 * mse) <= 2e-6 * mse and abs(scal[2] - mse) <= 2e-6 * mse
    p = hist_b.double() / P
    perp = float(torch.exp(-(p * torch.log(p + 1e-10)).sum()))
    assert abs(scal[1] - perp) <= 1e-4 * perp
    # 65 536 positions (64 random patches) of the model's own -- clustered -- latents against the C oracle (oracle/vq_oracle.c:
    # the reference's distance arithmetic and first-minimum rule), bit for bit; the whole batch would cost the CPU 4 . 10^9 distances
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    cvq = ctypes.CDLL(os.path.join(root, "oracle", "libvq_oracle.so"))
    pick = torch.randperm(B, generator=torch.Generator().manual_seed(11))[:64].sort().values
    z_sub = np.ascontiguousarray(z[pick.to(DEV)].cpu().numpy())
    cb_h = np.ascontiguousarray(cb.cpu().numpy())
    idx_ref = np.empty((64, 32, 32), np.int64)
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cvq.oracle_vq_encode(as_p(z_sub), as_p(cb_h), as_p(idx_ref), 64, 16, K, 32, 32)
    assert np.array_equal(idx_b[pick.to(DEV)].cpu().numpy(), idx_ref)
    del idx_e, out_e, out_b, q, z
    gc.collect()

    got = {}
    for graph in (True, False):
        mm = copy.deepcopy(m)
        tr = FusedTrainer(mm, lr=1e-3, use_graph=graph)
        vals = tr.step(x).tolist()
        assert all(np.isfinite(v) for v in vals), vals
        got[graph] = (vals, {k: v.detach().clone() for k, v in mm.state_dict().items()})
        del tr, mm
        gc.collect()
    assert got[True][0] == got[False][0]
    before = m.state_dict()
    moved = 0
    for k, a in got[True][1].items():
        b = got[False][1][k]
        if k == "vq.w.weight":
            # codebooks above 64 codes: the gradient's LDS adds run in arrival order inside a workgroup,
            # its low bits may differ between two runs; Adam's first step turns that into +-lr on entries whose gradient is
            # at rounding level, so compare the step itself
            assert float((a - b).abs().max()) <= 2.1e-3
            assert float(((a - before[k]) - (b - before[k])).abs().mean()) <= 1e-4
        else:
            assert torch.equal(a, b), k                               # everything else: bit for bit
        moved += int(not torch.equal(a, before[k]))
    assert moved > 40
