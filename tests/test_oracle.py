"""Pin the CPU oracle (oracle/) against the golden vectors captured from the reference.

These are `not gpu` tests: they prove the checker is the reference computation before any
HIP result is compared with it.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import vqvae_oracle as O


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def cvq():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libvq_oracle.so"))
    lib.oracle_vq_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5
    lib.oracle_vq_backward.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5
    return lib


def fresh(golden, **kw):
    m = O.OracleVQVAE(**kw)
    return O.load_numpy_state(m, golden("g1_state_dict.npz"))


# ----------------------------------------------------------------------- C oracle
def test_c_distances_bit_equal(cvq, golden):
    for name, D, K in (("g4_vq_indices.npz", 16, 64), ("g9_vq_d64.npz", 64, 512)):
        g = golden(name)
        z = golden("g5_vq_forward.npz")["z_before"] if D == 16 else g["z"]
        z0 = np.ascontiguousarray(z[0])
        cb = np.ascontiguousarray(g["codebook"])
        dist = np.empty((K, 16, 16), np.float32)
        cvq.oracle_vq_distances(_p(z0), _p(cb), _p(dist), D, K, 16, 16)
        assert np.array_equal(dist.view(np.uint32), g["dist_sample0"].view(np.uint32))


@pytest.mark.parametrize("name,zkey", [("g4_vq_indices.npz", None), ("g9_vq_d64.npz", "z"),
                                       ("g9_vq_k4096.npz", "z"), ("g9_vq_ties.npz", "z")])
def test_c_indices_bit_equal(cvq, golden, name, zkey):
    g = golden(name)
    z = np.ascontiguousarray(golden("g5_vq_forward.npz")["z_before"] if zkey is None else g[zkey])
    cb = np.ascontiguousarray(g["codebook"])
    B, D, H, W = z.shape
    idx = np.empty((B, H, W), np.int64)
    cvq.oracle_vq_encode(_p(z), _p(cb), _p(idx), B, D, cb.shape[0], H, W)
    assert np.array_equal(idx, g["idx"])


def test_c_forward_and_backward(cvq, golden):
    g5, g4, g6 = golden("g5_vq_forward.npz"), golden("g4_vq_indices.npz"), golden("g6_vq_backward.npz")
    z = np.ascontiguousarray(g5["z_before"])
    cb = np.ascontiguousarray(g4["codebook"])
    B, D, H, W = z.shape
    K = cb.shape[0]
    idx = np.empty((B, H, W), np.int64)
    out = np.empty_like(z)
    loss, perp = np.zeros(1, np.float32), np.zeros(1, np.float32)
    hist = np.zeros(K, np.int64)
    cvq.oracle_vq_forward(_p(z), _p(cb), 0.25, _p(idx), _p(out), _p(loss), _p(perp), _p(hist), B, D, K, H, W)
    assert np.array_equal(idx, g4["idx"])
    assert np.array_equal(out.view(np.uint32), g5["quantized"].view(np.uint32))   # z + (q - z), bit for bit
    assert abs(loss[0] - g5["loss"]) <= 1e-6 * abs(g5["loss"])
    assert abs(perp[0] - g5["perplexity"]) <= 1e-5 * abs(g5["perplexity"])
    assert hist.sum() == B * H * W

    dz, dw = np.empty_like(z), np.empty_like(cb)
    g_out = np.ascontiguousarray(g6["g_out"])
    cvq.oracle_vq_backward(_p(z), _p(cb), _p(idx), _p(g_out), float(g6["g_loss"]), 0.25, _p(dz), _p(dw), B, D, K, H, W)
    np.testing.assert_allclose(dz, g6["dz"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(dw, g6["dw"], rtol=2e-5, atol=1e-8)


# ------------------------------------------------------------------- torch oracle
def test_state_dict_contract(golden):
    g1 = golden("g1_state_dict.npz")
    m = O.OracleVQVAE()
    sd = m.state_dict()
    assert list(sd.keys()) == list(g1.keys()) and len(sd) == 68
    for k, v in sd.items():
        assert tuple(v.shape) == g1[k].shape, k
    assert sum(p.numel() for p in m.parameters()) == 24060


def test_encoder_batch_and_per_sample(golden):
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    g3 = golden("g3_encoder.npz")
    m = fresh(golden)
    z = m.enc(x)
    assert torch.equal(z, torch.from_numpy(g3["z_before"]))
    for k, v in m.state_dict().items():
        if "running" in k or "tracked" in k:
            assert np.array_equal(v.numpy(), g3["rs_batch/" + k]), k
    m = fresh(golden)
    zb, za = O.encode_per_sample(m, x)
    assert torch.equal(zb, torch.from_numpy(g3["z_before_per_sample"]))
    assert torch.equal(za, torch.from_numpy(g3["z_after_per_sample"]))
    for k, v in m.state_dict().items():
        if "running" in k or "tracked" in k:
            assert np.array_equal(v.numpy(), g3["rs_ps/" + k]), k


def test_vq_module(golden):
    g5, g4 = golden("g5_vq_forward.npz"), golden("g4_vq_indices.npz")
    m = fresh(golden)
    z = torch.from_numpy(g5["z_before"])
    assert np.array_equal(m.vq.encode_inputs(z).numpy(), g4["idx"])
    q, loss, perp = m.vq(z)
    assert np.array_equal(q.detach().numpy(), g5["quantized"])
    assert float(loss) == float(g5["loss"]) and float(perp) == float(g5["perplexity"])
    assert torch.equal(m.vq.decode_inputs(torch.from_numpy(g4["idx"])),
                       m.vq.w(torch.from_numpy(g4["idx"])).permute(0, 3, 1, 2))
    # known-answer identities
    zq = m.vq.decode_inputs(torch.from_numpy(g4["idx"]))
    mse = torch.mean((zq - z) ** 2)
    assert abs(float(loss) - 1.25 * float(mse)) < 1e-6
    p = np.bincount(g4["idx"].ravel(), minlength=64) / g4["idx"].size
    assert abs(float(perp) - np.exp(-(p * np.log(p + 1e-10)).sum())) < 1e-4


@pytest.mark.parametrize("masked", [False, True])
def test_forward(golden, masked):
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    g = golden("g5_forward_masked.npz" if masked else "g5_forward.npz")
    m = fresh(golden)
    dec, ld = m(x, batch_mask=torch.from_numpy(g["mask"]) if masked else None)
    assert np.array_equal(dec.detach().numpy(), g["decoded"])
    assert list(ld.keys()) == ["recon_loss", "commitment_loss", "time_matching_loss", "total_loss", "perplexity"]
    for k in ("recon_loss", "commitment_loss", "total_loss", "perplexity"):
        assert abs(float(ld[k]) - float(g[k])) <= 1e-6 * abs(float(g[k])), k
    assert ld["time_matching_loss"] == 0.


def test_gradients_and_adam(golden):
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    g6, g7 = golden("g6_grads.npz"), golden("g7_adam.npz")
    m = fresh(golden)
    _, ld = m(x)
    ld["total_loss"].backward()
    for k, p in m.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), g6["grad/" + k], rtol=1e-5, atol=1e-7, err_msg=k)
    assert m.channel_var.grad is None

    m = fresh(golden)
    opt = O.make_adam(m, 1e-4)
    m.zero_grad()
    for step in range(3):
        ld = O.train_step(m, opt, x)
        got = [float(ld[k]) for k in ("recon_loss", "commitment_loss", "total_loss", "perplexity")]
        np.testing.assert_allclose(got, g7["losses"][step], rtol=2e-6)
        if step in (0, 2):
            for k, v in m.state_dict().items():
                np.testing.assert_allclose(v.numpy(), g7[f"step{step + 1}/{k}"], rtol=1e-5, atol=2e-7, err_msg=k)


def test_time_matching_variants(golden):
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    for name, variant in (("g8_vqvae_time_matching.npz", "vq_vae"), ("g8_z16_time_matching.npz", "z16")):
        g = golden(name)
        m = fresh(golden, variant=variant)
        dec, ld = m(x, time_matching_mat=torch.from_numpy(g["tm"]))
        assert np.array_equal(dec.detach().numpy(), g["decoded"])
        for k in ("recon_loss", "commitment_loss", "time_matching_loss", "total_loss", "perplexity"):
            assert abs(float(ld[k]) - float(g[k])) <= 2e-6 * max(1.0, abs(float(g[k]))), (name, k)
        ld["total_loss"].backward()
        for k, p in m.named_parameters():
            if p.grad is not None:
                np.testing.assert_allclose(p.grad.numpy(), g["grad/" + k], rtol=2e-5, atol=2e-7, err_msg=k)
    assert list(ld.keys())[-2:] == ["perplexity", "total_loss"]       # vae.py:337-342 key order


def test_z32(golden):
    g = golden("g8_z32.npz")
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    m = O.load_numpy_state(O.OracleVQVAEz32(), g, prefix="sd/")
    assert torch.equal(m.enc(x), torch.from_numpy(g["z_before"]))
    m = O.load_numpy_state(O.OracleVQVAEz32(), g, prefix="sd/")
    dec, ld = m(x)
    assert np.array_equal(dec.detach().numpy(), g["decoded"])
    for k in ("recon_loss", "commitment_loss", "total_loss", "perplexity"):
        assert abs(float(ld[k]) - float(g[k])) <= 1e-6 * abs(float(g[k])), k


def test_z32_time_matching_mask_and_gradients(golden):
    """VQ_VAE_z32 with a time-matching matrix and a batch mask (vae.py:441-455): the oracle against the vectors captured
    from the reference -- losses, reconstruction and every parameter gradient (tests/golden/make_golden_z32_tm.py)."""
    g = golden("g8_z32_tm.npz")
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    m = O.load_numpy_state(O.OracleVQVAEz32(), g, prefix="sd/")
    dec, ld = m(x, time_matching_mat=torch.from_numpy(g["tm"]), batch_mask=torch.from_numpy(g["mask"]))
    assert np.array_equal(dec.detach().numpy(), g["decoded"])
    for k in ("recon_loss", "commitment_loss", "time_matching_loss", "total_loss", "perplexity"):
        assert abs(float(ld[k]) - float(g[k])) <= 2e-6 * max(1.0, abs(float(g[k]))), k
    ld["total_loss"].backward()
    n = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), g["grad/" + k], rtol=2e-5, atol=2e-7, err_msg=k)
            n += 1
    assert n == sum(1 for k in g if k.startswith("grad/"))


def test_z32_extra_loss(golden):
    """VQ_VAE_z32(extra_loss={name: fn}) (vae.py:463-469): the oracle against the vectors captured from the reference's own
    class with the two losses of tests/helpers/extra_losses.py, labels, a time-matching matrix and the caller-assigned alpha
    (tests/golden/make_golden_z32_extra.py) -- the loss dict's keys in the reference's order, every loss, every gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    from extra_losses import EXTRA
    g = golden("g8_z32_extra.npz")
    x = torch.from_numpy(golden("g2_input.npz")["x"])
    m = O.load_numpy_state(O.OracleVQVAEz32(extra_loss=dict(EXTRA)), g, prefix="sd/")
    m.alpha = float(g["alpha"])
    dec, ld = m(x, labels=torch.from_numpy(g["labels"]), time_matching_mat=torch.from_numpy(g["tm"]))
    assert list(ld.keys()) == [str(k) for k in g["loss_keys"]]
    assert np.array_equal(dec.detach().numpy(), g["decoded"])
    for k in ld:
        assert abs(float(ld[k]) - float(g["loss/" + k])) <= 2e-6 * max(1.0, abs(float(g["loss/" + k]))), k
    ld["total_loss"].backward()
    n = 0
    for k, p in m.named_parameters():
        if p.grad is not None:
            np.testing.assert_allclose(p.grad.numpy(), g["grad/" + k], rtol=2e-5, atol=2e-7, err_msg=k)
            n += 1
    assert n == sum(1 for k in g if k.startswith("grad/"))


def test_stress_codebooks(golden):
    for name, D, K in (("g9_vq_k4096.npz", 16, 4096), ("g9_vq_d64.npz", 64, 512)):
        g = golden(name)
        vq = O.OracleVQ(D, K)
        with torch.no_grad():
            vq.w.weight.copy_(torch.from_numpy(g["codebook"]))
        z = torch.from_numpy(g["z"])
        assert np.array_equal(vq.encode_inputs(z).numpy(), g["idx"])
        _, loss, perp = vq(z)
        assert float(loss) == float(g["loss"]) and float(perp) == float(g["perplexity"])


def test_zscore(golden):
    g = golden("g9_zscore.npz")
    p = np.squeeze(g["patches"])
    np.testing.assert_allclose(O.zscore_patch(p), g["zscore_patch"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.zscore(p), g["zscore"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.zscore(p, [40., 55.], [20., 30.]), g["zscore_given"], rtol=1e-12, atol=1e-12)


def test_chunked_distances_equal_the_full_tensor(golden):
    """OracleVQ.chunk (memory bound for the full-size parity tests): same indices, losses and gradients, bit for bit, as
    the reference's single (B, K, D, H, W) expression -- on the golden vectors of the reference itself and on a batch
    that does not divide by the chunk."""
    g5, g4 = golden("g5_vq_forward.npz"), golden("g4_vq_indices.npz")
    m = fresh(golden)
    m.vq.chunk = 3
    z = torch.from_numpy(g5["z_before"])
    assert np.array_equal(m.vq.encode_inputs(z).numpy(), g4["idx"])
    q, loss, perp = m.vq(z)
    assert np.array_equal(q.detach().numpy(), g5["quantized"]) and float(loss) == float(g5["loss"])
    for name, D, K in (("g9_vq_k4096.npz", 16, 4096), ("g9_vq_d64.npz", 64, 512), ("g9_vq_ties.npz", None, None)):
        g = golden(name)
        K, D = g["codebook"].shape
        vq = O.OracleVQ(D, K)
        with torch.no_grad():
            vq.w.weight.copy_(torch.from_numpy(g["codebook"]))
        vq.chunk = 1
        assert np.array_equal(vq.encode_inputs(torch.from_numpy(g["z"])).numpy(), g["idx"]), name
    torch.manual_seed(3)
    a, b = O.OracleVQVAE(), O.OracleVQVAE()
    b.load_state_dict(a.state_dict())
    b.vq.chunk = 2
    x = torch.randn(5, 2, 128, 128)
    la, lb = a(x)[1], b(x)[1]
    la["total_loss"].backward(); lb["total_loss"].backward()
    assert all(float(la[k]) == float(lb[k]) for k in la)
    for (k, p), (_, q_) in zip(a.named_parameters(), b.named_parameters()):
        assert p.grad is None and q_.grad is None or torch.equal(p.grad, q_.grad), k


def _relations_of(g, name):
    return {(int(a), int(b)): int(v) for (a, b), v in zip(g[f"{name}_pairs"], g[f"{name}_values"])}


def test_relation_steps_against_the_reference(golden):
    """concat_relations / reorder_with_trajectories (run_training.py:299-321, 97-160): the restatement gives the
    reference's order, matrix and generator position (g10: produced by executing the reference's own two functions)."""
    from oracle import relations_oracle as RO
    g = golden("g10_relations.npz")
    for name in "abcd":
        n, seed, rel = int(g[f"{name}_n"]), int(g[f"{name}_seed"]), _relations_of(g, name)
        order = RO.reorder_indices(n, rel, seed)
        after = np.random.randint(0, 2 ** 31, size=4)
        assert order == g[f"{name}_order"].tolist() and np.array_equal(after, g[f"{name}_after"]), name
        assert np.array_equal(np.asarray(RO.relation_matrix(n, rel, order).todense()), g[f"{name}_mat"]), name
    merged, labels = RO.concat_relations([_relations_of(g, "cc_r1"), _relations_of(g, "cc_r2")], [g["cc_l1"], g["cc_l2"]], [0, 30])
    assert list(merged.items()) == list(_relations_of(g, "cc_merged").items()) and np.array_equal(labels, g["cc_labels"])
