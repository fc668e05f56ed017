"""`not gpu`: host-side logic of the drop-in surface (construction, state-dict contract, helpers, no compute)."""
import os
import pickle

import numpy as np
import pytest
import torch

import dynamorph_amd
from dynamorph_amd import dist as D
from dynamorph_amd import train_utils as TU


def test_state_dict_contract_matches_reference(golden):
    g1 = golden("g1_state_dict.npz")
    for cls in (dynamorph_amd.VQ_VAE, dynamorph_amd.VQ_VAE_z16):
        m = cls()
        sd = m.state_dict()
        assert list(sd.keys()) == list(g1.keys()) and len(sd) == 68
        for k, v in sd.items():
            assert tuple(v.shape) == g1[k].shape and str(v.dtype).replace("torch.", "") == str(g1[k].dtype), k
        m.load_state_dict({k: torch.from_numpy(v) for k, v in g1.items()})
        assert sum(p.numel() for p in m.parameters()) == 24060
        assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 24058
        assert not m.channel_var.requires_grad


def test_constructor_surface():
    m = dynamorph_amd.VQ_VAE(num_inputs=2, num_hiddens=16, num_residual_hiddens=32, num_residual_layers=2,
                             num_embeddings=64, commitment_cost=0.25, channel_var=np.array([1., 1.]),
                             weight_recon=1., weight_commitment=1., weight_matching=0.005, device="cuda:0")
    assert (m.num_inputs, m.num_hiddens, m.num_embeddings) == (2, 16, 64)
    assert isinstance(m.enc, torch.nn.Sequential) and isinstance(m.dec, torch.nn.Sequential)
    assert isinstance(m.enc[12], dynamorph_amd.ResidualBlock) and len(m.enc) == 13 and len(m.dec) == 7
    assert isinstance(m.vq, dynamorph_amd.VectorQuantizer) and m.vq.embeddings is m.vq.w.weight
    # callers pass gpu=True (pipeline/patch_VAE.py:431) and alpha=... (plot_scripts/recon_loss.py:18)
    dynamorph_amd.VQ_VAE_z16(num_inputs=2, num_hiddens=16, num_residual_hiddens=32, num_residual_layers=2,
                             num_embeddings=64, gpu=True)
    dynamorph_amd.VQ_VAE(alpha=0.0005, gpu=True)
    z16 = dynamorph_amd.VQ_VAE_z16(w_a=1.2, w_t=0.2, w_n=-0.4, margin=0.3)
    assert (z16.w_a, z16.w_t, z16.w_n, z16.margin) == (1.2, 0.2, -0.4, 0.3)
    vq = dynamorph_amd.VectorQuantizer(embedding_dim=16, num_embeddings=8, commitment_cost=0.1, device="cpu")
    assert vq.w.weight.shape == (8, 16) and vq.commitment_cost == 0.1
    rb = dynamorph_amd.ResidualBlock(16, 32, 3)
    assert len(rb.layers) == 3 and [type(l).__name__ for l in rb.layers[0]] == ["ReLU", "Conv2d", "BatchNorm2d", "ReLU", "Conv2d", "BatchNorm2d"]


def test_no_cpu_fallback():
    m = dynamorph_amd.VQ_VAE()
    x = torch.randn(1, 2, 128, 128)
    for call in (lambda: m(x), lambda: m.enc(x), lambda: m.vq(torch.randn(1, 16, 16, 16)),
                 lambda: m.dec(torch.randn(1, 16, 16, 16)), lambda: m.vq.encode_inputs(torch.randn(1, 16, 16, 16)),
                 lambda: m.enc[12](torch.randn(1, 16, 16, 16))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    # the child layers only hold parameters (state-dict layout of the reference): calling one is an error, not ATen
    for layer in (m.enc[0], m.enc[1], m.enc[2], m.dec[2], m.enc[12].layers[0][1]):
        with pytest.raises(RuntimeError, match="parameter container"):
            layer(torch.randn(1, 2, 8, 8))
    assert isinstance(m.enc[0], torch.nn.Conv2d) and type(m.enc[0]).__name__ == "Conv2d"
    from dynamorph_amd.train import FusedTrainer
    with pytest.raises(RuntimeError):
        FusedTrainer(m)


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.abspath(dynamorph_amd.__file__))
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt, os.path.join(dp, f)


def test_zscore_helpers(golden):
    g = golden("g9_zscore.npz")
    p = np.squeeze(g["patches"])
    np.testing.assert_allclose(TU.zscore_patch(p), g["zscore_patch"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(TU.zscore(p), g["zscore"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(TU.zscore(p, [40., 55.], [20., 30.]), g["zscore_given"], rtol=1e-12, atol=1e-12)
    flat = np.ones((2, 2, 4, 4))                       # zero variance: eps keeps it finite
    assert np.isfinite(TU.zscore_patch(flat)).all()


def test_early_stopping_checkpoints_state_dict(tmp_path):
    path = str(tmp_path / "model.pt")
    m = torch.nn.Linear(2, 2)
    es = TU.EarlyStopping(patience=2, verbose=False, path=path, trace_func=lambda *_: None)
    es(1.0, m)
    assert os.path.exists(path) and es.val_loss_min == 1.0
    saved = torch.load(path)
    assert set(saved) == {"weight", "bias"}
    es(1.5, m); assert es.counter == 1 and not es.early_stop
    es(0.5, m); assert es.counter == 0 and es.val_loss_min == 0.5
    es(0.6, m); es(0.7, m)
    assert es.early_stop


def test_shard_range_tiles_the_units():
    for n in (0, 1, 7, 64, 1000, 16384):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_params_are_views():
    m = dynamorph_amd.VQ_VAE()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = D.FlatParams(m.parameters())
    assert fp.flat.numel() == 24058 and len(fp.params) == 43
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k                      # values unchanged
    p0 = fp.params[0]
    assert p0.data_ptr() == fp.flat.data_ptr()
    fp.flat.zero_()
    assert float(m.enc[0].weight.abs().sum()) == 0.0            # the flat buffer IS the storage
    fp.gview(p0).fill_(2.0)
    assert float(fp.grad[:p0.numel()].sum()) == 2.0 * p0.numel()
    fp.expose_grads()
    assert m.enc[0].weight.grad.data_ptr() == fp.grad.data_ptr()
    assert m.channel_var.grad is None
    assert D.world_size() == 1 and D.allreduce_mean_(fp.grad) is fp.grad
    assert D.max_over_ranks(3.5) == 3.5


def test_process_vae_io_contract_signature():
    import inspect
    from dynamorph_amd import patch_vae
    sig = inspect.signature(patch_vae.process_VAE)
    assert list(sig.parameters)[:5] == ["raw_folder", "supp_folder", "sites", "config_", "gpu"]


def test_relation_tensor_and_mask_slicing():
    """run_training.py:335-374 restated in dynamorph_amd.train: batch block of the sparse relation matrix, large-mask
    channel mapped from {-1, 1} to {0, 1}."""
    import scipy.sparse as sp
    from torch.utils.data import TensorDataset
    from dynamorph_amd.train import get_mask, get_relation_tensor
    rel = sp.dok_matrix((10, 10), dtype=np.float32)
    rel[1, 2] = rel[2, 1] = 1.
    rel[2, 7] = rel[7, 2] = 2.
    ids = [7, 2, 1, 4]
    out = get_relation_tensor(rel, ids, device=None)
    assert out.dtype == torch.float32 and out.shape == (4, 4)
    expect = np.zeros((4, 4), np.float32)
    expect[0, 1] = expect[1, 0] = 2.
    expect[1, 2] = expect[2, 1] = 1.
    assert np.array_equal(out.numpy(), expect)
    assert get_relation_tensor(None, ids) is None and get_mask(None, ids) is None
    m = torch.sign(torch.randn(10, 2, 8, 8))
    bm = get_mask(TensorDataset(m), ids, device="cpu")
    assert bm.shape == (4, 1, 8, 8)
    assert torch.equal(bm, (m[ids][:, 1:2] + 1) / 2) and set(bm.unique().tolist()) <= {0., 1.}


class _FakeDeviceTensor:
    """What ops._ptr looks at, for a tensor that claims to live on device `index` (no GPU needed: nothing is launched)."""

    def __init__(self, index, dtype=torch.float32):
        import types
        self.is_cuda, self.dtype, self.device = True, dtype, types.SimpleNamespace(index=index, type="cuda")

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return 0x1000


def test_operands_on_different_devices_raise_and_leave_no_state_behind(monkeypatch):
    """The device guard (dynamorph_amd/_lib.py, ops._op): every pointer of ONE call is checked against the call's device,
    also across the host-only queries in between; an error while a launch is being assembled leaves nothing behind."""
    from dynamorph_amd import _lib, ops
    lib = _lib.load()
    assert _lib.call_device.index is None
    a0, b0, a1 = _FakeDeviceTensor(0), _FakeDeviceTensor(0), _FakeDeviceTensor(1)
    # dm_apply(inp on cuda:0, resid on cuda:1): raises before anything is launched
    with pytest.raises(ValueError, match="different devices"):
        ops.apply(ops.Op(a0), 1, 1, 4, 4, resid=a1, out=b0)
    assert _lib.call_device.index is None                       # cleared although the call died half-way
    # a host-only query between two pointers of a call does not forget the first pointer's device
    @ops._op
    def call():
        ops._ptr(a0)
        assert lib.dm_vq_num_blocks(4096) > 0 and lib.dm_conv3x3_num_blocks(4, 16, 16, 16, 16, 9, 0, 0) == 4
        assert _lib.call_device.index == 0
        ops._ptr(a1)
    with pytest.raises(ValueError, match=r"cuda:0 and cuda:1"):
        call()
    assert _lib.call_device.index is None
    # an unrelated error (dtype) after the first pointer: the next call starts clean, on another device
    with pytest.raises(ValueError, match="dtype"):
        ops.apply(ops.Op(a0), 1, 1, 4, 4, resid=_FakeDeviceTensor(0, torch.float64), out=b0)
    assert _lib.call_device.index is None
    with pytest.raises(ValueError, match="different devices"):
        ops.apply(ops.Op(a1), 1, 1, 4, 4, resid=a0, out=a1)      # (would be 'cuda:0 and ...' with a stale index)
    # the weights of a convolution are operands too
    w = _FakeDeviceTensor(1)
    wv = ops.WView.__new__(ops.WView)
    wv._keep, wv._scratch, wv.struct = w, None, _lib.WeightView(0x1000, 0, 1, 1, 1, 1, None, 0)
    @ops._op
    def conv_like():
        ops._ptr(a0)
        wv.ref()
    with pytest.raises(ValueError, match="different devices"):
        conv_like()
    assert _lib.is_host_only("dm_conv3x3_scratch_floats") and not _lib.is_host_only("dm_conv3x3")


def test_deepcopy_and_pickle_give_an_independent_module():
    """model.enc / model.dec hold no reference to their parent: a deep copy's halves work on the COPY's parameters (a weak
    back-reference used to survive copy.deepcopy pointing at the original: the copy's enc then ran -- and advanced the
    BatchNorm running statistics of -- the original), and the whole module pickles (torch.save(model))."""
    import copy
    import io
    from dynamorph_amd import engine as E
    for cls in (dynamorph_amd.VQ_VAE, dynamorph_amd.VQ_VAE_z16, dynamorph_amd.VQ_VAE_z32):
        m = cls()
        c = copy.deepcopy(m)
        assert not hasattr(m.enc, "_owner") and not hasattr(m.dec, "_owner")
        assert all(a.data_ptr() != b.data_ptr() for a, b in zip(m.parameters(), c.parameters()))
        if cls is not dynamorph_amd.VQ_VAE_z32:
            Lc, Lm = E.Layers(enc=c.enc), E.Layers(enc=m.enc)
            assert Lc.enc1.weight is c.enc[1].weight and Lc.bn1.running_mean is c.enc[2].running_mean
            assert Lc.enc1.weight is not Lm.enc1.weight
            Ld = E.Layers(dec=c.dec)
            assert Ld.dec4.weight is c.dec[4].weight and Ld.channel_var is None and (Ld.nh, Ld.nin) == (16, 2)
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        r = torch.load(buf, weights_only=False)
        assert list(r.state_dict().keys()) == list(m.state_dict().keys())


def test_augmentation_codes_follow_the_reference_draw_order(golden):
    """run_training.py:396-403 draws a flip and a rotation per sample from numpy's global generator; ops.augment_codes must
    hand out the reference's codes from the same seed and leave the generator where the loop leaves it.  The codes are read
    off the batches the REFERENCE's loop produced (g11_train_loop.npz aug cases): the (flip, rotation) pairs that map
    each input sample onto its output."""
    from dynamorph_amd import ops
    g = golden("g11_train_loop.npz")
    for i in range(int(g["aug_cases"])):
        x = torch.from_numpy(g[f"aug{i}/x_f16"].astype(np.float32))
        y = torch.from_numpy(g[f"aug{i}/y_f16"].astype(np.float32))
        np.random.seed(int(g[f"aug{i}/seed"]))
        flips, rots = ops.augment_codes(len(x))
        assert flips.dtype == np.int32 and rots.dtype == np.int32 and len(flips) == len(x)
        assert np.array_equal(np.random.randint(0, 2 ** 31, size=2), g[f"aug{i}/after"])
        for b in range(len(x)):
            fits = [(f, k) for f in (0, 1, 2) for k in (0, 1, 2, 3)
                    if torch.equal(torch.rot90(x[b] if f == 0 else torch.flip(x[b], dims=(f,)), k=k, dims=[1, 2]), y[b])]
            # (12 pairs, 8 distinct transforms: a flip + rotation can equal the other flip + another rotation)
            assert (int(flips[b]), int(rots[b])) in fits and len(fits) <= 2, (i, b, fits)
    for n in (0, 2048):                                    # nothing drawn for an empty batch; one vectorised draw for a large one
        np.random.seed(3)
        f, r = ops.augment_codes(n)
        assert len(f) == n and len(r) == n and (n == 0 or (set(f.tolist()) == {0, 1, 2} and set(r.tolist()) == {0, 1, 2, 3}))


def test_feed_host_helpers():
    """mask_plane = run_training.py:371-372 for all samples at once; _csr_arrays sums duplicates like todense()."""
    import scipy.sparse as sp
    from dynamorph_amd.feed import _csr_arrays, dataset_tensor, mask_plane
    m = (torch.rand(5, 2, 8, 8) > 0.5).float() * 2 - 1
    ids = [3, 0, 4]
    want = (torch.utils.data.TensorDataset(m)[ids][0][:, 1:2, :, :] + 1.) / 2.
    assert torch.equal(mask_plane(m)[ids], want)
    assert dataset_tensor(torch.utils.data.TensorDataset(m)) is m and dataset_tensor(object()) is None
    coo = sp.coo_matrix((np.array([1.0, 1.0, 2.0]), (np.array([0, 0, 2]), np.array([1, 1, 0]))), shape=(3, 3))
    indptr, indices, data, n = _csr_arrays(coo)
    dense = np.zeros((3, 3), np.float32)
    for r in range(3):
        for e in range(int(indptr[r]), int(indptr[r + 1])):
            dense[r, int(indices[e])] += float(data[e])
    assert n == 3 and np.array_equal(dense, np.asarray(coo.todense(), np.float32))


# ------------------------------------------------------------------ the steps between the pickles and train()
def _relations_of(g, name):
    return {(int(a), int(b)): int(v) for (a, b), v in zip(g[f"{name}_pairs"], g[f"{name}_values"])}


def _random_trajectories(rng, n, max_len, density=0.8):
    ids, rel, at = rng.permutation(n), {}, 0
    while at < n * density:
        ln = int(rng.integers(2, max_len + 1))
        tr = ids[at:at + ln]
        at += ln
        for i in range(len(tr)):
            for j in range(len(tr)):
                if i != j:
                    rel[(int(tr[i]), int(tr[j]))] = 2 if abs(i - j) == 1 else 1
    return rel


def test_reorder_with_trajectories_matches_the_reference(golden):
    """run_training.py:97-160 through dm_reorder_with_trajectories: the reference's order, reordered tensor, relation
    matrix and generator position (fixture made by executing the reference's function), in O(n log n)."""
    from torch.utils.data import TensorDataset
    from dynamorph_amd import relations as R
    g = golden("g10_relations.npz")
    for name in "abcd":
        n, seed, rel = int(g[f"{name}_n"]), int(g[f"{name}_seed"]), _relations_of(g, name)
        ds, mat, inds = R.reorder_with_trajectories(TensorDataset(torch.arange(n, dtype=torch.float32).reshape(n, 1)), rel, seed=seed)
        after = np.random.randint(0, 2 ** 31, size=4)
        assert inds == g[f"{name}_order"].tolist() and all(type(i) is int for i in inds), name
        assert np.array_equal(after, g[f"{name}_after"]), "numpy's generator is left where the reference leaves it"
        assert np.array_equal(ds.tensors[0].numpy(), g[f"{name}_data"])
        assert np.array_equal(np.asarray(mat.todense()), g[f"{name}_mat"]) and mat.shape == (n, n)
    merged, labels = R.concat_relations([_relations_of(g, "cc_r1"), _relations_of(g, "cc_r2")], [g["cc_l1"], g["cc_l2"]], [0, 30])
    assert list(merged.items()) == list(_relations_of(g, "cc_merged").items()) and np.array_equal(labels, g["cc_labels"])


@pytest.mark.parametrize("n,max_len,seed", [(1, 2, 0), (2, 2, 1), (63, 3, 2), (64, 9, 3), (65, 4, None), (1500, 12, 123), (4097, 5, 9)])
def test_reorder_order_against_the_literal_loop(n, max_len, seed):
    """Against the oracle's literal restatement (list(pool) + np.random.choice per pick), sizes around the powers of two
    of the rejection mask; seed None continues the current generator state as the reference does."""
    from dynamorph_amd import relations as R
    from oracle import relations_oracle as RO
    rel = _random_trajectories(np.random.default_rng(n), n, max_len) if n > 1 else {}
    np.random.seed(77)
    state = np.random.get_state()
    want = RO.reorder_indices(n, rel, seed)
    want_after = np.random.randint(0, 2 ** 31, size=3)
    np.random.set_state(state)
    got = R.trajectory_order(n, rel, seed)
    assert got.tolist() == want and np.array_equal(np.random.randint(0, 2 ** 31, size=3), want_after)
    assert sorted(got.tolist()) == list(range(n))


def test_reorder_keyerrors_of_the_reference():
    """One-directional pairs: the reference raises KeyError from relation_dict[elem] (a reached sample that starts no
    pair) or from inds_pool.remove (a trajectory reaching a sample that is gone); same exception, same sample."""
    from dynamorph_amd import relations as R
    from oracle import relations_oracle as RO
    rng = np.random.default_rng(5)
    random_directed = [{(int(a), int(b)): 2 for a, b in rng.integers(0, 4, (k, 2)) if a != b} for k in (2, 3, 4, 5, 6, 7)]
    for rel in [{(0, 1): 2}, {(0, 1): 2, (1, 0): 2, (2, 1): 2, (1, 2): 2, (3, 0): 2}] + random_directed:
        for seed in range(6):
            try:
                want = ("ok", RO.reorder_indices(4, rel, seed))
            except KeyError as e:
                want = ("KeyError", int(e.args[0]))
            try:
                got = ("ok", R.trajectory_order(4, rel, seed).tolist())
            except KeyError as e:
                got = ("KeyError", int(e.args[0]))
            assert got == want, (rel, seed)
