"""GPU: feeding the training step from HBM (csrc/feed.hip, dynamorph_amd/feed.py) and train()'s three feeds.

Reference semantics: run_training.py:504-532 (batch loop), :396-403 (augmentation), :335-374 (relation block, masks)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _aug_cases(golden):
    """The reference's own augmentation: batches that went through run_one_batch's loop (run_training.py:396-403) when
    tests/golden/make_golden_train_loop.py executed it -- input, output, numpy seed, generator position afterwards."""
    g = golden("g11_train_loop.npz")
    for i in range(int(g["aug_cases"])):
        yield (int(g[f"aug{i}/seed"]), torch.from_numpy(g[f"aug{i}/x_f16"].astype(np.float32)),
               torch.from_numpy(g[f"aug{i}/y_f16"].astype(np.float32)), g[f"aug{i}/after"])
    a = {k[2:]: v for k, v in g.items() if k.startswith("a/")}          # run a's first training batch: 8 x 2 x 128 x 128
    n = int(a["step_len"][0])
    data = torch.from_numpy(g["data_f16"].astype(np.float32))
    yield ("run a", data, torch.from_numpy(a["first_train_x_f16"].astype(np.float32)), a["step_ids"][:n])


def test_gather_augment_equals_the_reference_loop(golden):
    """dm_gather_augment with the codes ops.augment_codes draws from the reference's seed == what the reference's loop made
    of the same batch, bit for bit; all 12 (flip, rotation) pairs occur over the cases; C in {1, 2, 3, 4}, H in {20 .. 128}."""
    from dynamorph_amd import ops
    d = lambda t: torch.as_tensor(t).to(DEV, torch.int32)
    seen = set()
    for seed, x, want, extra in _aug_cases(golden):
        if seed == "run a":
            # inside train(): the split start is drawn first (run_training.py:490), then the batch's codes; rows = data[ids]
            np.random.seed(5)
            np.random.randint(0, 24 - 6)
            ids, B = torch.from_numpy(extra), len(extra)
        else:
            np.random.seed(seed)
            ids, B = torch.arange(len(x)), len(x)
        flips, rots = ops.augment_codes(B)
        if seed != "run a":
            assert np.array_equal(np.random.randint(0, 2 ** 31, size=2), extra), "generator position after the draws"
        seen |= set(zip(flips.tolist(), rots.tolist()))
        N, C, H = x.shape[0], x.shape[1], x.shape[2]
        out = torch.full((B + 2, C, H, H), 7.0, device=DEV)
        ops.gather_augment(x.to(DEV), d(ids), d(flips), d(rots), out, B)
        assert torch.equal(out[:B].cpu(), want), (seed, tuple(x.shape))
        assert bool((out[B:] == 7.0).all())                              # rows past n are not touched
        # no codes: a pure gather; no ids: the first B samples in order
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(N))[:B]
        ops.gather_augment(x.to(DEV), d(perm), None, None, out, B)
        assert torch.equal(out[:B].cpu(), x[perm])
        if seed != "run a":
            ops.gather_augment(x.to(DEV), None, d(flips), d(rots), out, B)
            assert torch.equal(out[:B].cpu(), want)
        # an id outside the dataset reads as zeros, never as a stray address
        bad = d(perm).clone()
        bad[0] = N + 5
        ops.gather_augment(x.to(DEV), bad, None, None, out, B)
        assert bool((out[0] == 0).all()) and torch.equal(out[1:B].cpu(), x[perm][1:])
        with pytest.raises(ValueError):
            ops.gather_augment(x.to(DEV), d(perm), None, None, torch.empty(B - 1, C, H, H, device=DEV), B)
    assert len(seen) == 12, sorted(seen)


def test_gather_rows_and_csr_block():
    import scipy.sparse as sp
    from dynamorph_amd import ops
    from dynamorph_amd.feed import _csr_arrays
    g = torch.Generator().manual_seed(5)
    N, B = 300, 70
    planes = torch.randn(N, 1, 32, 32, generator=g)
    ids = torch.randperm(N, generator=g)[:B]
    out = torch.empty(B, 1, 32, 32, device=DEV)
    ops.gather_rows(planes.to(DEV), ids.to(DEV, torch.int32), out)
    assert torch.equal(out.cpu(), planes[ids])
    rng = np.random.RandomState(3)
    dense = np.zeros((N, N), np.float64)
    for _ in range(4000):                                           # 1 = adjacent frames, 2 = same trajectory (run_training.py)
        i, j = rng.randint(0, N, 2)
        dense[i, j] = dense[j, i] = rng.choice([1.0, 2.0])
    for mat in (sp.csr_matrix(dense), sp.coo_matrix(dense), dense):
        indptr, indices, data, n = _csr_arrays(mat)
        csr = (indptr.to(DEV), indices.to(DEV), data.to(DEV), n)
        pos = torch.zeros(N, dtype=torch.int64, device=DEV)
        blk = torch.full((B, B), -3.0, device=DEV)
        for stamp in (1, 2, 3):                                     # the table is reused: stale entries must not leak
            ids = torch.randperm(N, generator=g)[:B]
            ops.csr_block(*csr, ids.to(DEV, torch.int32), pos, stamp, blk)
            want = np.asarray(sp.csr_matrix(dense)[ids.tolist(), :][:, ids.tolist()].todense(), dtype=np.float32)
            assert np.array_equal(blk.cpu().numpy(), want)
    empty = sp.csr_matrix((N, N))
    indptr, indices, data, n = _csr_arrays(empty)
    blk = torch.full((B, B), -3.0, device=DEV)
    ops.csr_block(indptr.to(DEV), indices.to(DEV), data.to(DEV), n, ids.to(DEV, torch.int32), pos, 9, blk)
    assert bool((blk == 0).all())


def _dataset(n, seed, masks=False, relation=False):
    import scipy.sparse as sp
    g = torch.Generator().manual_seed(seed)
    data = torch.utils.data.TensorDataset(torch.randn(n, 2, 128, 128, generator=g))
    mask = rel = None
    if masks:
        mask = torch.utils.data.TensorDataset((torch.rand(n, 2, 128, 128, generator=g) > 0.3).float() * 2 - 1)
    if relation:
        rng = np.random.RandomState(seed)
        dense = np.zeros((n, n))
        for i in range(n - 1):
            dense[i, i + 1] = dense[i + 1, i] = 2.0 if rng.rand() < 0.5 else 1.0
        np.fill_diagonal(dense, 2.0)
        rel = sp.csr_matrix(dense)
    return data, mask, rel


@pytest.mark.parametrize("masks,relation,transform", [(False, False, True), (True, True, True), (True, False, None)])
def test_train_feeds_agree_bit_for_bit(tmp_path, masks, relation, transform):
    """train(feed='resident'), feed='stream' and feed='sync' (the reference's loop as it is) from one seed: the same
    batches, augmentation draws and steps, hence bit-identical parameters, buffers and epoch losses -- ragged last
    batches of both phases included."""
    import dynamorph_amd
    from dynamorph_amd.train import train
    data, mask, rel = _dataset(53, 17, masks, relation)
    torch.manual_seed(2)
    m0 = dynamorph_amd.VQ_VAE().to(DEV)
    got = {}
    for feed in ("sync", "resident", "stream"):
        m = copy.deepcopy(m0)
        np.random.seed(123)

        class W:
            rows = {}

            def add_scalar(self, key, value, epoch):
                self.rows.setdefault(key, []).append(float(value))
        w = W()
        w.rows = {}
        st = {}
        train(m, data, str(tmp_path / feed), relation_mat=rel, mask=mask, n_epochs=3, lr=1e-3, batch_size=16, device=DEV,
              shuffle_data=not relation, transform=transform, val_split_ratio=0.3, patience=10, writer=w, feed=feed, stats=st)
        assert st["feed"] == feed and st["phase_samples"] == {"train": 38, "val": 15}
        got[feed] = ({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, w.rows, np.random.randint(0, 1 << 30))
    for feed in ("resident", "stream"):
        for k, v in got["sync"][0].items():
            assert torch.equal(v, got[feed][0][k]), (feed, k)
        assert got[feed][2] == got["sync"][2], "numpy's generator was left in a different state"
        for key, vals in got["sync"][1].items():
            np.testing.assert_allclose(got[feed][1][key], vals, rtol=2e-6, atol=1e-7, err_msg=f"{feed} {key}")


def test_train_feed_with_a_torch_optimizer_and_z32(tmp_path):
    """The device feeds also serve the autograd path (fused=False: model(x) / backward / torch Adam) and VQ_VAE_z32."""
    import dynamorph_amd
    from dynamorph_amd.train import train
    data, mask, rel = _dataset(20, 4, True, True)
    torch.manual_seed(3)
    for cls, fused in ((dynamorph_amd.VQ_VAE, False), (dynamorph_amd.VQ_VAE_z32, True)):
        m0 = cls().to(DEV)
        res = {}
        for feed in ("sync", "resident"):
            m = copy.deepcopy(m0)
            np.random.seed(7)
            train(m, data, str(tmp_path / f"{cls.__name__}_{feed}"), relation_mat=rel, mask=mask, n_epochs=2, batch_size=8,
                  device=DEV, transform=True, val_split_ratio=0.25, patience=5, fused=fused, feed=feed)
            res[feed] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        for k, v in res["sync"].items():
            assert torch.equal(v, res["resident"][k]), (cls.__name__, k)


def test_fused_evaluate_equals_the_module_forward():
    """FusedTrainer.evaluate (the validation pass as a captured forward-only graph) returns the loss values of
    model(x, ...) and advances the BatchNorm running statistics exactly like it."""
    import dynamorph_amd
    from dynamorph_amd.train import FusedTrainer
    torch.manual_seed(8)
    for cls in (dynamorph_amd.VQ_VAE, dynamorph_amd.VQ_VAE_z16, dynamorph_amd.VQ_VAE_z32):
        m = cls().to(DEV)
        ref = copy.deepcopy(m)
        x = torch.randn(6, 2, 128, 128, device=DEV)
        mask = (torch.rand(6, 1, 128, 128, device=DEV) > 0.4).float()
        tm = torch.randint(0, 3, (6, 6), device=DEV).float()
        tm = torch.maximum(tm, tm.t())
        tr = FusedTrainer(m)
        for kw in (dict(), dict(batch_mask=mask), dict(batch_mask=mask, time_matching_mat=tm)):
            with torch.no_grad():
                _, ld = ref(x, **kw)
            vals = tr.evaluate(x, kw.get("batch_mask"), kw.get("time_matching_mat")).tolist()
            want = [float(ld[k]) for k in ("recon_loss", "commitment_loss", "total_loss", "perplexity")]
            if "time_matching_mat" in kw:
                want.append(float(ld["time_matching_loss"]))
            np.testing.assert_allclose(vals, want, rtol=2e-6, atol=1e-7, err_msg=f"{cls.__name__} {sorted(kw)}")
            vals2 = tr.evaluate(x, kw.get("batch_mask"), kw.get("time_matching_mat")).tolist()      # the replayed graph
            with torch.no_grad():
                ref(x, **kw)
            np.testing.assert_allclose(vals2, want, rtol=2e-6, atol=1e-7)
        for (k, a), (_, b) in zip(m.named_buffers(), ref.named_buffers()):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (cls.__name__, k)
        for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
            assert torch.equal(a, b), k                                                                # nothing was trained
