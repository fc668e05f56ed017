#!/usr/bin/env python3
"""Golden fixture for the training loop: tests/golden/g11_train_loop.npz.

Runs ONLY in the build container, where /root/reference is mounted.  run_training.py does not import here (h5py, cv2,
tensorboard are absent), so four of its functions are taken out of its syntax tree and executed on their own, the way
make_golden_relations.py does it:

    get_relation_tensor  run_training.py:335-355
    get_mask             run_training.py:358-374
    run_one_batch        run_training.py:377-417
    train                run_training.py:455-551

Their namespace: numpy, torch (as `t`), os, the reference's own EarlyStopping (pipeline/train_utils.py:8-60; numpy 2 has
no `np.Inf`, the alias is restored before it is constructed) and a recording stand-in for tensorboard's SummaryWriter.
The models are the reference's: HiddenStateExtractor.vae.VQ_VAE_z16 and HiddenStateExtractor.vq_vae.VQ_VAE on the CPU.
The reference source never travels; only the .npz written here does.

Three runs over ONE dataset of 24 patches (float16-representable N(0,1) values, stored as float16), learning rate 1e-4
(configs/config_example.yml:181):

    a  VQ_VAE_z16, transform=True, masks, relation matrix, shuffle_data=False, 2 epochs   (what run_training.main runs)
    b  VQ_VAE,     transform=True, no masks,               shuffle_data=True,  3 epochs
    c  VQ_VAE,     transform=None, masks,                  shuffle_data=True,  3 epochs   (reproducible without a GPU)

Per run: the sample ids of every dataset[...] call in order, the augmented batch the model saw at the first training
step in full plus a position-weighted checksum of every step's batch, the mask (first training / validation step in full,
checksums of all) and relation block (every step) the model was given, the five loss values of every step, every
writer.add_scalar row, the state dict written to model.pt, the final state dict and four draws from numpy's generator after train() returned.

Plus four small batches of other shapes through run_one_batch's augmentation loop alone (aug0..aug3: input, output, seed,
generator position) -- what tests/test_gpu_feed.py holds dm_gather_augment against.

    cd /tmp && python3 /root/repo/tests/golden/make_golden_train_loop.py
"""
import ast
import os
import sys
import tempfile
import types

REF = os.environ.get("DYNAMORPH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.sparse import csr_matrix  # noqa: E402
from torch.utils.data import TensorDataset  # noqa: E402

if not hasattr(np, "Inf"):
    np.Inf = np.inf                               # pipeline/train_utils.py:32 (numpy 1 spelling)
import HiddenStateExtractor.vae as ref_vae  # noqa: E402
import HiddenStateExtractor.vq_vae as ref_vq  # noqa: E402
from pipeline.train_utils import EarlyStopping  # noqa: E402

torch.set_num_threads(8)


class Recorder:
    """SummaryWriter stand-in: keeps (tag, value, epoch) rows in call order."""
    rows = []

    def __init__(self, *a, **k):
        pass

    def add_scalar(self, tag, value, step):
        Recorder.rows.append((tag, float(value), int(step)))

    def flush(self):
        pass

    def close(self):
        pass


tree = ast.parse(open(os.path.join(REF, "run_training.py")).read())
wanted = ("get_relation_tensor", "get_mask", "run_one_batch", "train")
mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
ns = {"np": np, "t": torch, "os": os, "SummaryWriter": Recorder, "EarlyStopping": EarlyStopping}
exec(compile(mod, "run_training.py (four functions)", "exec"), ns)


class RecordingDataset(TensorDataset):
    """dataset[ids] as the reference calls it, remembering the ids."""

    def __init__(self, *tensors):
        super().__init__(*tensors)
        self.calls = []

    def __getitem__(self, ids):
        self.calls.append(np.asarray(ids, dtype=np.int64).copy())
        return super().__getitem__(ids)


N, BATCH = 24, 8
g = torch.Generator().manual_seed(20261005)
data = torch.randn(N, 2, 128, 128, generator=g).half().float()
masks = torch.where(torch.rand(N, 2, 128, 128, generator=g) > 0.4, 1.0, -1.0)
# trajectories of 2-4 consecutive samples: adjacent frames 2, same trajectory 1 (both directions), as
# generate_trajectory_relations-style code fills the dict; samples 20-23 belong to none
rel = {}
for lo, ln in ((0, 3), (3, 4), (7, 2), (9, 4), (13, 3), (16, 4)):
    for i in range(ln):
        for j in range(ln):
            if i != j:
                rel[(lo + i, lo + j)] = 2 if abs(i - j) == 1 else 1
keys = np.array(list(rel.keys()))
relation_mat = csr_matrix((np.array(list(rel.values()), dtype=np.float64), (keys[:, 0], keys[:, 1])), shape=(N, N))

weights = torch.arange(1, 2 * 128 * 128 + 1, dtype=torch.float64).reshape(1, 2, 128, 128) / (2 * 128 * 128)


def checksum(x):
    """Per-sample position-weighted sum in double: any flip / rotation / wrong sample changes it."""
    return (x.double() * weights).sum(dim=(1, 2, 3)).numpy()


out = {"data_f16": data.half().numpy(), "masks_i8": masks.to(torch.int8).numpy(),
       "relation_dense_i8": np.asarray(relation_mat.todense()).astype(np.int8), "batch_size": np.int64(BATCH)}


def run(tag, cls, seed, n_epochs, lr, use_mask, use_rel, shuffle, transform, val_split_ratio):
    torch.manual_seed(0)
    model = cls(device="cpu")
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ds = RecordingDataset(data.clone())
    mk = TensorDataset(masks.clone()) if use_mask else None
    steps = []

    def pre(mod_, args, kwargs):
        steps.append({"x": args[0].detach().clone(), "training": None,
                      "tm": None if kwargs.get("time_matching_mat") is None else kwargs["time_matching_mat"].clone(),
                      "mask": None if kwargs.get("batch_mask") is None else kwargs["batch_mask"].clone()})

    def post(mod_, args, kwargs, result):
        steps[-1]["losses"] = [float(v) for v in result[1].values()]
        steps[-1]["keys"] = list(result[1].keys())

    h1 = model.register_forward_pre_hook(pre, with_kwargs=True)
    h2 = model.register_forward_hook(post, with_kwargs=True)
    Recorder.rows = []
    with tempfile.TemporaryDirectory() as tmp:
        np.random.seed(seed)
        ns["train"](model, ds, tmp, relation_mat=relation_mat if use_rel else None, mask=mk, n_epochs=n_epochs, lr=lr,
                    batch_size=BATCH, device="cpu", shuffle_data=shuffle, transform=transform,
                    val_split_ratio=val_split_ratio, patience=20)
        after = np.random.randint(0, 2 ** 31, size=4)
        ckpt = torch.load(os.path.join(tmp, "model.pt"))
    h1.remove()
    h2.remove()
    assert len(steps) == len(ds.calls)
    o = {"seed": np.int64(seed), "n_epochs": np.int64(n_epochs), "lr": np.float64(lr), "shuffle": np.int64(shuffle),
         "transform": np.int64(transform is not None), "val_split_ratio": np.float64(val_split_ratio),
         "use_mask": np.int64(use_mask), "use_rel": np.int64(use_rel), "after": after.astype(np.int64),
         "step_len": np.array([len(c) for c in ds.calls], dtype=np.int64),
         "step_ids": np.concatenate(ds.calls),
         "step_checksum": np.concatenate([checksum(s["x"]) for s in steps]),
         "step_losses": np.array([s["losses"] for s in steps], dtype=np.float64),
         "loss_keys": np.array(steps[0]["keys"]),
         "rows_tag": np.array([r[0] for r in Recorder.rows]), "rows_value": np.array([r[1] for r in Recorder.rows]),
         "rows_epoch": np.array([r[2] for r in Recorder.rows], dtype=np.int64)}
    # the first training step and the first validation step in full (values are float16-representable: exact)
    n_val = int(np.floor(val_split_ratio * N))
    first_val = int(np.ceil((N - n_val) / BATCH))
    for name, i in (("first_train", 0), ("first_val", first_val)):
        s = steps[i]
        assert torch.equal(s["x"].half().float(), s["x"])
        if transform is not None and i == 0:       # (without augmentation the batch is data[ids]; the checksums cover the rest)
            o[name + "_x_f16"] = s["x"].half().numpy()
        o[name + "_step"] = np.int64(i)
        if s["mask"] is not None:
            o[name + "_mask_u8"] = s["mask"].to(torch.uint8).numpy()
            assert torch.equal(s["mask"], s["mask"].to(torch.uint8).float())
        if s["tm"] is not None:
            o[name + "_tm"] = s["tm"].numpy().astype(np.float32)
    if use_rel:
        o["step_tm"] = np.concatenate([s["tm"].numpy().astype(np.int8).reshape(-1) for s in steps])
    if use_mask:
        mw = weights[:, :1] * 2
        o["step_mask_checksum"] = np.concatenate([(s["mask"].double() * mw).sum(dim=(1, 2, 3)).numpy() for s in steps])
    for k, v in sd0.items():
        o["sd0/" + k] = v.numpy()
    for k, v in ckpt.items():
        o["ckpt/" + k] = v.numpy()
    for k, v in model.state_dict().items():
        o["final/" + k] = v.detach().numpy().copy()
    for k, v in o.items():
        out[f"{tag}/{k}"] = v
    print(tag, cls.__name__, "steps", len(steps), "rows", len(Recorder.rows), "last losses", steps[-1]["losses"])


run("a", ref_vae.VQ_VAE_z16, seed=5, n_epochs=2, lr=1e-4, use_mask=True, use_rel=True, shuffle=False, transform=True,
    val_split_ratio=0.25)
run("b", ref_vq.VQ_VAE, seed=6, n_epochs=3, lr=1e-4, use_mask=False, use_rel=False, shuffle=True, transform=True,
    val_split_ratio=0.25)
run("c", ref_vq.VQ_VAE, seed=7, n_epochs=3, lr=1e-4, use_mask=True, use_rel=False, shuffle=True, transform=None,
    val_split_ratio=0.3)

# ---------------------------------------------------------------- run_one_batch's augmentation on other shapes
# (run_training.py:396-403 through the reference's own run_one_batch with a stand-in model that keeps what it is handed;
# float16-representable values; the seeds are the first ones whose draws cover all 12 (flip, rotation) pairs over the cases)
class KeepBatch:
    def __call__(self, batch, **kwargs):
        self.seen = batch.clone()
        return None, {"total_loss": torch.zeros(())}


def aug_case(seed, shape):
    gg = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(*shape, generator=gg).half().float()
    keep = KeepBatch()
    np.random.seed(seed)
    ns["run_one_batch"](keep, x.clone(), {}, model_kwargs={}, transform=True, training=False)
    return x, keep.seen, np.random.randint(0, 2 ** 31, size=2)


AUG_SHAPES = [(7, 3, 20, 20), (9, 1, 64, 64), (5, 2, 36, 36), (12, 4, 40, 40)]       # (2 x 128 x 128: run a's first batch)
for seed in range(1000):
    pairs = set()
    for i, shape in enumerate(AUG_SHAPES):
        np.random.seed(seed + i)
        for _ in range(shape[0]):
            f = int(np.random.choice([0, 1, 2]))          # (only to pick the seed: the vectors below are the reference's)
            pairs.add((f, int(np.random.choice([0, 1, 2, 3]))))
    if len(pairs) == 12:
        break
for i, shape in enumerate(AUG_SHAPES):
    x, y, after = aug_case(seed + i, shape)
    out[f"aug{i}/seed"] = np.int64(seed + i)
    out[f"aug{i}/x_f16"] = x.half().numpy()
    out[f"aug{i}/y_f16"] = y.half().numpy()
    out[f"aug{i}/after"] = after.astype(np.int64)
    assert torch.equal(y.half().float(), y)
out["aug_cases"] = np.int64(len(AUG_SHAPES))
print("augmentation cases from seed", seed)

path = os.path.join(OUT, "g11_train_loop.npz")
np.savez_compressed(path, **out)
print("g11_train_loop.npz", os.path.getsize(path) / 1024, "KiB", len(out), "arrays")
