#!/usr/bin/env python3
"""Golden fixture for the dataset-ordering steps: tests/golden/g10_relations.npz.

Runs ONLY in the build container, where /root/reference is mounted.  run_training.py does not import here (h5py, cv2,
tensorboard, torchvision are absent), so its two functions are taken out of its syntax tree and executed on their own:
concat_relations (run_training.py:299-321) and reorder_with_trajectories (run_training.py:97-160) need nothing but numpy,
queue, scipy's csr_matrix and TensorDataset.  The reference source never travels; only the .npz written here does.

    cd /tmp && python3 /root/repo/tests/golden/make_golden_relations.py
"""
import ast
import os
import queue

import numpy as np
import torch
from scipy.sparse import csr_matrix
from torch.utils.data import TensorDataset

REF = os.environ.get("DYNAMORPH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

tree = ast.parse(open(os.path.join(REF, "run_training.py")).read())
wanted = ("concat_relations", "reorder_with_trajectories")
mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
ns = {"np": np, "queue": queue, "csr_matrix": csr_matrix, "TensorDataset": TensorDataset, "t": torch}
exec(compile(mod, "run_training.py (two functions)", "exec"), ns)


def make_relations(rng, n, n_traj, max_len):
    """Trajectories of consecutive-frame patches: adjacent frames 2, other pairs of a trajectory 1, both directions,
    inserted the way generate_trajectory_relations-style code does (pair by pair)."""
    ids = rng.permutation(n)
    rel, at = {}, 0
    for _ in range(n_traj):
        ln = int(rng.integers(2, max_len + 1))
        if at + ln > n:
            break
        tr = ids[at:at + ln]
        at += ln
        for i in range(ln):
            for j in range(ln):
                if i != j:
                    rel[(int(tr[i]), int(tr[j]))] = 2 if abs(i - j) == 1 else 1
    return rel


out = {}
rng = np.random.default_rng(2024)
cases = [("a", 40, 6, 5, 123), ("b", 257, 40, 7, 7), ("c", 1000, 150, 6, 123), ("d", 12, 0, 3, 5)]
for name, n, n_traj, max_len, seed in cases:
    rel = make_relations(rng, n, n_traj, max_len)
    if not rel:                                        # the reference cannot build a matrix from no pairs: one lone pair
        rel = {(0, 1): 1, (1, 0): 1}
    data = TensorDataset(torch.arange(n, dtype=torch.float32).reshape(n, 1))
    ds, mat, order = ns["reorder_with_trajectories"](data, rel, seed=seed)
    after = np.random.randint(0, 2 ** 31, size=4)      # where the generator was left
    keys = np.array(list(rel.keys()), dtype=np.int64)
    out[f"{name}_n"] = np.int64(n)
    out[f"{name}_seed"] = np.int64(seed)
    out[f"{name}_pairs"] = keys
    out[f"{name}_values"] = np.array(list(rel.values()), dtype=np.int64)
    out[f"{name}_order"] = np.array([int(i) for i in order], dtype=np.int64)
    out[f"{name}_data"] = ds.tensors[0].numpy().copy()
    out[f"{name}_mat"] = np.asarray(mat.todense()).astype(np.int64)
    out[f"{name}_after"] = after.astype(np.int64)

# concat_relations: two datasets with offsets
r1, r2 = make_relations(rng, 30, 5, 4), make_relations(rng, 25, 4, 4)
l1, l2 = rng.integers(0, 9, 30), rng.integers(0, 7, 25)
merged, labels = ns["concat_relations"]([r1, r2], [l1, l2], [0, 30])
for tag, r in (("cc_r1", r1), ("cc_r2", r2), ("cc_merged", merged)):
    out[tag + "_pairs"] = np.array(list(r.keys()), dtype=np.int64)
    out[tag + "_values"] = np.array(list(r.values()), dtype=np.int64)
out["cc_l1"], out["cc_l2"], out["cc_labels"] = l1.astype(np.int64), l2.astype(np.int64), labels.astype(np.int64)

path = os.path.join(OUT, "g10_relations.npz")
np.savez_compressed(path, **out)
print("g10_relations.npz", os.path.getsize(path) / 1024, "KiB", len(out), "arrays")
