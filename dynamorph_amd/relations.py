"""The steps between the pickled datasets and train(): run_training.py:856-891 (main) calls concat_relations (299-321)
and reorder_with_trajectories (97-160) before train (455-551).  Same names, arguments and results.

reorder_with_trajectories is the one that costs: the reference rebuilds `list(inds_pool)` and an array of it for every
random pick -- quadratic in the number of patches (minutes at 2*10^4, hours at 10^5).  Here the order comes from the
library's host function dm_reorder_with_trajectories (a Fenwick tree over the remaining ids, the legacy generator's
32-bit words parsed the way np.random.choice consumes them), so np.random.seed(seed) gives the reference's order and
leaves numpy's generator where the reference leaves it.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def concat_relations(relations, labels, offsets):
    """run_training.py:299-321: the relation dicts of several datasets as one, ids shifted by each dataset's offset;
    labels shifted the same way and concatenated."""
    new_relations = {}
    new_labels = []
    for relation, label, offset in zip(relations, labels, offsets):
        new_relations.update({(id1 + offset, id2 + offset): v for (id1, id2), v in relation.items()})
        new_labels.append(label + offset)
    return new_relations, np.concatenate(new_labels, axis=0)


def _adjacency(relations, n):
    """CSR over the first id of the value-2 pairs, each row in the dict's order (run_training.py:116-120)."""
    keys = np.fromiter((k[j] for k, v in relations.items() if v == 2 for j in (0, 1)), dtype=np.int64).reshape(-1, 2)
    if len(keys) and (keys.min() < 0 or keys[:, 0].max() >= n):
        raise IndexError("relation ids outside the dataset")
    order = np.argsort(keys[:, 0], kind="stable")
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(keys[:, 0], minlength=n), out=ptr[1:])
    return ptr, np.ascontiguousarray(keys[order, 1])


def trajectory_order(n, relations, seed=None):
    """inds_in_order of run_training.py:112-140 for n samples (numpy int64 array); seeds / advances numpy's global
    legacy generator exactly as the reference does."""
    if seed is not None:
        np.random.seed(seed)
    order = np.empty(n, np.int64)
    if n == 0:
        return order
    ptr, idx = _adjacency(relations, n)
    lib = L.load()
    err = C.c_int64(-1)
    state = np.random.get_state()
    m = n + n // 2 + 64                              # < 2 words per pick on average, at most one pick per sample
    while True:
        raw = np.random.randint(0, 2 ** 32, size=m, dtype=np.uint32)
        np.random.set_state(state)
        used = lib.dm_reorder_with_trajectories(raw.ctypes.data, m, n, ptr.ctypes.data, idx.ctypes.data, order.ctypes.data,
                                                C.byref(err))
        if used != -1:
            break
        m *= 2
    if used in (-2, -3):
        raise KeyError(int(err.value))               # relation_dict[elem] / inds_pool.remove(e) in the reference
    if used < 0:
        raise RuntimeError("dm_reorder_with_trajectories failed")
    if used:
        np.random.randint(0, 2 ** 32, size=int(used), dtype=np.uint32)      # advance by exactly the words consumed
    return order


def reorder_with_trajectories(dataset, relations, seed=None):
    """run_training.py:97-160.  dataset: a TensorDataset; relations: {(i, j): 1 (same trajectory) | 2 (adjacent frames)}.
    Returns (TensorDataset reordered, scipy CSR relation matrix in the new order, inds_in_order as a list of int)."""
    from scipy.sparse import csr_matrix
    from torch.utils.data import TensorDataset
    n = len(dataset)
    inds = trajectory_order(n, relations, seed)
    new_tensor = dataset.tensors[0][inds]
    values = [v for v in relations.values() if v in (1, 2)]
    pairs = np.array(list(relations.keys()))
    if len(values) != len(relations):
        raise ValueError("relations hold values other than 1 and 2")     # (the reference fails in csr_matrix there)
    relation_mat = csr_matrix((np.array(values), (pairs[:, 0], pairs[:, 1])), shape=(n, n))
    relation_mat = relation_mat[inds][:, inds]
    return TensorDataset(new_tensor), relation_mat, inds.tolist()
