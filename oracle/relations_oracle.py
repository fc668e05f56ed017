"""oracle/relations_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-Python restatement of the two host steps the reference runs between its pickled datasets and train()
(run_training.py:856-891): concat_relations (run_training.py:299-321) and reorder_with_trajectories
(run_training.py:97-160), step for step -- including the quadratic `np.random.choice(list(pool))` per pick, which is what
fixes the order and the consumption of numpy's legacy generator.  Only tests/ may import it; dynamorph_amd/ never does.

Pinned: tests/test_oracle.py checks it against tests/golden/g10_relations.npz, produced by executing the reference's own
two functions (tests/golden/make_golden_relations.py).
"""
import queue

import numpy as np
from scipy.sparse import csr_matrix


def concat_relations(relations, labels, offsets):
    """run_training.py:299-321"""
    merged, shifted = {}, []
    for relation, label, offset in zip(relations, labels, offsets):
        for (a, b), v in relation.items():
            merged[(a + offset, b + offset)] = v
        shifted.append(label + offset)
    return merged, np.concatenate(shifted, axis=0)


def reorder_indices(n, relations, seed=None):
    """run_training.py:110-140: the order only"""
    if seed is not None:
        np.random.seed(seed)
    pool = set(range(n))
    order = []
    following = {}
    for pair, v in relations.items():
        if v == 2:
            following.setdefault(pair[0], []).append(pair[1])
    while pool:
        pick = np.random.choice(list(pool))
        if pick not in following:
            order.append(pick)
            pool.remove(pick)
            continue
        traj = [pick]
        q = queue.Queue()
        q.put(pick)
        while not q.empty():
            for e in following[q.get_nowait()]:
                if e not in traj:
                    traj.append(e)
                    q.put(e)
        order.extend(traj)
        for e in traj:
            pool.remove(e)
    return [int(i) for i in order]


def relation_matrix(n, relations, order):
    """run_training.py:143-159: the (n, n) CSR matrix of the pairs, rows and columns in the new order"""
    pairs = np.array(list(relations.keys()))
    vals = np.array([v for v in relations.values() if v in (1, 2)])
    mat = csr_matrix((vals, (pairs[:, 0], pairs[:, 1])), shape=(n, n))
    idx = np.array(order)
    return mat[idx][:, idx]
