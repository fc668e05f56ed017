import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. `pytest tests/` here."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def codes_gate(flips, z_ref, codebook, what="codes", rel=1e-4):
    """The end-to-end index gate of DESIGN.md section 3: the HIP encoder accumulates in another order than oneDNN, so a
    code may differ from the reference's ONLY where the reference's own two best distances are within `rel` of each other.
    flips: bool (B, H, W), True where the code (or the quantised vector) differs; z_ref: the REFERENCE latents
    (B, D, H, W); codebook (K, D).  Asserts zero flips away from such near-ties and at most 1e-5 * P + 1 flips in all;
    prints the counts."""
    import torch
    flips = torch.as_tensor(flips).bool().cpu()
    z_ref, codebook = torch.as_tensor(z_ref).double().cpu(), torch.as_tensor(codebook).double().cpu()
    B, D, H, W = z_ref.shape
    d = torch.cdist(z_ref.permute(0, 2, 3, 1).reshape(-1, D), codebook).pow(2)            # (P, K)
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    near = ((top2[:, 1] - top2[:, 0]) <= rel * top2[:, 0]).reshape(B, H, W)
    nflip, nnear, P = int(flips.sum()), int(near.sum()), flips.numel()
    print(f"{what}: {nflip} of {P} codes differ, {nnear} reference near-ties (relative gap <= {rel})")
    assert not bool((flips & ~near).any()), f"{what}: {int((flips & ~near).sum())} codes differ away from near-ties"
    assert nflip <= 1e-5 * P + 1, f"{what}: {nflip} flips in {P} positions"
