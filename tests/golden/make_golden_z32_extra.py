#!/usr/bin/env python3
"""Generate tests/golden/g8_z32_extra.npz by importing the reference (build container only; same rules as make_golden.py).

VQ_VAE_z32 (HiddenStateExtractor/vae.py:348-474) with extra_loss = {name: loss_fn} (vae.py:463-469): every
`loss_fn(labels, z_after_flat)` is added to total_loss times self.alpha and listed in the loss dict.  The reference's
constructor never sets `alpha` (its signature has no such argument): the attribute is assigned here, as a caller must.
Losses, reconstruction and every parameter gradient of one forward + backward with labels, a time-matching matrix and
the two losses of tests/helpers/extra_losses.py.

    cd /tmp && python3 /root/repo/tests/golden/make_golden_z32_extra.py
"""
import os
import sys
import types

REF = os.environ.get("DYNAMORPH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(os.path.dirname(OUT), "helpers"))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import HiddenStateExtractor.vae as ref_vae  # noqa: E402
from extra_losses import EXTRA  # noqa: E402

torch.set_num_threads(8)


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=True)


x = torch.from_numpy(np.load(os.path.join(OUT, "g2_input.npz"))["x"])              # (4, 2, 128, 128)
torch.manual_seed(0)
m = ref_vae.VQ_VAE_z32(device="cpu", extra_loss=dict(EXTRA))
m.alpha = 0.05
sd0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
labels = torch.tensor([0, 1, 1, 0])
tm = torch.tensor([[2., 1., 0., 0.], [1., 2., 1., 0.], [0., 1., 2., 1.], [0., 0., 1., 2.]])
dec, ld = m(x, labels=labels, time_matching_mat=tm)
ld["total_loss"].backward()
arrs = {"tm": f32(tm), "labels": labels.numpy().astype(np.int64), "alpha": np.float32(m.alpha), "decoded": f32(dec),
        "loss_keys": np.array(list(ld.keys()))}
arrs.update({f"loss/{k}": np.float32(float(v)) for k, v in ld.items()})
arrs.update({f"grad/{k}": f32(p.grad) for k, p in m.named_parameters() if p.grad is not None})
arrs.update({f"sd/{k}": v for k, v in sd0.items()})
path = os.path.join(OUT, "g8_z32_extra.npz")
np.savez_compressed(path, **arrs)
print(f"g8_z32_extra.npz {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays; losses", {k: float(v) for k, v in ld.items()})
