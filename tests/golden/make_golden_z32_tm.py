#!/usr/bin/env python3
"""Generate tests/golden/g8_z32_tm.npz by importing the reference (build container only; same rules as make_golden.py).

VQ_VAE_z32 (HiddenStateExtractor/vae.py:348-474) forward + backward WITH a time-matching matrix (the weighted / hinge
term on z_after, vae.py:441-455) and a batch mask: losses, reconstruction and every parameter gradient -- the branch the
first round's fixtures left to the oracle alone.

    cd /tmp && python3 /root/repo/tests/golden/make_golden_z32_tm.py
"""
import os
import sys
import types

REF = os.environ.get("DYNAMORPH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import HiddenStateExtractor.vae as ref_vae  # noqa: E402

torch.set_num_threads(8)


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=True)


x = torch.from_numpy(np.load(os.path.join(OUT, "g2_input.npz"))["x"])              # (4, 2, 128, 128)
torch.manual_seed(0)
m = ref_vae.VQ_VAE_z32(device="cpu")
sd0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
tm = torch.tensor([[2., 1., 0., 0.], [1., 2., 1., 0.], [0., 1., 2., 1.], [0., 0., 1., 2.]])
mask = (torch.rand(4, 1, 128, 128, generator=torch.Generator().manual_seed(17)) > 0.35).float()
dec, ld = m(x, time_matching_mat=tm, batch_mask=mask)
ld["total_loss"].backward()
arrs = {"tm": f32(tm), "mask": f32(mask), "decoded": f32(dec)}
arrs.update({k: np.float32(float(v)) for k, v in ld.items()})
arrs.update({f"grad/{k}": f32(p.grad) for k, p in m.named_parameters() if p.grad is not None})
arrs.update({f"sd/{k}": v for k, v in sd0.items()})
path = os.path.join(OUT, "g8_z32_tm.npz")
np.savez_compressed(path, **arrs)
print(f"g8_z32_tm.npz {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays; losses", {k: float(v) for k, v in ld.items()})
