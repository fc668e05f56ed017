#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the reference.

Runs ONLY in the build container, where /root/reference is mounted.  The
reference source never travels; only the .npz files written here do.

    cd /tmp && python3 /root/repo/tests/golden/make_golden.py

Fixture ids follow SURVEY.md section 8c (G1..G9).  Everything is produced by
the reference's own classes:
    HiddenStateExtractor/vq_vae.py : VectorQuantizer (25-116), VQ_VAE (228-342)
    HiddenStateExtractor/vae.py    : VQ_VAE_z16 (216-346), VQ_VAE_z32 (348-474)
    pipeline/train_utils.py        : zscore (228-250), zscore_patch (252-274)
and torch.optim.Adam exactly as run_training.py:485 constructs it.
"""
import os
import sys
import types

REF = os.environ.get("DYNAMORPH_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
# vq_vae.py:8 does a top-level `import cv2` the model classes never use.
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import HiddenStateExtractor.vq_vae as ref_vq  # noqa: E402
import HiddenStateExtractor.vae as ref_vae  # noqa: E402
from pipeline.train_utils import zscore, zscore_patch  # noqa: E402

torch.set_num_threads(8)


def sd_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name:34s} {os.path.getsize(path) / 1024:9.1f} KiB  {len(arrs)} arrays")


def fresh(cls=ref_vq.VQ_VAE, **kw):
    torch.manual_seed(0)
    return cls(device="cpu", **kw)


def f32(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=True)


# ---------------------------------------------------------------- G1, G2
model = fresh()
g1 = sd_np(model.state_dict())
save("g1_state_dict.npz", **g1)

torch.manual_seed(1)
x = torch.randn(4, 2, 128, 128)
save("g2_input.npz", x=f32(x))

# ---------------------------------------------------------------- G3..G5
# batch-statistics BatchNorm (how run_training.py drives the model: train mode)
acts = {}


def hook(name):
    def fn(mod, inp, out):
        acts[name] = f32(out)
    return fn


model = fresh()
handles = []
for i in range(12):
    handles.append(model.enc[i].register_forward_hook(hook(f"enc{i}")))
for li in range(2):
    for j in range(6):
        handles.append(model.enc[12].layers[li][j].register_forward_hook(hook(f"res{li}_{j}")))
for i in range(7):
    handles.append(model.dec[i].register_forward_hook(hook(f"dec{i}")))
z_before = model.enc(x)
for h in handles:
    h.remove()
sd_after_enc = sd_np(model.state_dict())
idx = model.vq.encode_inputs(z_before)
w = model.vq.w.weight
dist = torch.sum((z_before.unsqueeze(1) - w.reshape((1, 64, 16, 1, 1))) ** 2, 2)
zq, vq_loss, vq_perp = model.vq(z_before)

# per-sample calls = process_VAE semantics (pipeline/patch_VAE.py:445-452)
model_ps = fresh()
zb_ps, za_ps, idx_ps = [], [], []
for i in range(x.shape[0]):
    zb = model_ps.enc(x[i:i + 1])
    za, _, _ = model_ps.vq(zb)
    zb_ps.append(f32(zb))
    za_ps.append(f32(za))
    idx_ps.append(model_ps.vq.encode_inputs(zb).numpy())
sd_after_ps = sd_np(model_ps.state_dict())

save("g3_encoder.npz",
     z_before=f32(z_before),
     z_before_per_sample=np.concatenate(zb_ps, 0),
     z_after_per_sample=np.concatenate(za_ps, 0),
     idx_per_sample=np.concatenate(idx_ps, 0),
     **{f"rs_batch/{k}": v for k, v in sd_after_enc.items() if "running" in k or "tracked" in k},
     **{f"rs_ps/{k}": v for k, v in sd_after_ps.items() if "running" in k or "tracked" in k})
# enc0 (the 1x1 conv output, 2 MiB) is dropped: the HIP design folds it into enc1 and never materialises it.
# enc2/enc3 are kept for sample 0 only.
save("g3_debug_acts.npz", **{k: (v[:1] if k in ("enc2", "enc3") else v) for k, v in acts.items()
                              if not k.startswith("dec") and k != "enc0"})
save("g4_vq_indices.npz", idx=idx.numpy(), dist_sample0=f32(dist[0]),
     codebook=f32(w))
save("g5_vq_forward.npz", z_before=f32(z_before), quantized=f32(zq),
     loss=f32(vq_loss), perplexity=f32(vq_perp))

# full forward, no mask / with mask
model = fresh()
dec_acts = {}
handles = [model.dec[i].register_forward_hook(
    (lambda n: (lambda m, i_, o: dec_acts.__setitem__(n, f32(o))))(f"dec{i}")) for i in range(7)]
decoded, losses = model(x)
for h in handles:
    h.remove()
save("g5_forward.npz", decoded=f32(decoded),
     **{k: np.float32(float(v)) for k, v in losses.items()})
# the two 4x128x128 tensors are kept for sample 0 only
save("g5_debug_dec_acts.npz", **{k: (v[:1] if k in ("dec4", "dec5") else v) for k, v in dec_acts.items()
                                  if k != "dec6"})

torch.manual_seed(2)
mask = (torch.rand(4, 1, 128, 128) > 0.4).float() * 0.5 + 0.5   # get_mask gives values in {0.5,1}: run_training.py:372
model = fresh()
decoded_m, losses_m = model(x, batch_mask=mask)
save("g5_forward_masked.npz", mask=f32(mask), decoded=f32(decoded_m),
     **{k: np.float32(float(v)) for k, v in losses_m.items()})

# ---------------------------------------------------------------- G6 gradients
model = fresh()
z_b = model.enc(x)
z_b.retain_grad()
z_a, c_loss, perp = model.vq(z_b)
z_a.retain_grad()
dec = model.dec(z_a)
recon = torch.mean(torch.nn.functional.mse_loss(dec, x, reduction="none") / model.channel_var)
total = model.weight_recon * recon + model.weight_commitment * c_loss
total.backward()
grads = {f"grad/{k}": f32(p.grad) for k, p in model.named_parameters() if p.grad is not None}
save("g6_grads.npz", dz_before=f32(z_b.grad), dz_after=f32(z_a.grad),
     total_loss=np.float32(float(total)), **grads)

# VQ-only backward with a random upstream gradient
model = fresh()
zb_leaf = z_before.detach().clone().requires_grad_(True)
zq2, l2, _ = model.vq(zb_leaf)
torch.manual_seed(3)
g_up = torch.randn_like(zq2)
(torch.sum(zq2 * g_up) + 1.7 * l2).backward()
save("g6_vq_backward.npz", z=f32(zb_leaf), g_out=f32(g_up), g_loss=np.float32(1.7),
     dz=f32(zb_leaf.grad), dw=f32(model.vq.w.weight.grad))

# ---------------------------------------------------------------- G7 Adam steps
model = fresh()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(.9, .999))   # run_training.py:485
model.zero_grad()
step_losses = []
snaps = {}
for step in range(3):
    _, ld = model(x)
    ld["total_loss"].backward()
    opt.step()
    model.zero_grad()
    step_losses.append([float(ld[k]) for k in ("recon_loss", "commitment_loss", "total_loss", "perplexity")])
    if step in (0, 2):
        for k, v in sd_np(model.state_dict()).items():
            snaps[f"step{step + 1}/{k}"] = v
save("g7_adam.npz", losses=np.asarray(step_losses, np.float32), **snaps)

# ---------------------------------------------------------------- G8 variants
torch.manual_seed(0)
m16 = ref_vae.VQ_VAE_z16(device="cpu")
assert all(np.array_equal(v, g1[k]) for k, v in sd_np(m16.state_dict()).items())
tm = torch.tensor([[2., 1., 0., 0.], [1., 2., 1., 0.], [0., 1., 2., 1.], [0., 0., 1., 2.]])
dec16, l16 = m16(x, time_matching_mat=tm)
l16["total_loss"].backward()
save("g8_z16_time_matching.npz", tm=f32(tm), decoded=f32(dec16),
     **{k: np.float32(float(v)) for k, v in l16.items()},
     **{f"grad/{k}": f32(p.grad) for k, p in m16.named_parameters() if p.grad is not None})

model = fresh()
dec_tm, l_tm = model(x, time_matching_mat=tm)    # vq_vae.py:324-332 (sum form)
l_tm["total_loss"].backward()
save("g8_vqvae_time_matching.npz", tm=f32(tm), decoded=f32(dec_tm),
     **{k: np.float32(float(v)) for k, v in l_tm.items()},
     **{f"grad/{k}": f32(p.grad) for k, p in model.named_parameters() if p.grad is not None})

torch.manual_seed(0)
m32 = ref_vae.VQ_VAE_z32(device="cpu")
zb32 = m32.enc(x)
dec32, l32 = m32(x)
save("g8_z32.npz", z_before=f32(zb32), decoded=f32(dec32),
     **{k: np.float32(float(v)) for k, v in l32.items()},
     **{f"sd/{k}": v for k, v in sd_np(m32.state_dict()).items()})

# ---------------------------------------------------------------- G9 stress / helpers
torch.manual_seed(0)
vq_big = ref_vq.VectorQuantizer(16, 4096, device="cpu")
torch.manual_seed(5)
z_big = torch.randn(2, 16, 32, 32)
q_big, l_big, p_big = vq_big(z_big)
save("g9_vq_k4096.npz", codebook=f32(vq_big.w.weight), z=f32(z_big),
     idx=vq_big.encode_inputs(z_big).numpy(), quantized=f32(q_big),
     loss=f32(l_big), perplexity=f32(p_big))

# D=64 codebook: exercises the "blocks of 16 along d" summation order
torch.manual_seed(0)
vq_d64 = ref_vq.VectorQuantizer(64, 512, device="cpu")
torch.manual_seed(6)
z_d64 = torch.randn(2, 64, 16, 16)
dist64 = torch.sum((z_d64.unsqueeze(1) - vq_d64.w.weight.reshape((1, 512, 64, 1, 1))) ** 2, 2)
q64, l64, p64 = vq_d64(z_d64)
save("g9_vq_d64.npz", codebook=f32(vq_d64.w.weight), z=f32(z_d64),
     idx=vq_d64.encode_inputs(z_d64).numpy(), dist_sample0=f32(dist64[0]),
     loss=f32(l64), perplexity=f32(p64))

# ties / NaN rule of argmax(-dist) (vq_vae.py:68)
cb_tie = np.zeros((8, 16), np.float32)
cb_tie[1] = 1.0
cb_tie[2] = 1.0          # duplicate of code 1 -> first index wins
cb_tie[5] = -1.0
z_tie = np.zeros((1, 16, 4, 4), np.float32)
z_tie[0, :, 0, 0] = 1.0   # ties codes 1 and 2
z_tie[0, :, 0, 1] = 0.0   # ties codes 0,3,4,6,7
z_tie[0, :, 0, 2] = np.nan
z_tie[0, :, 0, 3] = -1.0
vq_t = ref_vq.VectorQuantizer(16, 8, device="cpu")
with torch.no_grad():
    vq_t.w.weight.copy_(torch.from_numpy(cb_tie))
save("g9_vq_ties.npz", codebook=cb_tie, z=z_tie,
     idx=vq_t.encode_inputs(torch.from_numpy(z_tie)).numpy())

rng = np.random.RandomState(7)
patches = rng.rand(3, 2, 1, 8, 8) * 100.0
save("g9_zscore.npz", patches=patches,
     zscore_patch=zscore_patch(np.squeeze(patches)),
     zscore=zscore(np.squeeze(patches)),
     zscore_given=zscore(np.squeeze(patches), channel_mean=[40., 55.], channel_std=[20., 30.]))

print("done")
