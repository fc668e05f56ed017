#!/usr/bin/env python3
"""The full-size C3 gradient gate (tests/test_gpu_fullsize.py) as a table: per tensor, the HIP gradient's and the fp32
reference's max error against the float64 truth and their ratio, for the kernels the environment selects (DM_CONV4_PAIR etc.).
Caches the oracle's two runs in /tmp so that several configurations can be compared in one call."""
import copy, gc, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dynamorph_amd
from dynamorph_amd.train import FusedTrainer
from oracle import vqvae_oracle as O
B = 2048
torch.manual_seed(2048)
ref = O.OracleVQVAE()
ref.vq.chunk = 64
x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234))
cache = "/tmp/c3_grads_cache.pt"
sd0 = copy.deepcopy(ref.state_dict())
if os.path.exists(cache):
    g32, g64 = torch.load(cache)
else:
    ref64 = copy.deepcopy(ref).double()
    _, ld64 = ref64(x.double()); ld64["total_loss"].backward()
    g64 = {k: p.grad for k, p in ref64.named_parameters() if p.grad is not None}
    del ld64; gc.collect()
    _, ld_r = ref(x); ld_r["total_loss"].backward()
    g32 = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    torch.save((g32, g64), cache)
m = dynamorph_amd.VQ_VAE().to("cuda:0")
m.load_state_dict(sd0)
tr = FusedTrainer(m, lr=1e-4, use_graph=False)
tr.forward_backward(x.to("cuda:0"))
tr.expose_grads()
print("config:", {k: v for k, v in os.environ.items() if k.startswith("DM_")})
worst = 0
for k, p in m.named_parameters():
    if not p.requires_grad or k not in g64: continue
    t = g64[k]; scale = max(float(t.abs().max()), 1e-6)
    e_ref = float((g32[k].double() - t).abs().max()); e_hip = float((p.grad.cpu().double() - t).abs().max())
    flag = "" if e_hip <= max(1.5 * e_ref, 2e-4 * scale) else "  <-- over the gate"
    print(f"{k:34s} scale {scale:9.2e}  ref {e_ref:9.2e}  hip {e_hip:9.2e}  ratio {e_hip / max(e_ref, 1e-30):6.2f}  hip/scale {e_hip / scale:8.1e}{flag}")
