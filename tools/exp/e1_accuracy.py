#!/usr/bin/env python3
"""Accuracy of the first convolution (enc.0 o enc.1 composite) against float64: max / rms error of a1 and of the gradient of
enc.0.weight after one full step, for the kernel selected by DM_CONV4_PAIR (read once per process)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dynamorph_amd
from dynamorph_amd import engine as E
from oracle import vqvae_oracle as O
import copy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(3)
ref = O.OracleVQVAE()
x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(5))
ref64 = copy.deepcopy(ref).double()
with torch.no_grad():
    a1_64 = ref64.enc[1](ref64.enc[0](x.double()))
    a1_32 = ref.enc[1](ref.enc[0](x))
m = dynamorph_amd.VQ_VAE().to("cuda:0")
m.load_state_dict(ref.state_dict())
L = E.Layers(m)
with torch.no_grad():
    z, cx = E.encoder_forward(L, x.to("cuda:0"))
a1 = cx.a1.cpu().double()
for name, a in (("hip", a1), ("cpu fp32 reference", a1_32.double())):
    e = (a - a1_64).abs()
    print(f"DM_CONV4_PAIR={os.environ.get('DM_CONV4_PAIR','1')}  a1 {name}: max err {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}  (|a1| rms {float(a1_64.pow(2).mean().sqrt()):.3f})")
# per-column error profile (border columns / the last pair)
e = (a1 - a1_64).abs().amax(dim=(0, 1, 2))
print("   max err by output column: first 4", [f"{v:.1e}" for v in e[:4].tolist()], "last 4", [f"{v:.1e}" for v in e[-4:].tolist()], "interior max", f"{float(e[4:-4].max()):.1e}")
