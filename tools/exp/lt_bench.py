#!/usr/bin/env python3
"""Latent-tail kernel alone (csrc/latent_tail.hip) on 1024 patches, graph-timed; DM_LT_DBG ablations are set by the caller:
    for d in 0 1 2 4 3 7; do DM_LT_DBG=$d python3 tools/exp/lt_bench.py; done"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd import engine as E, ops

dev = "cuda:0"
B = int(os.environ.get("LT_B", "1024"))
m = dynamorph_amd.VQ_VAE().to(dev)
L = E.Layers(m)
a3 = torch.randn(B, 16, 16, 16, device=dev)
coef3 = torch.tensor([1.0, 0.0, 0.1, 0.0], device=dev).repeat(B, 16, 1).contiguous()
w = lambda p: p.detach()
res = [(w(ca.weight), w(ca.bias), w(bna.weight), w(bna.bias), bna.eps, w(cb.weight), w(cb.bias), w(bnb.weight), w(bnb.bias), bnb.eps)
       for ca, bna, cb, bnb in L.res]
call = lambda: ops.latent_tail_forward(a3, coef3, w(L.enc10.weight), w(L.enc10.bias), w(L.bn4.weight), w(L.bn4.bias), L.bn4.eps, res)
for _ in range(3):
    call()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        call()
for _ in range(3):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record(); e1.synchronize()
print(f"DM_LT_DBG={os.environ.get('DM_LT_DBG', '0')}  B={B}: {e0.elapsed_time(e1) * 10:.1f} us per launch")
