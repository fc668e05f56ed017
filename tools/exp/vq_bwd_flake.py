#!/usr/bin/env python3
"""Diagnostic: tests/test_gpu_kernels.py::test_vq_backward_large_codebooks[512-64-32] failed once in a full-suite run (1008 of
32768 elements of the ATOMIC form's codebook gradient off by ~3 %).  Repeat the two calls and report where the forms differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops

DEV = "cuda:0"
K, D, H, B = 512, 64, 32, 5
g = torch.Generator().manual_seed(51)
z = torch.randn(B, D, H, H, generator=g).to(DEV)
cb = torch.randn(K, D, generator=g).to(DEV)
go = torch.randn(B, D, H, H, generator=g).to(DEV)
idx, _, _, _ = ops.vq_forward(z, cb, want_out=False)
gl = torch.tensor([1.3], device=DEV)
_, ref = ops.vq_backward(z, cb, idx, go, gl, 0.25)                     # slab form
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    junk = torch.randn(1 << 20, device=DEV)                               # churn the allocator a little
    dw = torch.zeros(K, D, device=DEV)
    _, dwa = ops.vq_backward(z, cb, idx, go, gl, 0.25, dw=dw)
    _, dws = ops.vq_backward(z, cb, idx, go, gl, 0.25)
    ea = (dwa - ref).abs()
    es = (dws - ref).abs()
    tol = 1e-5 * ref.abs() + 1e-6 * float(ref.abs().max())
    na, ns = int((ea > tol).sum()), int((es > 0).sum())
    if na or ns:
        bad += 1
        w = (ea > tol).reshape(-1).nonzero().reshape(-1)
        print(f"iter {it}: atomic form {na} off (flat index {int(w.min()) if na else -1}..{int(w.max()) if na else -1}, "
              f"max err {float(ea.max()):.3e}); slab form {ns} elements differ from its first run")
    del junk
print("iterations with a mismatch:", bad)
