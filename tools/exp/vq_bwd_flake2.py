#!/usr/bin/env python3
"""Round 5 probe for the one-off failure of tests/test_gpu_kernels.py::test_vq_backward_large_codebooks[512-64-32]
(atomic form of dm_vq_backward: 1 008 of 32 768 elements off by single positions' contributions, once in four full runs).

Puts the test's two calls behind the allocation pattern that preceded them in the suite (the C5 full-size test: ~20 GB of
activations, a captured FusedTrainer step, then everything freed, optionally torch.cuda.empty_cache()) and repeats them.
On a mismatch the wrong elements are attributed: the kernel runs 5 workgroups of 1024 positions (grid-stride, P = 5120),
so the host knows every workgroup's partial sum -- a lost / doubled flush of workgroup w shows as -+ partial[w] on exactly
the elements it touches; a zero fill landing late shows as the sum of the partials flushed before it.

    python tools/exp/vq_bwd_flake2.py [iterations] [--zero torch|sync] [--empty-cache 0|1] [--c5 0|1]
"""
import argparse
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("iters", nargs="?", type=int, default=200)
ap.add_argument("--zero", default="torch", choices=["torch", "sync"])
ap.add_argument("--empty-cache", type=int, default=1)
ap.add_argument("--c5", type=int, default=1)
ap.add_argument("--churn-every", type=int, default=25, help="repeat the big allocate / free (+ empty_cache) every N iterations")
args = ap.parse_args()
DEV = "cuda:0"
K, D, H, B = 512, 64, 32, 5


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def c5_pattern():
    """What ran before the failing test: the stress model at its bench batch, graph capture included."""
    import copy
    import dynamorph_amd
    from dynamorph_amd.train import FusedTrainer
    torch.manual_seed(5)
    m = dynamorph_amd.VQ_VAE(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4)).to(DEV)
    x = torch.randn(1024, 4, 256, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(77))
    for graph in (True, False):
        mm = copy.deepcopy(m)
        tr = FusedTrainer(mm, lr=1e-3, use_graph=graph)
        tr.step(x)
        del tr, mm
        gc.collect()
    del m, x
    gc.collect()


def churn():
    if args.c5:
        c5_pattern()
    else:
        junk = [torch.randn(1 << 28, device=DEV) for _ in range(8)]      # 8 GB
        del junk
    gc.collect()
    if args.empty_cache:
        torch.cuda.empty_cache()


z, cb, g = rnd(B, D, H, H, seed=51), rnd(K, D, seed=52), rnd(B, D, H, H, seed=53)
zd, cbd, gd = z.to(DEV), cb.to(DEV), g.to(DEV)
idx, _, _, _ = ops.vq_forward(zd, cbd, want_out=False)
gl = torch.tensor([1.3], device=DEV)
ih = idx.cpu().reshape(-1)
q = cb[ih]                                                             # (P, D)
zp = z.permute(0, 2, 3, 1).reshape(-1, D)
N = z.numel()
contrib = (1.3 * 2 * (q.double() - zp.double()) / N)                    # (P, D) per-position contribution
P = ih.numel()
nwg = (P + 1023) // 1024
part = torch.zeros(nwg, K, D, dtype=torch.float64)
for w in range(nwg):
    sl = slice(w * 1024, min(P, (w + 1) * 1024))
    part[w].index_add_(0, ih[sl], contrib[sl])
ref = part.sum(0)
tol = 1e-5 * ref.abs() + 1e-6 * float(ref.abs().max())
print(f"P = {P}, {nwg} workgroups; zero fill: {args.zero}; empty_cache: {args.empty_cache}; c5 pattern: {args.c5}", flush=True)

bad = 0
for it in range(args.iters):
    if it % args.churn_every == 0:
        churn()
        print(f"iter {it}: churned ({torch.cuda.memory_reserved() >> 20} MB reserved)", flush=True)
    if args.zero == "torch":
        dw = torch.zeros(K, D, device=DEV)
    elif args.zero == "sync":
        dw = torch.zeros(K, D, device=DEV)
        torch.cuda.synchronize()
    _, dwa = ops.vq_backward(zd, cbd, idx, g.to(DEV), gl, 0.25, dw=dw)
    _, dws = ops.vq_backward(zd, cbd, idx, g.to(DEV), gl, 0.25)
    da = dwa.cpu().double() - ref
    ds = dws.cpu().double() - ref
    wa, ws = (da.abs() > tol), (ds.abs() > tol)
    for name, dd, ww in (("atomic", da, wa), ("slab", ds, ws)):
        if not bool(ww.any()):
            continue
        bad += 1
        flat = ww.reshape(-1).nonzero().reshape(-1)
        rows = torch.unique(flat // D)
        print(f"iter {it}: {name} form: {flat.numel()} elements off in {rows.numel()} codes; flat {int(flat.min())}..{int(flat.max())}; "
              f"max err {float(dd.abs().max()):.3e}; ptr {dwa.data_ptr():#x}", flush=True)
        # which workgroup's partial explains the error on the wrong elements?
        for w in range(nwg):
            for sign, what in ((-1.0, "LOST"), (1.0, "DOUBLED")):
                resid = (dd - sign * part[w])[ww].abs().max()
                if float(resid) <= 1e-9:
                    print(f"   = partial of workgroup {w} {what} on those elements", flush=True)
        # contiguity of the wrong range, in units of a flush iteration (1024 consecutive floats) and of a code row (64)
        seg = torch.unique(flat // 1024)
        print(f"   flush iterations touched: {seg.tolist()[:16]}; code rows: {rows.tolist()[:24]}", flush=True)
        cols = torch.unique(flat % D)
        print(f"   columns d: {cols.tolist()}", flush=True)
        np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", f"vq_bwd_flake_{name}_{it}.npy"),
                dd.numpy())
print("mismatching calls:", bad, "of", 2 * args.iters, flush=True)
