#!/usr/bin/env python3
"""Does the next batch's gather + augmentation (HBM-bound, 86 us) hide under the training step's graph when it is launched
on a second stream?  Times K steps (a) alone, (b) each followed by the gather on the same stream (what train() does),
(c) with the gather on a side stream into a second buffer while the step's graph replays."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd import ops
from dynamorph_amd.train import FusedTrainer

dev = "cuda:0"
B, N, K = 2048, 8192, 60
torch.manual_seed(0)
m = dynamorph_amd.VQ_VAE(num_inputs=2, channel_var=np.ones(2)).to(dev)
tr = FusedTrainer(m, lr=1e-4)
data = torch.randn(N, 2, 128, 128, device=dev)
x = tr.prepare(torch.randn(B, 2, 128, 128, device=dev))
other = torch.empty_like(x)
ids = torch.randint(0, N, (B,), device=dev, dtype=torch.int32)
fl = torch.randint(0, 2, (B,), device=dev, dtype=torch.int32)
ro = torch.randint(0, 4, (B,), device=dev, dtype=torch.int32)
side = torch.cuda.Stream(device=dev)
cur = torch.cuda.current_stream(dev)


def run(mode):
    for it in range(K + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "step":
            tr.step(x)
        elif mode == "serial":
            ops.gather_augment(data, ids, fl, ro, x, B)
            tr.step(x)
        else:
            side.wait_stream(cur)                       # the previous step has finished with `other`... (ping-pong stand-in)
            with torch.cuda.stream(side):
                ops.gather_augment(data, ids, fl, ro, other, B)
            tr.step(x)
            cur.wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for mode in ("step", "serial", "overlap", "step", "serial", "overlap"):
    print(f"{mode:8s} {run(mode):.4f} ms per iteration", flush=True)
