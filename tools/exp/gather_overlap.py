#!/usr/bin/env python3
"""Can the resident feed's gather of batch i+1 hide under the step of batch i?  Three timings at B = 2048:
step alone | gather then step on one stream (what train() does) | gather on a side stream, into another buffer, beside the step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import VQ_VAE, ops
from dynamorph_amd.train import FusedTrainer

dev = torch.device("cuda", 0)
B, N = 2048, 8192
torch.manual_seed(0)
tr = FusedTrainer(VQ_VAE().to(dev), lr=1e-4)
x = tr.prepare(torch.randn(B, 2, 128, 128, device=dev))
data = torch.randn(N, 2, 128, 128, device=dev)
ids = torch.randperm(N, device=dev)[:B].to(torch.int32)
fl = torch.randint(0, 3, (B,), device=dev, dtype=torch.int32)
ro = torch.randint(0, 4, (B,), device=dev, dtype=torch.int32)
other = torch.empty_like(x)
side = torch.cuda.Stream()


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def alone():
    tr.step(x)


def serial():
    ops.gather_augment(data, ids, fl, ro, x, B)
    tr.step(x)


def overlapped():
    with torch.cuda.stream(side):
        ops.gather_augment(data, ids, fl, ro, other, B)
    tr.step(x)


print("step alone            %.4f ms" % timed(alone))
print("gather + step, serial %.4f ms" % timed(serial))
print("gather on side stream %.4f ms" % timed(overlapped))
