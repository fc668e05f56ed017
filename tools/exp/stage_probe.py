#!/usr/bin/env python3
"""Pageable -> pinned staging rate for DRAM-resident patches (2 GB source, 134 MB batches)."""
import time, torch
from concurrent.futures import ThreadPoolExecutor
torch.zeros(1, device="cuda:0")
N, n = 16384, 1024
x = torch.randn(N, 2, 128, 128)
pin = torch.empty(n, 2, 128, 128, pin_memory=True)
page = torch.empty(n, 2, 128, 128).fill_(0)
mb = pin.numel() * 4 / 1e6
def run(name, f):
    t0 = time.perf_counter()
    for lo in range(0, N, n): f(lo)
    dt = (time.perf_counter() - t0) / (N // n)
    print(f"{name:44s} {dt*1e3:7.2f} ms/batch {mb/dt/1e3:6.1f} GB/s")
print("threads", torch.get_num_threads(), torch.get_num_interop_threads())
run("copy_ -> pinned", lambda lo: pin.copy_(x[lo:lo + n]))
run("copy_ -> pageable", lambda lo: page.copy_(x[lo:lo + n]))
for T in (4, 8, 16, 32):
    pool = ThreadPoolExecutor(T); c = n // T
    run(f"{T} python threads -> pinned", lambda lo: list(pool.map(lambda i: pin[i*c:(i+1)*c].copy_(x[lo+i*c:lo+(i+1)*c]), range(T))))
import numpy as np
pn = pin.numpy(); xn = x.numpy()
run("numpy copyto -> pinned", lambda lo: np.copyto(pn, xn[lo:lo + n]))
xd = x.double()[:4096]
pin64 = torch.empty(n, 2, 128, 128, pin_memory=True)
N = 4096
run("copy_ float64 -> float32 pinned", lambda lo: pin64.copy_(xd[lo:lo + n]))
