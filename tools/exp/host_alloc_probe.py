#!/usr/bin/env python3
"""Cost of making the host result arrays of encode_patches ready to receive DMA: fresh pageable (first touch), pageable
pre-faulted by a parallel fill, pinned."""
import time, torch
torch.cuda.init(); torch.zeros(1, device="cuda:0")
torch.empty(1 << 20).pin_memory()
N, L = 16384, 4096
mb = N * L * 4 / 1e6
def clock(name, f):
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:46s} {dt*1e3:8.1f} ms  {mb/dt/1e3:6.1f} GB/s"); return r
dev = torch.randn(N, L, device="cuda:0")
a = clock("torch.empty pageable", lambda: torch.empty(N, L))
clock("D2H into fresh pageable (first touch)", lambda: a.copy_(dev))
clock("D2H into the same pageable again", lambda: a.copy_(dev))
b = clock("torch.empty + fill_(0) (parallel first touch)", lambda: torch.empty(N, L).fill_(0))
clock("D2H into pre-faulted pageable", lambda: b.copy_(dev))
c = clock("torch.empty(pin_memory=True)", lambda: torch.empty(N, L, pin_memory=True))
clock("D2H into pinned", lambda: c.copy_(dev, non_blocking=True))
del c
c = clock("torch.empty(pin_memory=True) again (cached)", lambda: torch.empty(N, L, pin_memory=True))
x = torch.randn(4096, 2, 128, 128); xm = x.numel() * 4 / 1e6
xd = torch.empty_like(x, device="cuda:0")
t0 = time.perf_counter(); xd.copy_(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"H2D from pageable, DRAM resident {xm:.0f} MB: {dt*1e3:.1f} ms {xm/dt/1e3:.1f} GB/s")
xp = x.pin_memory()
t0 = time.perf_counter(); xd.copy_(xp, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"H2D from pinned, DRAM resident {xm:.0f} MB: {dt*1e3:.1f} ms {xm/dt/1e3:.1f} GB/s")
