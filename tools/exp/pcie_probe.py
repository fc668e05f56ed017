#!/usr/bin/env python3
"""Host-side transfer probe for encode_patches: rates of the candidate ways to move a batch of patches to the GPU."""
import time, torch, ctypes
from concurrent.futures import ThreadPoolExecutor
n = 1024
x = torch.randn(n, 2, 128, 128)
mb = x.numel() * 4 / 1e6
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
print("torch threads", torch.get_num_threads())
t0 = time.perf_counter(); pin = torch.empty_like(x).pin_memory(); print(f"pin_memory alloc {mb:.0f} MB: {(time.perf_counter()-t0)*1e3:.1f} ms")
dev = torch.empty_like(x, device="cuda:0")
d = t(lambda: pin.copy_(x)); print(f"host copy_ pageable->pinned: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
pool = ThreadPoolExecutor(8)
def par():
    list(pool.map(lambda i: pin[i*128:(i+1)*128].copy_(x[i*128:(i+1)*128]), range(8)))
d = t(par); print(f"host copy_ 8 threads: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
d = t(lambda: dev.copy_(pin, non_blocking=True)); print(f"H2D from pinned: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
d = t(lambda: dev.copy_(x)); print(f"H2D from pageable: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
d = t(lambda: pin.copy_(dev, non_blocking=True)); print(f"D2H to pinned: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
d = t(lambda: x.copy_(dev)); print(f"D2H to pageable: {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
rt = torch.cuda.cudart()
big = torch.randn(8 * n, 2, 128, 128)
t0 = time.perf_counter(); r = rt.cudaHostRegister(big.data_ptr(), big.numel() * 4, 0); dt = time.perf_counter() - t0
print(f"cudaHostRegister {8*mb:.0f} MB: rc={r} {dt*1e3:.1f} ms {8*mb/dt/1e3:.1f} GB/s equivalent")
d = t(lambda: dev.copy_(big[:n], non_blocking=True)); print(f"H2D from registered (is_pinned={big.is_pinned()}): {d*1e3:.2f} ms {mb/d/1e3:.1f} GB/s")
t0 = time.perf_counter(); r = rt.cudaHostUnregister(big.data_ptr()); print(f"unregister rc={r} {(time.perf_counter()-t0)*1e3:.1f} ms")
