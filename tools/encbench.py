#!/usr/bin/env python3
"""Host-to-host rate of encode_patches (C2 with the PCIe transfers inside): patches start in host memory (pageable or
pinned), latents end in host numpy arrays.  The bench line's C2 figure has its inputs resident in HBM; this is the
PCIe-inclusive number DESIGN.md quotes next to it.

    gpurun -- python tools/encbench.py [N] [batch]
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynamorph_amd
from dynamorph_amd.patch_vae import encode_patches

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
m = dynamorph_amd.VQ_VAE().to("cuda:0")
x = torch.randn(N, 2, 128, 128)
for name, src in (("pageable", x), ("pinned", x.pin_memory()), ("pageable f64", x[:N // 4].double())):
    N = src.shape[0]
    encode_patches(m, src[:2 * bs], device="cuda:0", batch_size=bs)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        zb, za = encode_patches(m, src, device="cuda:0", batch_size=bs)
        best = min(best, time.perf_counter() - t0)
    gb = N * (2 * 128 * 128 * 4 + 2 * 16 * 16 * 16 * 4) / 1e9
    print(f"{name:9s} N={N} batch={bs}: {best * 1e3:8.1f} ms  {N / best:10.0f} patches/s  {gb / best:6.1f} GB/s over PCIe (both directions)")
