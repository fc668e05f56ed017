import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd.patch_vae import encode_patches
N, bs = 16384, 1024
m = dynamorph_amd.VQ_VAE().to("cuda:0")
x = torch.randn(N, 2, 128, 128)
encode_patches(m, x[:2 * bs], device="cuda:0", batch_size=bs)
encode_patches(m, x, device="cuda:0", batch_size=bs)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); encode_patches(m, x, device="cuda:0", batch_size=bs); dt = time.perf_counter() - t0
pr.disable()
print(f"{dt*1e3:.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
