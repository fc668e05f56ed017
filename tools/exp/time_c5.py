import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
from dynamorph_amd.train import FusedTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = dynamorph_amd.VQ_VAE(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4)).cuda()
tr = FusedTrainer(m, lr=1e-4)
x = torch.randn(B, 4, 256, 256, device="cuda")
for _ in range(3): tr.step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): out = tr.step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"C5 VQ_VAE(4ch, K=4096) 256x256 B={B}: {dt*1e3:.2f} ms/step = {B/dt:.0f} patches/s  losses {out.tolist()}", flush=True)
