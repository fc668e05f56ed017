import sys, time, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
def run(cls, kw, B, steps=5):
    torch.manual_seed(0)
    m = cls(**kw).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    x = torch.randn(B, 2, 128, 128, device="cuda")
    for i in range(2 + steps):
        if i == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        _, ld = m(x); ld["total_loss"].backward(); opt.step(); m.zero_grad()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{cls.__name__} {kw} B={B}: {dt*1e3:.2f} ms/step = {B/dt:.0f} patches/s", flush=True)
run(dynamorph_amd.VQ_VAE_z32, {}, 512, 10)
run(dynamorph_amd.VQ_VAE, {}, 512, 10)
ex = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)
run(dynamorph_amd.VQ_VAE_z32, ex, 64, 3)
run(dynamorph_amd.VQ_VAE, ex, 64, 3)
