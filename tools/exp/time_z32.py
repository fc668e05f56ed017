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
ex = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which in ("all", "z32"):
    run(dynamorph_amd.VQ_VAE_z32, ex, B, 5)
if which in ("all", "vq"):
    run(dynamorph_amd.VQ_VAE, ex, B, 5)
