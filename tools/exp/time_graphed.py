import sys, time, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
from dynamorph_amd.train import GraphedTrainer
def run(kw, B, graphed, steps=10):
    torch.manual_seed(0)
    m = dynamorph_amd.VQ_VAE_z32(**kw).cuda()
    x = torch.randn(B, 2, 128, 128, device="cuda")
    if graphed:
        tr = GraphedTrainer(m, lr=1e-4)
        step = lambda: tr.step(x)
    else:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        def step():
            _, ld = m(x); ld["total_loss"].backward(); opt.step(); m.zero_grad()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"z32 {kw or 'default'} B={B} {'graph' if graphed else 'eager'}: {dt*1e3:.2f} ms/step = {B/dt:.0f} patches/s", flush=True)
ex = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)
for g in (False, True):
    run({}, 512, g); run({}, 2048, g); run(ex, 256, g)
