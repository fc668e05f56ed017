import sys, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
from oracle import vqvae_oracle as O
for kw in (dict(num_hiddens=16, num_residual_hiddens=32), dict(num_hiddens=32, num_residual_hiddens=32)):
    torch.manual_seed(31 + kw["num_hiddens"])
    ref = O.OracleVQVAE(**kw).double()
    ref32 = O.OracleVQVAE(**kw)
    ref32.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    m = dynamorph_amd.VQ_VAE(**kw).cuda()
    m.load_state_dict(ref32.state_dict())
    x = torch.randn(2, 2, 128, 128, generator=torch.Generator().manual_seed(8))
    _, ld_r = ref(x.double()); ld_r["total_loss"].backward()
    _, ld_s = ref32(x); ld_s["total_loss"].backward()
    _, ld = m(x.cuda()); ld["total_loss"].backward()
    gr, gs = dict(ref.named_parameters()), dict(ref32.named_parameters())
    print(kw)
    for k, p in m.named_parameters():
        if not p.requires_grad or gr[k].grad is None: continue
        b = gr[k].grad
        sc = max(b.abs().max().item(), 1e-12)
        print(f"  {k:28s} scale {sc:9.3e}  hip-f64 {(p.grad.cpu().double() - b).abs().max().item() / sc:9.2e}   oracle32-f64 {(gs[k].grad.double() - b).abs().max().item() / sc:9.2e}")
