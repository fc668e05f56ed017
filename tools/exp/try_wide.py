import sys, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
from oracle import vqvae_oracle as O
for kw in (dict(num_hiddens=128, num_residual_hiddens=16, num_embeddings=32), dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)):
    torch.manual_seed(31 + kw["num_hiddens"])
    ref = O.OracleVQVAE(**kw).double()
    m = dynamorph_amd.VQ_VAE(**kw).cuda()
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    ref.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    x = torch.randn(2, 2, 128, 128, generator=torch.Generator().manual_seed(8))
    mask = (torch.rand(2, 1, 128, 128, generator=torch.Generator().manual_seed(9)) > 0.3).float()
    _, ld_r = ref(x.double(), batch_mask=mask.double()); ld_r["total_loss"].backward()
    _, ld = m(x.cuda(), batch_mask=mask.cuda()); ld["total_loss"].backward()
    gr = dict(ref.named_parameters())
    print(kw)
    for k, p in m.named_parameters():
        if not p.requires_grad or gr[k].grad is None: continue
        b = gr[k].grad
        sc = max(b.abs().max().item(), 1e-12)
        e = (p.grad.cpu().double() - b).abs().max().item() / sc
        if e > 2e-5 and sc > 1e-9: print(f"  {k:28s} scale {sc:9.3e}  hip-f64 {e:9.2e}")
