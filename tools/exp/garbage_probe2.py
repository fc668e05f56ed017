"""The example-width FusedTrainer against eager autograd + torch Adam, step by step, with the allocator's pool poisoned with NaNs
before the trainer captures and before every step: which parameter departs first, and by how much?"""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd.train import FusedTrainer
from garbage_probe import poison

def run(dirty, use_graph):
    kw = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)
    B = 3
    torch.manual_seed(4321)
    m1 = dynamorph_amd.VQ_VAE_z32(weight_matching=1.0, **kw).to("cuda")
    m2 = copy.deepcopy(m1)
    opt = torch.optim.Adam(m1.parameters(), lr=1e-3)
    if dirty: poison()
    tr = FusedTrainer(m2, lr=1e-3, use_graph=use_graph)
    mask = (torch.rand(B, 1, 128, 128, generator=torch.Generator().manual_seed(2)) > 0.4).float().cuda()
    for step in range(3):
        x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(10 + step)).cuda()
        tm = torch.randint(0, 3, (B, B), generator=torch.Generator().manual_seed(20 + step)).float().cuda()
        if dirty: poison()
        _, ld = m1(x, time_matching_mat=tm, batch_mask=mask)
        ld["total_loss"].backward()
        opt.step(); m1.zero_grad()
        if dirty: poison()
        vals = tr.step(x, mask, tm).tolist()
        worst = sorted(((float((a - b).abs().max()), k) for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items())
                        if a.dtype.is_floating_point), reverse=True)[:4]
        print(f"dirty {dirty} graph {use_graph} step {step}: recon {vals[0]:.7f} vs {float(ld['recon_loss']):.7f}; largest state differences {[(round(d, 7), k) for d, k in worst]}")

for dirty in (False, True):
    for use_graph in (True, False):
        run(dirty, use_graph)
