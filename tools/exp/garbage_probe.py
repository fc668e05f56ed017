"""Does any kernel of a training step read memory it (or its producer) did not write?  The caching allocator hands out blocks
that earlier work has used: this probe fills the pool with NaNs before every step, so that an unwritten slab / statistics row /
scratch word that reaches a result shows up as a NaN (or as a difference from the clean run)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd.train import FusedTrainer

def poison(mb=2048):
    blocks = []
    for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        cnt = max(1, min(64, (mb << 20) // (7 * n * 4)))
        blocks += [torch.full((n,), float("nan"), device="cuda") for _ in range(cnt)]
    del blocks          # back to the caching allocator, contents intact

def grads_of(family, kw, B, dirty, with_tm=True):
    torch.manual_seed(4321)
    m = getattr(dynamorph_amd, family)(**kw).to("cuda")
    x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(10)).cuda()
    mask = (torch.rand(B, 1, 128, 128, generator=torch.Generator().manual_seed(2)) > 0.4).float().cuda()
    tm = torch.randint(0, 3, (B, B), generator=torch.Generator().manual_seed(20)).float().cuda() if with_tm else None
    if dirty:
        poison()
    args = dict(batch_mask=mask)
    if tm is not None and family != "VQ_VAE":
        args["time_matching_mat"] = tm
    _, ld = m(x, **args)
    ld["total_loss"].backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, {k: float(v) for k, v in ld.items()}

if __name__ == "__main__":
    for family, kw, B in (("VQ_VAE_z32", dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512), 3),
                          ("VQ_VAE_z32", {}, 5), ("VQ_VAE", {}, 6),
                          ("VQ_VAE", dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512), 3)):
        clean, lc = grads_of(family, kw, B, False)
        dirty, ld = grads_of(family, kw, B, True)
        bad = []
        for k in clean:
            d = (clean[k] - dirty[k]).abs().max().item()
            if not (d == 0.0):
                bad.append((k, d, float(clean[k].abs().max())))
        print(family, kw.get("num_hiddens", 16), "B", B, "losses equal:", lc == ld, "gradients that differ under a poisoned pool:", bad[:8], len(bad))
