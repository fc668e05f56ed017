import sys, torch
sys.path.insert(0, "/root/repo")
import dynamorph_amd
from dynamorph_amd import engine as E, ops
kw = dict(num_hiddens=128, num_residual_hiddens=16, num_embeddings=32)
torch.manual_seed(159)
m = dynamorph_amd.VQ_VAE(**kw).cuda()
x = torch.randn(2, 2, 128, 128, generator=torch.Generator().manual_seed(8)).cuda()
L = E.Layers(m)
z, cx = E.encoder_forward(L, x)
out = {"z": z}
for k in ("a1", "a2", "a3", "a4", "coef1", "coef2", "coef3"):
    out[k] = getattr(cx, k)
for i, sv in enumerate((cx.saved1, cx.saved2, cx.saved3, cx.saved4)):
    out[f"saved{i+1}"] = sv
g_z = torch.randn(z.shape, generator=torch.Generator().manual_seed(9)).cuda()
grads = {}
def G(p):
    if id(p) not in grads: grads[id(p)] = torch.zeros_like(p)
    return grads[id(p)]
E.encoder_backward(L, cx, g_z, G)
for n, p in m.named_parameters():
    if id(p) in grads: out["grad/" + n] = grads[id(p)]
torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}, sys.argv[1])
