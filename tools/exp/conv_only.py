"""Run one hot kernel of the C3 step alone, N times (for profilers): python3 tools/exp/conv_only.py {enc4|enc7|enc10|res3|tail} [N] [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd import ops, engine as E
from dynamorph_amd.ops import DM_LOAD_AFFINE_RELU, DM_LOAD_RELU, Op, weight_view
which = sys.argv[1] if len(sys.argv) > 1 else "enc7"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
torch.manual_seed(0)
model = dynamorph_amd.VQ_VAE().to("cuda:0")
x = torch.randn(B, 2, 128, 128, device="cuda:0")
Ly = E.Layers(model)
w = lambda p: p.detach()
with torch.no_grad():
    z, cx = E.encoder_forward(Ly, x)
    zq, _, _ = E.vq_forward(Ly.codebook.weight, z, float(model.commitment_cost))
    _, dcx = E.decoder_forward(Ly, zq, x, None, defer_tail=True)
nh, nrh, c1 = Ly.nh, Ly.nrh, Ly.nh // 2
H1, W1, H2, W2, H3, W3 = cx.dims
a2 = torch.empty_like(cx.a2); a3 = torch.empty_like(cx.a3); a4 = torch.empty_like(cx.a4)
sv = cx.res[0]; ca, bna, cb2, bnb = Ly.res[0]; ra = torch.empty_like(sv.ra)
var = Ly.channel_var.detach().to(x.device, torch.float32).reshape(-1).contiguous(); gs = torch.ones(1, device=x.device)
cases = {
    "enc4": lambda: ops.conv4x4s2(Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), weight_view(w(Ly.enc4.weight), c1 * 16, 16, 4, 1), B, c1, nh, H1, W1, out=a2, want_stats=True, bias=w(Ly.enc4.bias)),
    "enc7": lambda: ops.conv4x4s2(Op(cx.a2, DM_LOAD_AFFINE_RELU, cx.coef2), weight_view(w(Ly.enc7.weight), nh * 16, 16, 4, 1), B, nh, nh, H2, W2, out=a3, want_stats=True, bias=w(Ly.enc7.bias)),
    "enc10": lambda: ops.conv3x3(Op(cx.a3, DM_LOAD_AFFINE_RELU, cx.coef3), weight_view(w(Ly.enc10.weight), nh * 9, 9, 3, 1), B, nh, nh, H3, W3, taps=9, out=a4, want_stats=True, bias=w(Ly.enc10.bias)),
    "res3": lambda: ops.conv3x3(Op(sv.h_in, DM_LOAD_RELU), weight_view(w(ca.weight), nh * 9, 9, 3, 1), B, nh, nrh, H3, W3, taps=9, out=ra, want_stats=True, bias=w(ca.bias)),
    "tail": lambda: ops.dec_tail_train(dcx.d2, w(Ly.dec4.weight), w(Ly.dec4.bias), w(Ly.dec6.weight), w(Ly.dec6.bias), x, None, var, gs),
}
fn = cases[which]
for _ in range(N): fn()
torch.cuda.synchronize()
print("done", which, N)
