import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from dynamorph_amd import ops
torch.manual_seed(0)
B, cs, ct, hs, ws = int(os.environ.get("DM_B", 2)), 32, 2, 64, 64
dy = torch.randn(B, cs, hs, ws); t = torch.randn(B, ct, 2 * hs, 2 * ws)
w = torch.zeros(cs, ct, 4, 4, requires_grad=True, dtype=torch.float64)
F.conv2d(t.double(), w, None, stride=2, padding=1).backward(dy.double())
dst = torch.empty(cs, ct, 4, 4, device="cuda")
ops.wgrad(ops.Op(dy.cuda()), ops.Op(t.cuda()), dst, B, cs, ct, hs, ws, 4)
err = (dst.cpu().double() - w.grad).abs()
print("max err", err.max().item(), "ref max", w.grad.abs().max().item())
print("err by (ct, ky, kx), max over cs:")
print(err.amax(0))
# which single terms would explain the error of tap (ct, ky, kx) for cs = 0?
e0 = (dst.cpu().double() - w.grad)[0]
print("signed error cs=0:", e0)
tf = t.double()
for c in range(ct):
    flat = tf[:, c].reshape(B, -1)
    # (a) column -1 not zeroed for tap (1, 0): the element before row 2y's first
    idx = (torch.arange(hs) * 2) * (2 * ws) - 1
    prev = torch.where(idx >= 0, flat[:, idx.clamp(min=0)], torch.zeros(()).double())
    if c > 0:
        prev[:, 0] = tf[:, c - 1].reshape(B, -1)[:, -1]
    cand_a = (dy[:, 0, :, 0].double() * prev).sum().item()
    # (b) the real column 2x-1 terms of x = 16 h' + 4 kq (kq = 0 lanes' first K step) missing everywhere
    cols = torch.arange(0, ws, 16)
    cand_b = sum((dy[:, 0, :, x].double() * tf[:, c, 0::2, 2 * x - 1]).sum().item() for x in cols.tolist() if x > 0)
    print(f"ct {c}: observed {e0[c, 1, 0].item():+.4f}  (a) extra column -1: {cand_a:+.4f}  (b) -missing first K step of kq = 0: {-cand_b:+.4f}")
