import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamorph_amd import ops, _lib
lib = _lib.load()
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
x = torch.randn(B, 2, 128, 128, device=dev)
d2 = torch.randn(B, 4, 64, 64, device=dev).clamp(min=0)
w4 = torch.randn(4, 4, 4, 4, device=dev) * .3; b4 = torch.randn(4, device=dev)
w6 = torch.randn(2, 4, 1, 1, device=dev); b6 = torch.randn(2, device=dev)
var = torch.ones(2, device=dev); gs = torch.ones(1, device=dev)
dec, _ = ops.dec_tail_forward(d2, w4, b4, w6, b6, x, None, var)
for _ in range(3): ops.dec_tail_backward(d2, w4, b4, w6, dec, x, None, var, gs)
out = (ctypes.c_ulonglong * 8)()
lib.dm_debug_tail_stamps(out, 1)
N = 10
for _ in range(N): ops.dec_tail_backward(d2, w4, b4, w6, dec, x, None, var, gs)
lib.dm_debug_tail_stamps(out, 1)
names = ["wgrad(prev)+barrier top", "commit", "barrier after commit", "issue next + phase A", "barrier after A", "phase B + issue rows", "barrier after B", "phase 3 dgrad"]
tot = sum(out)
for n, v in zip(names, out):
    print(f"{n:28s} {v / N / 2048 / 32:9.0f} ticks/wave/tile  {100 * v / tot:5.1f}%")
print("total ticks/wave/tile", tot / N / 2048 / 32, "(s_memtime ticks)")
