import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamorph_amd import ops, _lib
from dynamorph_amd.ops import Op, weight_view
lib = _lib.load()
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
x = torch.randn(B, 2, 128, 128, device=dev)
w1 = torch.randn(8, 3, 4, 4, device=dev) * 0.1; bias1 = torch.randn(8, device=dev)
a1 = torch.empty(B, 8, 64, 64, device=dev)
fn = lambda: ops.conv4x4s2(Op(x, ones=True), weight_view(w1, 48, 16, 4, 1), B, 3, 8, 128, 128, out=a1, want_stats=True, bias=bias1)
for _ in range(3): fn()
out = (ctypes.c_ulonglong * 8)()
lib.dm_debug_conv_stamps(out, 1)
N = 10
for _ in range(N): fn()
lib.dm_debug_conv_stamps(out, 1)
names = ["MFMA loop + epilogue (prev tile)", "barrier top", "commit (incl. vmcnt wait)", "barrier after commit", "issue next loads"]
tot = sum(out)
waves = 768 * 4; tiles = 16384 / 768
for n, v in zip(names, out):
    print(f"{n:34s} {v / N / waves / tiles:9.0f} ticks/wave/tile  {100 * v / tot:5.1f}%")
print("total", tot / N / waves / tiles)
