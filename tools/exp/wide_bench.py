"""Micro-bench of the implicit-GEMM kernels on the example-config shapes (z32, num_hiddens 64, 32x32 latent, B 256)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import os
from dynamorph_amd import _lib
if os.environ.get("DM_LIB"):
    _lib.LIB_PATH = os.environ["DM_LIB"]
from dynamorph_amd import ops
ONLY = os.environ.get("DM_ONLY", "")
B, C, HW = 256, 64, 32
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, C, HW, HW, device=dev)
g = torch.randn(B, C, HW, HW, device=dev)
w3 = torch.randn(C, C, 3, 3, device=dev) * 0.05
w1 = torch.randn(C, C, 1, 1, device=dev) * 0.1
w4 = torch.randn(C, C // 2, 4, 4, device=dev) * 0.05     # convT 64 -> 32
w4s = torch.randn(C, C // 2, 4, 4, device=dev) * 0.05    # conv 4x4/s2 32 -> 64 on 64x64
x64 = torch.randn(B, C // 2, 64, 64, device=dev)
dst3 = torch.empty(C, C, 3, 3, device=dev)
dst4 = torch.empty(C, C // 2, 4, 4, device=dev)
def t(name, fn, flops, n=5):
    if ONLY and ONLY not in name: return
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:28s} {dt*1e6:8.1f} us  {flops/dt/1e12:6.1f} TFLOP/s", flush=True)
P = B * HW * HW
t("conv3x3 64->64", lambda: ops.conv3x3(ops.Op(x, 1), ops.weight_view(w3, C * 9, 9, 3, 1), B, C, C, HW, HW, taps=9, want_stats=True), 2 * P * C * C * 9)
t("conv1x1 64->64", lambda: ops.conv3x3(ops.Op(x, 1), ops.weight_view(w1, C, 1, 0, 0), B, C, C, HW, HW, taps=1, want_stats=True), 2 * P * C * C)
t("convT 64->32", lambda: ops.conv3x3(ops.Op(x), ops.weight_view(w4, 16, (C // 2) * 16, 4, 1), B, C, 4 * (C // 2), HW, HW, taps=9, pixel_shuffle=True, want_stats=True), 2 * P * C * (C // 2) * 16)
t("conv4x4s2 32->64", lambda: ops.conv4x4s2(ops.Op(x64), ops.weight_view(w4s, (C // 2) * 16, 16, 4, 1), B, C // 2, C, 64, 64, want_stats=True), 2 * P * C * (C // 2) * 16)
t("wgrad 3x3 64x64", lambda: ops.wgrad(ops.Op(g), ops.Op(x, 1), dst3, B, C, C, HW, HW, 3), 2 * P * C * C * 9)
t("wgrad 4x4 64x32", lambda: ops.wgrad(ops.Op(x), ops.Op(x64), dst4, B, C, C // 2, HW, HW, 4), 2 * P * C * (C // 2) * 16)
