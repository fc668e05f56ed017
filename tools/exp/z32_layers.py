"""The example configuration's layers (VQ_VAE_z32, 64 / 64 channels, 128-pixel patches, B = 768) one at a time, with the
operand modes the training step uses.  DM_ONLY=<substring> selects; DM_B overrides the batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import _lib
if os.environ.get("DM_LIB"):
    _lib.LIB_PATH = os.environ["DM_LIB"]
from dynamorph_amd import ops
ONLY = os.environ.get("DM_ONLY", "")
B = int(os.environ.get("DM_B", "768"))
C, HW, NIN = 64, 32, 2
dev = "cuda"
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)
def coef(c):
    return torch.stack([r(c).abs() + 0.5, r(c) * 0.1, r(c) * 0.1, torch.zeros(c, device=dev)], 1).contiguous()
x32, g32, a32 = r(B, C, HW, HW), r(B, C, HW, HW), r(B, C, HW, HW)
x64, g64 = r(B, C // 2, 64, 64), r(B, C // 2, 64, 64)
x128, g128 = r(B, NIN, 128, 128), r(B, NIN, 128, 128)
w3, w1 = r(C, C, 3, 3) * 0.05, r(C, C, 1, 1) * 0.1
w4 = r(C, C // 2, 4, 4) * 0.05
w0 = r(C // 2, NIN, 4, 4) * 0.1
wu1 = r(C // 2, NIN, 4, 4) * 0.1
cf64, cf32 = coef(C), coef(C // 2)
d33, d11, d44, d0 = torch.empty_like(w3), torch.empty_like(w1), torch.empty_like(w4), torch.empty_like(w0)
Op = ops.Op
def t(name, fn, flops, nbytes, n=10):
    if ONLY and ONLY not in name: return
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:34s} {dt*1e6:8.1f} us  {flops/dt/1e12:6.1f} TFLOP/s  {nbytes/dt/1e12:5.2f} TB/s", flush=True)
P = B * HW * HW
T32, T64, T128 = x32.numel() * 4, x64.numel() * 4, x128.numel() * 4
t("res conv3x3 fwd", lambda: ops.conv3x3(Op(x32, 1), ops.weight_view(w3, C * 9, 9, 3, 1), B, C, C, HW, HW, taps=9, want_stats=True), 2 * P * C * C * 9, 2 * T32)
t("res conv1x1 fwd", lambda: ops.conv3x3(Op(x32, 3, cf64), ops.weight_view(w1, C, 1, 0, 0), B, C, C, HW, HW, taps=1, want_stats=True), 2 * P * C * C, 2 * T32)
t("res conv1x1 dgrad", lambda: ops.conv3x3(Op(g32, 4, cf64, p1=a32), ops.weight_view(w1, 1, C, 0, 0), B, C, C, HW, HW, taps=1, want_stats=True,
                                          like=g32, mask=Op(x32, 2, cf64), stat_q=x32), 2 * P * C * C, 4 * T32)
t("res conv3x3 dgrad", lambda: ops.conv3x3(Op(g32, 4, cf64, p1=a32), ops.weight_view(w3, 9, C * 9, -3, -1, off=8), B, C, C, HW, HW, taps=9,
                                          want_stats=True, like=g32, mask=Op(x32), resid=a32, stat_q=x32), 2 * P * C * C * 9, 5 * T32)
t("res wgrad 1x1", lambda: ops.wgrad(Op(g32, 4, cf64, p1=a32), Op(x32, 3, cf64), d11, B, C, C, HW, HW, 1), 2 * P * C * C, 3 * T32)
t("res wgrad 3x3", lambda: ops.wgrad(Op(g32, 4, cf64, p1=a32), Op(x32, 1), d33, B, C, C, HW, HW, 3), 2 * P * C * C * 9, 3 * T32)
t("conv0 fwd 2->32 s2", lambda: ops.conv4x4s2(Op(x128), ops.weight_view(w0, NIN * 16, 16, 4, 1), B, NIN, C // 2, 128, 128, want_stats=True), 2 * B * 64 * 64 * 32 * NIN * 16, T128 + T64)
t("conv1 fwd 32->64 s2", lambda: ops.conv4x4s2(Op(x64, 3, cf32), ops.weight_view(w4, (C // 2) * 16, 16, 4, 1), B, C // 2, C, 64, 64, want_stats=True), 2 * P * C * (C // 2) * 16, T64 + T32)
t("up0 fwd convT 64->32", lambda: ops.conv3x3(Op(x32), ops.weight_view(w4, 16, (C // 2) * 16, 4, 1), B, C, 4 * (C // 2), HW, HW, taps=9, pixel_shuffle=True, want_stats=True), 2 * P * C * (C // 2) * 16, T32 + T64)
t("up1 fwd convT 32->2", lambda: ops.conv3x3(Op(x64, 3, cf32), ops.weight_view(wu1, 16, NIN * 16, 4, 1), B, C // 2, 4 * NIN, 64, 64, taps=9, pixel_shuffle=True), 2 * B * 64 * 64 * 32 * NIN * 16, T64 + T128)
t("wgrad conv0 (S 32ch aff2, T x)", lambda: ops.wgrad(Op(g64, 4, cf32, p1=x64), Op(x128), d0, B, C // 2, NIN, 64, 64, 4), 2 * B * 64 * 64 * 32 * NIN * 16, 2 * T64 + T128)
t("wgrad up1 (S d1 aff-relu, T g)", lambda: ops.wgrad(Op(x64, 3, cf32), Op(g128), d0, B, C // 2, NIN, 64, 64, 4), 2 * B * 64 * 64 * 32 * NIN * 16, T64 + T128)
t("wgrad conv1 (S 64 aff2, T 32ch)", lambda: ops.wgrad(Op(g32, 4, cf64, p1=a32), Op(x64, 3, cf32), d44, B, C, C // 2, HW, HW, 4), 2 * P * C * (C // 2) * 16, 2 * T32 + T64)
t("dgrad up1 (conv s2 2->32)", lambda: ops.conv4x4s2(Op(g128), ops.weight_view(wu1, 16, NIN * 16, 4, 1), B, NIN, C // 2, 128, 128, want_stats=True,
                                                  mask=Op(x64, 2, cf32), stat_q=x64), 2 * B * 64 * 64 * 32 * NIN * 16, T128 + 2 * T64)
t("dgrad conv1 (convT 64->32 aff2)", lambda: ops.conv3x3(Op(g32, 4, cf64, p1=a32), ops.weight_view(w4, 16, (C // 2) * 16, 4, 1), B, C, 4 * (C // 2), HW, HW, taps=9,
                                                        pixel_shuffle=True, want_stats=True, mask=Op(x64, 2, cf32), stat_q=x64), 2 * P * C * (C // 2) * 16, 2 * T32 + 2 * T64)
t("dgrad up0 (conv s2 32->64)", lambda: ops.conv4x4s2(Op(x64), ops.weight_view(w4, (C // 2) * 16, 16, 4, 1), B, C // 2, C, 64, 64), 2 * P * C * (C // 2) * 16, T64 + T32)
t("res 1x1 backward fused", lambda: ops.conv1x1_bwd_fused(Op(g32, 4, cf64, p1=a32), x32, cf64, w1, d11, B, C, C, HW, HW), 4 * P * C * C, 4 * T32)
