#!/usr/bin/env python3
"""Kernel micro-bench at BASELINE C3 shapes (B=2048): single launches timed with events, algorithmic GB/s.

    gpurun -- python tools/kbench.py            # all kernels below
    KB_B=256 python tools/kbench.py tail        # subset by name prefix, smaller batch
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamorph_amd import ops
from dynamorph_amd.ops import Op, weight_view


def t_ms(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


dev = "cuda:0"; B = int(os.environ.get("KB_B", "2048"))
want = sys.argv[1:] or [""]
torch.manual_seed(0)
x = torch.randn(B, 2, 128, 128, device=dev)
cases = {}

w1 = torch.randn(8, 3, 4, 4, device=dev) * 0.1; bias1 = torch.randn(8, device=dev)
a1 = torch.empty(B, 8, 64, 64, device=dev)
border1 = torch.randn(3, 3, 8, device=dev)
cases["e1_conv"] = (lambda: ops.conv4x4s2(Op(x), weight_view(w1, 48, 16, 4, 1), B, 2, 8, 128, 128,
                                          out=a1, want_stats=True, bias_border=border1), B * (2 * 128 * 128 + 8 * 64 * 64) * 4)

d2 = torch.randn(B, 4, 64, 64, device=dev).clamp(min=0)
w4 = torch.randn(4, 4, 4, 4, device=dev) * .3; b4 = torch.randn(4, device=dev)
w6 = torch.randn(2, 4, 1, 1, device=dev); b6 = torch.randn(2, device=dev)
var = torch.ones(2, device=dev); gs = torch.ones(1, device=dev)
dec, _ = ops.dec_tail_forward(d2, w4, b4, w6, b6, x, None, var)
cases["tail_fwd"] = (lambda: ops.dec_tail_forward(d2, w4, b4, w6, b6, x, None, var), B * 327680)
cases["tail_bwd"] = (lambda: ops.dec_tail_backward(d2, w4, b4, w6, dec, x, None, var, gs), B * 393216)
cases["tail_train"] = (lambda: ops.dec_tail_train(d2, w4, b4, w6, b6, x, None, var, gs), B * 262144)

zq = torch.randn(B, 16, 16, 16, device=dev); cbk = torch.randn(64, 16, device=dev)
vq_bufs = ops.vq_forward_repeat(zq, cbk, 1)
cases["vq_fwd"] = (lambda: ops.vq_forward_repeat(zq, cbk, 1, bufs=vq_bufs), B * 34816)      # prep + distance kernel + counter reduction

rbk, hk = torch.randn(B, 16, 16, 16, device=dev), torch.randn(B, 16, 16, 16, device=dev)
coefk = torch.rand(16, 4, device=dev) + 0.5
cases["vq_join"] = (lambda: ops.vq_forward_join(rbk, hk, coefk, cbk), B * (4 * 16384 + 2048))     # rb, h_in read; z, out, idx written

# train()'s resident feed: a batch gathered from a dataset in HBM with per-sample flip / rot90 (read + write of every patch)
srcg = torch.randn(B + B // 2, 2, 128, 128, device=dev)
idsg = torch.randperm(srcg.shape[0], device=dev)[:B].to(torch.int32)
flg = torch.randint(0, 3, (B,), device=dev, dtype=torch.int32); rog = torch.randint(0, 4, (B,), device=dev, dtype=torch.int32)
outg = torch.empty(B, 2, 128, 128, device=dev)
cases["gather_aug"] = (lambda: ops.gather_augment(srcg, idsg, flg, rog, outg), B * 2 * 131072)

# the fused backward kernels of round 4 at the C3 shapes (one staging, data + weight gradient)
gh, rbk2 = (torch.randn(B, 16, 16, 16, device=dev) for _ in range(2))
rak = torch.randn(B, 32, 16, 16, device=dev)
cdk, cxk32 = torch.randn(16, 4, device=dev) * 0.5, torch.rand(32, 4, device=dev) + 0.5
w11 = torch.randn(16, 32, 1, 1, device=dev) * 0.2; g11 = torch.zeros_like(w11)
cases["bwd_1x1"] = (lambda: ops.conv1x1_bwd_fused(Op(gh, 4, cdk, p1=rbk2), rak, cxk32, w11, g11, B, 16, 32, 16, 16),
                    B * (2 * 16384 + 32768 + 32768))
dyr, rar = (torch.randn(B, 32, 16, 16, device=dev) for _ in range(2))
hin, res_, qq = (torch.randn(B, 16, 16, 16, device=dev) for _ in range(3))
cd32 = torch.randn(32, 4, device=dev) * 0.5
w33 = torch.randn(32, 16, 3, 3, device=dev) * 0.2; g33 = torch.zeros_like(w33)
cases["bwd_3x3_res"] = (lambda: ops.conv3x3_bwd_fused(Op(dyr, 4, cd32, p1=rar), hin, None, w33, g33, B, 32, resid=res_, q=qq),
                        B * (2 * 32768 + 4 * 16384))
a2k = torch.randn(B, 16, 32, 32, device=dev); cx16 = torch.rand(16, 4, device=dev) + 0.5
w44 = torch.randn(16, 16, 4, 4, device=dev) * 0.2; g44 = torch.zeros_like(w44)
cases["bwd_4x4s2"] = (lambda: ops.conv4x4s2_bwd_fused(Op(gh, 4, cdk, p1=rbk2), a2k, cx16, w44, g44, B), B * (2 * 16384 + 2 * 65536))
d0k = torch.randn(B, 8, 32, 32, device=dev).clamp(min=0); g2k = torch.randn(B, 4, 64, 64, device=dev)
wT2 = torch.randn(8, 4, 4, 4, device=dev) * 0.2; gT2 = torch.zeros_like(wT2)
cases["bwd_convT_dec2"] = (lambda: ops.convT_bwd_fused(d0k, g2k, wT2, gT2, mask_relu=True, want_stats=True), B * (65536 + 2 * 32768))
g0k = torch.randn(B, 8, 32, 32, device=dev); wT0 = torch.randn(16, 8, 4, 4, device=dev) * 0.2; gT0 = torch.zeros_like(wT0)
cases["bwd_convT_dec0"] = (lambda: ops.convT_bwd_fused(zq, g0k, wT0, gT0), B * (32768 + 2 * 16384))

# stress shape of BASELINE configs[4]: 4096 codes, 32 x 32 latents of KB_B5 patches (bench.py --workload c5 default batch)
B5 = int(os.environ.get("KB_B5", "1024"))
zq5 = torch.randn(B5, 16, 32, 32, device=dev); cbk5 = torch.randn(4096, 16, device=dev)
vq5_bufs = ops.vq_forward_repeat(zq5, cbk5, 1)
cases["vq_fwd_k4096"] = (lambda: ops.vq_forward_repeat(zq5, cbk5, 1, bufs=vq5_bufs), B5 * 1024 * 136)

vq_idx = ops.vq_forward(zq, cbk, want_out=False)[0]
gq = torch.randn_like(zq)
cases["vq_bwd"] = (lambda: ops.vq_backward_slabs(zq, cbk, vq_idx, gq, None, 0.25), B * 256 * (3 * 64 + 8))  # z, g_out, dz, idx
cases["vq_bwd_atomic"] = (lambda: ops.vq_backward(zq, cbk, vq_idx, gq, None, 0.25, dw=torch.zeros_like(cbk)), B * 256 * (3 * 64 + 8))

src = torch.randn(64 << 20, device=dev); dst = torch.empty_like(src)          # 256 MiB each: past the Infinity Cache
cases["copy_calib"] = (lambda: torch.add(src, 1.0, out=dst), 2 * src.numel() * 4)            # known bytes: calibrates the PMC counters

for name, (fn, nbytes) in cases.items():
    if not any(name.startswith(p) for p in want):
        continue
    ms = t_ms(fn)
    print(f"{name:10s} {ms * 1e3:8.1f} us   {nbytes / ms / 1e6:8.1f} GB/s (algorithmic)")
