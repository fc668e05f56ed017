"""Backward of enc.4 at the headline batch: the two kernels (transposed-convolution data gradient + weight gradient)
against kernel D (one staging).  DM_FUSED_BWD_BLOCK=512 / 256 picks kernel D's workgroup size (read once per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = "cuda:0"
CD, CX, H, W = 16, 8, 32, 32
g = torch.Generator(dev).manual_seed(1)
dy = torch.randn(B, CD, H, W, device=dev, generator=g); a_out = torch.randn(B, CD, H, W, device=dev, generator=g)
a_in = torch.randn(B, CX, 2 * H, 2 * W, device=dev, generator=g)
cD = torch.stack([torch.randn(CD, device=dev), torch.randn(CD, device=dev) * .1, torch.randn(CD, device=dev) * .1, torch.zeros(CD, device=dev)], 1).contiguous()
cT = torch.stack([torch.rand(CX, device=dev) + .5, torch.zeros(CX, device=dev), torch.randn(CX, device=dev) * .3, torch.zeros(CX, device=dev)], 1).contiguous()
w = torch.randn(CD, CX, 4, 4, device=dev) * .2
dst = torch.empty_like(w)


def t_ms(fn0, iters=10, warm=2, launches=10):
    """average duration of one call inside a HIP graph of `launches` back-to-back calls (no host launch gaps)"""
    fn0()
    gg = torch.cuda.CUDAGraph()
    sd = torch.cuda.Stream()
    sd.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sd):
        fn0()
    torch.cuda.current_stream().wait_stream(sd)
    with torch.cuda.graph(gg):
        for _ in range(launches):
            fn0()
    fn = gg.replay
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters / launches


dyop = lambda: ops.Op(dy, 4, cD, p1=a_out)
pend = []
def pair():
    pend.clear()
    ops.wgrad(dyop(), ops.Op(a_in, 3, cT), dst, B, CD, CX, H, W, 4, pending=pend)
    ops.conv3x3(dyop(), ops.weight_view(w, 16, CX * 16, 4, 1), B, CD, 4 * CX, H, W, taps=9, pixel_shuffle=True, want_stats=True,
                mask=ops.Op(a_in, 2, cT), stat_q=a_in)
def fused():
    pend.clear()
    ops.conv_bwd_s2_fused(dyop(), ops.Op(a_in, 3, cT), ops.weight_view(w, 16, CX * 16, 4, 1), dst, B, CD, CX, H, W,
                          mask=ops.Op(a_in, 2, cT), stat_q=a_in, pending=pend)
flop = 2.0 * B * H * W * CD * CX * 16 * 2            # useful multiply-adds of both gradients
byt = B * (2 * CD * H * W + 2 * CX * 4 * H * W) * 4   # dy, a_out, a_in read once, dx written
for name, fn in (("two kernels", pair), ("kernel D  ", fused)):
    ms = t_ms(fn)
    print(f"B={B} block={os.environ.get('DM_FUSED_BWD_BLOCK', '256')} {name}: {ms * 1e3:7.1f} us  {flop / ms / 1e9:6.1f} TFLOP/s useful  "
          f"{byt / ms / 1e6:7.1f} GB/s of distinct tensors", flush=True)
