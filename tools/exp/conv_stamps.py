"""Phase shares of conv4x4s2_kernel from the diagnostic library (make -C dynamorph_amd/csrc stamps):
DM_LIB_PATH=dynamorph_amd/libdynamorph_hip_stamps.so python tools/exp/conv_stamps.py
Phases (s_memtime between them, summed over waves): 7 loop top | 0 barrier (previous tile consumed) | 1 wait for the
tile's loads | 2 commit | 3 barrier | 4 operand/epilogue-load setup | 5 MFMA loop | 6 epilogue."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd import ops, engine as E, _lib as L
from dynamorph_amd.ops import DM_LOAD_AFFINE_RELU, Op, weight_view
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
model = dynamorph_amd.VQ_VAE().to("cuda:0")
x = torch.randn(B, 2, 128, 128, device="cuda:0")
Ly = E.Layers(model)
w = lambda p: p.detach()
with torch.no_grad():
    z, cx = E.encoder_forward(Ly, x)
nh, c1 = Ly.nh, Ly.nh // 2
H1, W1, H2, W2, H3, W3 = cx.dims
a2 = torch.empty_like(cx.a2); a3 = torch.empty_like(cx.a3)
lib = ctypes.CDLL(os.environ["DM_LIB_PATH"])
names = {0: "barrier: previous tile consumed", 1: "wait for the tile's loads", 2: "commit", 3: "barrier: tile visible",
         4: "tile setup + epilogue loads", 5: "MFMA loop", 6: "epilogue (bias, stats, stores)", 7: "loop top"}
cases = {
    "enc.4 (8->16, 64x64 -> 32x32)": lambda: ops.conv4x4s2(Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), weight_view(w(Ly.enc4.weight), c1 * 16, 16, 4, 1),
                                                       B, c1, nh, H1, W1, out=a2, want_stats=True, bias=w(Ly.enc4.bias)),
    "enc.7 (16->16, 32x32 -> 16x16)": lambda: ops.conv4x4s2(Op(cx.a2, DM_LOAD_AFFINE_RELU, cx.coef2), weight_view(w(Ly.enc7.weight), nh * 16, 16, 4, 1),
                                                        B, nh, nh, H2, W2, out=a3, want_stats=True, bias=w(Ly.enc7.bias)),
}
buf = (ctypes.c_ulonglong * 8)()
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.dm_conv_stamps_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.dm_conv_stamps_read(buf, 1)
    st = list(buf); tot = sum(st)
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:.1f} us (one launch, stamped build); {tot} stamped wave-cycles")
    order = [7, 0, 1, 2, 3, 4, 5, 6]
    for i in order:
        print(f"   {names[i]:36s} {100 * st[i] / tot:5.1f} %")
