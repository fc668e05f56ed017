import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dynamorph_amd import ops, _lib
from dynamorph_amd.ops import Op
lib = _lib.load()
dev = "cuda:0"; B = 2048
torch.manual_seed(0)
x = torch.randn(B, 2, 128, 128, device=dev)
dy = torch.randn(B, 8, 64, 64, device=dev); a1 = torch.randn(B, 8, 64, 64, device=dev)
coef = torch.randn(8, 4, device=dev)
dw = torch.empty(8, 3, 4, 4, device=dev)
def run(two):
    S = Op(dy, ops.DM_LOAD_AFFINE2, coef, a1) if two else Op(dy)
    return lambda: ops.wgrad(S, Op(x, ones=True), dw, B, 8, 3, 64, 64, 4)
for two in (False, True):
    try:
        fn = run(two)
        for _ in range(3): fn()
    except Exception as e:
        print("skip", two, e); continue
    out = (ctypes.c_ulonglong * 8)()
    lib.dm_debug_wg_stamps(out, 1)
    N = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); e1.synchronize()
    lib.dm_debug_wg_stamps(out, 1)
    names = ["MFMA loop (prev tile)", "barrier top", "commit S", "commit T", "barrier after commit", "issue next loads"]
    tot = sum(out)
    print("AFFINE2 S operand" if two else "IDENT S operand", f"{e0.elapsed_time(e1)/N*1e3:.1f} us per launch")
    for n, v in zip(names, out):
        print(f"  {n:26s} {v / N / 2048 / 32:9.0f} ticks/wave/tile  {100 * v / tot:5.1f}%")
