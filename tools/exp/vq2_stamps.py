"""Phase shares of the MFMA VQ kernel from the diagnostic library (make -C dynamorph_amd/csrc stamps):
DM_LIB_PATH=dynamorph_amd/libdynamorph_hip_stamps.so python tools/exp/vq2_stamps.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D, K, H, W = 16, 64, 16, 16
dev = "cuda:0"
z = torch.randn(B, D, H, W, device=dev); cb = torch.randn(K, D, device=dev)
lib = L.load()
wsb = lib.dm_vq_workspace_bytes(K, D)
names = ["wait for prefetched z (before stores)", "prefetch issue", "-", "MFMA + in-lane top-2", "reduce-scatter + tolerance",
         "re-checks + all-gather", "gather + out", "stores issue"]
for rep in range(3):
    idx = torch.empty(B, H, W, device=dev, dtype=torch.int64); out = torch.empty_like(z)
    slabs = torch.empty(lib.dm_vq_num_blocks(B * H * W), device=dev, dtype=torch.float64)
    hist = torch.empty(K, device=dev, dtype=torch.int32); ws = torch.empty(wsb // 4, device=dev)
    L.check(lib.dm_vq_forward_variant(z.data_ptr(), cb.data_ptr(), idx.data_ptr(), out.data_ptr(), slabs.data_ptr(),
                                      hist.data_ptr(), B, D, K, H, W, ws.data_ptr(), wsb, L.DM_VQ_MFMA,
                                      torch.cuda.current_stream().cuda_stream), "vq")
    torch.cuda.synchronize()
st = ws[4:20].view(torch.int64).cpu().tolist()
tot = sum(st)
nchunks = B * H * W // 64
print(f"B={B}: {nchunks} chunks, {tot / nchunks:.0f} stamped cycles per chunk (sum over waves / chunks)")
for n, v in zip(names, st):
    print(f"  {n:28s} {v / nchunks:9.0f} cyc/chunk  {100 * v / tot:5.1f} %")
