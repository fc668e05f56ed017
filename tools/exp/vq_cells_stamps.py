"""Phase shares of vq_cells_kernel from the diagnostic library:
    cd dynamorph_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVQ2_STAMPS -c vq.hip -o build_measure/vq_st.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdm_vqst.so build_measure/vq_st.o $(ls *.o | grep -v "^vq.o")
    DM_LIB_PATH=$PWD/dynamorph_amd/libdm_vqst.so python tools/exp/vq_cells_stamps.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops, _lib as L
B, D, K, H, W = 1024, 16, 4096, 32, 32
dev = "cuda:0"
z = torch.randn(B, D, H, W, device=dev) * 1.7
cb = torch.randn(K, D, device=dev)
lib = L.load()
wsb = lib.dm_vq_workspace_bytes(K, D)
names = ["latents + operand split", "code stream", "drain + last group end", "owned latents, merge, best cells exactly", "exact re-checks",
         "gather, value, stores, counters"]
for rep in range(3):
    idx = torch.empty(B, H, W, device=dev, dtype=torch.int64); out = torch.empty_like(z)
    slabs = torch.empty(lib.dm_vq_num_blocks(B * H * W), device=dev, dtype=torch.float64)
    hist = torch.empty(K, device=dev, dtype=torch.int32); ws = torch.empty(wsb // 4, device=dev)
    L.check(lib.dm_vq_forward_variant(z.data_ptr(), cb.data_ptr(), idx.data_ptr(), out.data_ptr(), slabs.data_ptr(),
                                      hist.data_ptr(), B, D, K, H, W, ws.data_ptr(), wsb, L.DM_VQ_BF16,
                                      torch.cuda.current_stream().cuda_stream), "vq")
    torch.cuda.synchronize()
st = ws[4:20].view(torch.int64).cpu().tolist()
tot = sum(st)
npass = B * H * W // 128
print(f"{npass} passes of 128 positions, {tot / npass:.0f} stamped cycles per pass (s_memtime ticks, sum over waves / passes)")
tot = sum(st[:6])
for n, v in zip(names, st):
    print(f"  {n:44s} {v / npass:9.0f} ticks/pass  {100 * v / tot:5.1f} %")
print(f"  positions re-checked {st[7]} ({st[7] / (npass * 128):.2%}), groups of 128 codes visited per re-checked position {st[6] / max(st[7], 1):.2f}")
