import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops
dev = "cuda:0"; B = 1024
def seg(C, spg):
    stats = torch.rand(B * spg, C, 2, device=dev, dtype=torch.float64) + 1.0
    return (stats, spg, 4096 // spg, torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.int64, device=dev), 0.1)
cfgs = {"one C=8 spg=8": [(8, 8)], "one C=16 spg=1": [(16, 1)], "one C=32 spg=1": [(32, 1)],
        "eight layers": [(8, 8), (16, 4), (16, 2), (16, 1), (32, 1), (16, 2), (32, 1), (16, 2)]}
for name, layers in cfgs.items():
    segs = [seg(C, spg) for C, spg in layers]
    for _ in range(3): ops.bn_running_replay(list(segs))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.bn_running_replay(list(segs))
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch")
