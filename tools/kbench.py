#!/usr/bin/env python3
"""Kernel micro-bench at BASELINE C3 shapes (B=2048): times single launches with events, prints GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamorph_amd import ops
from dynamorph_amd.ops import Op, weight_view

def t_ms(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

dev = "cuda:0"; B = int(os.environ.get("KB_B", "2048"))
torch.manual_seed(0)
x = torch.randn(B, 2, 128, 128, device=dev)
w = torch.randn(8, 3, 4, 4, device=dev) * 0.1
bias = torch.randn(8, device=dev)
a1 = torch.empty(B, 8, 64, 64, device=dev)
def e1():
    ops.conv4x4s2(Op(x, ones=True), weight_view(w, 48, 16, 4, 1), B, 3, 8, 128, 128, out=a1, want_stats=True, bias=bias)
for dbg in (0,):
    os.environ["DM_CONV_DBG"] = str(dbg)
    ms = t_ms(e1)
    print(f"E1 conv dbg={dbg} (1=noMFMA 2=noStore 4=noLoad): {ms*1e3:8.1f} us   {B*(2*128*128+8*64*64)*4/ms/1e6:8.1f} GB/s")
os.environ["DM_CONV_DBG"] = "0"
