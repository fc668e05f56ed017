#!/usr/bin/env python3
"""Time-matching term micro-bench (run_training.py always has it on): the fused MFMA op (dm_time_matching_forward/backward) vs
the VALU pair kernels + torch expressions of round 1, at the reference's example batch (config_example.yml: batch 768) on the
z16 latent (n = 4096) and the 65 536-wide latent of the example z32 widths.   gpurun -- python tools/tmbench.py"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamorph_amd import ops


def t_ms(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


dev = "cuda:0"
out = []
for B, n in ((768, 4096), (2048, 4096), (768, 65536), (128, 4096)):
    z = torch.randn(B, n, device=dev)
    tm = torch.randint(0, 3, (B, B), device=dev).float()
    loss, S = ops.time_matching_forward(z, tm, 1, 1.1, 0.1, -0.5, 0.5)
    fw = t_ms(lambda: ops.time_matching_forward(z, tm, 1, 1.1, 0.1, -0.5, 0.5))
    bw = t_ms(lambda: ops.time_matching_backward(z, S, None, 0.005))
    fl = 2.0 * B * B * n
    rec = {"B": B, "n": n, "fused_forward_ms": round(fw, 4), "fused_backward_ms": round(bw, 4),
           "forward_tflops": round(fl / fw / 1e9, 1), "backward_tflops": round(fl / bw / 1e9, 1)}
    if B * B * n <= 768 * 768 * 4096:
        sim = ops.pair_msd(z)
        rec["round1_pair_msd_ms"] = round(t_ms(lambda: ops.pair_msd(z), iters=3, warm=1), 4)
        rec["round1_pair_msd_backward_ms"] = round(t_ms(lambda: ops.pair_msd_backward(z, sim), iters=3, warm=1), 4)
    print(json.dumps(rec), flush=True)

# mode 0 (vq_vae.py:331) on a batch's relation matrix -- trajectories of 8 consecutive frames, as train() feeds it -- : the sparse
# form (related pairs from differences, no GEMM) against the dense form of the same call
for B, n in ((2048, 4096), (768, 4096)):
    z = torch.randn(B, n, device=dev)
    i = torch.arange(B, device=dev)
    same = (i[:, None] // 8) == (i[None, :] // 8)
    tm = torch.where(same & (i[:, None] != i[None, :]), torch.where((i[:, None] - i[None, :]).abs() == 1, 2.0, 1.0), 0.0).float()
    rec = {"B": B, "n": n, "mode": 0, "relation_entries_per_row": round(float((tm != 0).sum()) / B, 2)}
    for name, sparse in (("sparse", True), ("dense", False)):
        loss, S = ops.time_matching_forward(z, tm, 0, allow_sparse=sparse)
        rec[name + "_forward_ms"] = round(t_ms(lambda: ops.time_matching_forward(z, tm, 0, allow_sparse=sparse)), 4)
        rec[name + "_backward_ms"] = round(t_ms(lambda: ops.time_matching_backward(z, S, None, 0.005)), 4)
        rec[name + "_loss"] = float(loss)
    print(json.dumps(rec), flush=True)

# mode 1 (vae.py:327-336, what run_training.py's main trains) on the same relation matrix with latents that lie apart (unit-
# variance elements: sim ~ 2 > margin / |w_n| = 1, every unrelated hinge inactive): the gradient product multiplies only the
# blocks of S the forward call marked
for B, n in ((2048, 4096), (768, 65536)):
    z = torch.randn(B, n, device=dev)
    i = torch.arange(B, device=dev)
    same = (i[:, None] // 8) == (i[None, :] // 8)
    tm = torch.where(same & (i[:, None] != i[None, :]), torch.where((i[:, None] - i[None, :]).abs() == 1, 2.0, 1.0), 0.0).float()
    loss, S = ops.time_matching_forward(z, tm, 1, 1.1, 0.1, -0.5, 0.5)
    rec = {"B": B, "n": n, "mode": 1, "relation_entries_per_row": 7.0,
           "forward_ms": round(t_ms(lambda: ops.time_matching_forward(z, tm, 1, 1.1, 0.1, -0.5, 0.5)), 4),
           "backward_marked_blocks_ms": round(t_ms(lambda: ops.time_matching_backward(z, S, None, 0.005)), 4)}
    Sd = S.clone()
    rec["backward_every_block_ms"] = round(t_ms(lambda: ops.time_matching_backward(z, Sd, None, 0.005)), 4)
    print(json.dumps(rec), flush=True)
