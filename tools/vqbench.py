#!/usr/bin/env python3
"""VectorQuantizer forward micro-bench: the exact kernel vs the MFMA filter + exact re-check, per shape.

    gpurun -- python tools/vqbench.py                 # headline (K 64, D 16, B 2048), stress (K 4096), example (K 512, D 64)
    rocprofv3 --kernel-trace --stats ... -- python3 tools/vqbench.py headline

Times whole dm_vq_forward calls (prep + distance kernel) with events on the launch stream; per-kernel durations come
from the rocprofv3 kernel trace of the same command.  Algorithmic bytes = P * (2 * D * 4 + 8) (read z, write the
straight-through value, write int64 indices): SURVEY section 8(d).
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamorph_amd import ops
from dynamorph_amd._lib import DM_VQ_BF16, DM_VQ_EXACT, DM_VQ_MFMA


def t_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters


SHAPES = {"headline": (2048, 16, 64, 16, 16), "stress": (256, 16, 4096, 32, 32), "example": (256, 64, 512, 32, 32),
          "c2": (1024, 16, 64, 16, 16), "c5init": (1024, 16, 4096, 32, 32), "c5model": (1024, 16, 4096, 32, 32), "big": (8192, 16, 64, 16, 16), "tiny": (1, 16, 64, 16, 16), "small": (256, 16, 64, 16, 16)}
want = sys.argv[1:] or ["headline", "stress", "example", "c2"]
dev = "cuda:0"
for name in want:
    B, D, K, H, W = SHAPES[name]
    z = torch.randn(B, D, H, W, device=dev, generator=torch.Generator(dev).manual_seed(1))
    cb = torch.randn(K, D, device=dev, generator=torch.Generator(dev).manual_seed(2))
    if name == "c5model":                                  # BASELINE configs[4] as bench.py times it: the model's own latents / codebook
        import copy, numpy as np, dynamorph_amd
        from dynamorph_amd import engine as E
        torch.manual_seed(5)
        m = dynamorph_amd.VQ_VAE(num_inputs=4, num_embeddings=K, channel_var=np.ones(4)).to(dev)
        x = torch.randn(B, 4, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(77))
        with torch.no_grad():
            z, _ = E.encoder_forward(E.Layers(copy.deepcopy(m)), x)
        cb = m.vq.w.weight.detach().clone()
        del x, m
        print(f"c5model: |z| rms {float(z.pow(2).mean().sqrt()):.3f}, codebook rms {float(cb.pow(2).mean().sqrt()):.2e}", flush=True)
    if name == "c5init":                                   # the module's own initial codebook (vq_vae.py:40): uniform(-1/K, 1/K)
        cb = (torch.rand(K, D, device=dev, generator=torch.Generator(dev).manual_seed(2)) * 2 - 1) / K
    P = B * H * W
    nbytes = P * (2 * D * 4 + 8)
    ref = None
    for vname, variant in (("exact", DM_VQ_EXACT), ("mfma", DM_VQ_MFMA), ("bf16", DM_VQ_BF16)):
        if variant == DM_VQ_BF16 and D % 16:
            continue
        if os.environ.get("VQBENCH_ONLY") and vname not in os.environ["VQBENCH_ONLY"].split(","):
            continue
        idx, out, slabs, hist, nre = ops.vq_forward(z, cb, variant=variant, want_rechecked=True)
        if ref is None:
            ref = idx
        same = bool(torch.equal(idx, ref))
        ms = t_ms(lambda: ops.vq_forward(z, cb, variant=variant))
        # the distance kernel alone: T(21 launches) - T(1 launch) of dm_vq_forward_repeat
        bufs = ops.vq_forward_repeat(z, cb, 1, variant=variant)
        k1 = t_ms(lambda: ops.vq_forward_repeat(z, cb, 1, variant=variant, bufs=bufs), iters=10)
        k21 = t_ms(lambda: ops.vq_forward_repeat(z, cb, 21, variant=variant, bufs=bufs), iters=5)
        kms = max((k21 - k1) / 20, 1e-6)
        print(f"{name:9s} {vname:6s} B={B} D={D} K={K} P={P}: {ms * 1e3:8.1f} us/call  kernel {kms * 1e3:7.1f} us = {nbytes / kms / 1e6 / 8000:.3f} of HBM peak"
              f" ({2.0 * K * D * P / kms / 1e9:7.1f} TFLOP/s of filter product)  rechecked={int(nre.cpu())} ({int(nre.cpu()) / P:.2e})  same_idx={same}  codes_used={int((hist > 0).sum())}",
              flush=True)
