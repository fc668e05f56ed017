#!/usr/bin/env python3
"""bench.py -- cell-patches/s of the VQ-VAE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N fresh ranks itself, dynamorph_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (the same ranks started from outside)

A "step" is one pass of the hot path over one batch of synthetic 2x128x128 fp32 patches that is already
resident in HBM.  Default workload = BASELINE.json configs[2]/[3]: one TRAINING step (forward + backward +
Adam, batch 2048 per GPU; weak scaling, one RCCL all-reduce of the flat 96 KB gradient bucket per step).
`--workload c2` times configs[1] instead (inference latents, batch 1024, per-sample BatchNorm statistics).

`--workload c5` times configs[4] (stress: VQ_VAE(num_inputs=4, num_embeddings=4096) on 4x256x256 patches, training step).

Rank 0 prints ONE JSON line.  It also carries
  roofline     -- the dominant kernel of the step, timed live with events on the launch stream
  cpu_baseline -- the CPU oracle (oracle/vqvae_oracle.py, "port") timed on this host's cores on a bounded
                  sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = 157.3 TFLOP/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA ~2.5 PFLOP/s dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 200; 50 for c5 / z32ex)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["c3", "c2", "c5", "z32ex"], default="c3")
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch (default 2048 for c3, 1024 for c2, 1024 for c5, 768 for z32ex)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-fused", action="store_true", help="z32ex: the autograd + torch.optim.Adam loop instead of FusedTrainer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-targets", action="store_true", help="skip the north-star target measurements and the C2 sub-record")
    args = ap.parse_args()
    if args.steps <= 0:
        args.steps = 50 if args.workload in ("c5", "z32ex") else 200
    return args


def event_time_ms(fn, iters=20, warmup=3):
    """Average duration of fn() on the current stream (the stream our kernels are launched on)."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def pmc_traffic(kernel_key, batch):
    """HBM bytes per launch of the roofline kernel, from the separate `rocprofv3 --pmc` passes summarised in
    profiles/r0N_pmc_traffic.json (tools/pmc_traffic.py writes it; bench.py cannot run under the profiler
    itself).  None when no measurement for this kernel and batch has been committed."""
    stem = {"dec_tail_train": "dec_tail_backward_kernel", "conv4x4s2_e1": "conv4x4s2_kernel", "vq_cells_k4096": "vq_cells_kernel"}.get(
        kernel_key, kernel_key.split("_k4096")[0].replace("vq_forward_mfma", "vq_forward_mfma_kernel"))
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):   # the newest measurement that has this kernel
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f).get(kernel_key)
        except (OSError, ValueError):
            rec = None
        # (a record names the kernel it was measured on: a pattern that drifted onto another kernel must not pass as this one's)
        if rec and rec.get("batch") == batch and stem in rec.get("kernel", ""):
            return rec.get("hbm_bytes_per_launch")
    return None


PROFILE_STATS = ("r06_c3_b2048_kernel_stats.csv", "r05_c3_b2048_kernel_stats.csv", "r04_c3_b2048_kernel_stats.csv")


def profile_avg_us(kernel_stem):
    """(average per-dispatch duration in us, file) of a kernel inside the step from the newest committed `rocprofv3
    --kernel-trace --stats` summary of `bench.py` -- a RECORD of an earlier build, reported beside the live measurement under
    its own name and never in place of it; (None, None) when there is none."""
    import csv
    for name in PROFILE_STATS:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                for r in csv.DictReader(f):
                    if kernel_stem in r["Name"]:
                        return round(float(r["AverageNs"]) / 1e3, 2), name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


PROFILE_SQ = ("r06_c3_b2048_sq_counters.json", "r05_c3_b2048_sq_counters.json")


def profile_mfma_busy(kernel):
    """(SQ_VALU_MFMA_BUSY_CYCLES per SIMD as a percentage of the cycles a wave of `kernel` is resident, file) from the newest
    committed `rocprofv3 --pmc` summary (tools/sqprof.sh -> profiles/r0N_c3_b2048_sq_counters.json): the north star's "MFMA
    utilisation ... evidenced by rocprof MFMA-busy".  A RECORD of a profiled run, reported beside the live FLOP-based
    fraction under its own name; (None, None) when there is none."""
    stem = kernel.rstrip(">")
    for name in PROFILE_SQ:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ks = json.load(f).get("kernels", {})
        except (OSError, ValueError):
            continue
        for k, v in ks.items():
            if k.startswith(stem):
                return v.get("mfma_busy_pct_of_wave_cycles"), name
    return None, None


EXAMPLE_CONFIG = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)     # config_example.yml:157-163


def roofline_wide_conv(B):
    """z32ex: the 3x3 64 -> 64 convolution of the residual blocks on the 32 x 32 latent (the largest share of that step; since
    round 6 conv3x3_wide_stream_kernel -- all weights resident in LDS, activations straight into the matrix instruction's B
    operand; DM_WIDE_STREAM=0: the tiled conv_wide_kernel<1, 9, 4>), forward form, alone, timed with events.  MFMA bound:
    achieved = 2*P*64*64*9 FLOP / duration."""
    from dynamorph_amd import ops
    C, HW = 64, 32
    x = torch.randn(B, C, HW, HW, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, C, HW, HW, device="cuda")
    wv = ops.weight_view(w, C * 9, 9, 3, 1)

    def fn():
        ops.conv3x3(ops.Op(x, 1), wv, B, C, C, HW, HW, taps=9, out=out, want_stats=True)
    ms = event_time_ms(fn)
    flops = 2.0 * B * HW * HW * C * C * 9
    ach = flops / (ms * 1e-3) / 1e12
    stream = int(os.environ.get("DM_WIDE_STREAM", "511")) & 128
    return {"kernel": ("conv3x3_wide_stream_kernel<false>" if stream else "wide_pack_kernel + conv_wide_kernel<1, 9, 4>") +
                      " (residual 3x3, 64 -> 64 channels, 32 x 32, forward form)", "bound": "mfma",
            "achieved": round(ach, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": None, "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops}


def roofline_vq_large_codebook(model, x):
    """c5 (BASELINE.json configs[4]: "LDS-tiled codebook distance kernel vs roofline"): the K = 4096 distance + argmin
    kernel (round 6: vq_cells_kernel, csrc/vq_cells.h -- matrix filter on the bf16 pipe + exact evaluation of the best cell /
    exact re-check; DM_VQ_CELLS=0: the round-5 kernel that walks the codebook through LDS) on the model's own latents, alone, as T(21 launches) - T(1 launch) with events on the launch stream.  Compute
    bound: achieved = 2*K*D*P FLOP of the filter product / average launch, against the f32 MFMA peak."""
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    L = E.Layers(model)
    with torch.no_grad():
        z, _ = E.encoder_forward(L, x)
    cbk = L.codebook.weight.detach()
    bufs = ops.vq_forward_repeat(z, cbk, 1)
    t1 = event_time_ms(lambda: ops.vq_forward_repeat(z, cbk, 1, bufs=bufs), iters=10, warmup=3)
    t21 = event_time_ms(lambda: ops.vq_forward_repeat(z, cbk, 21, bufs=bufs), iters=5, warmup=1)
    ms = max((t21 - t1) / 20.0, 1e-6)
    ops.vq_forward_repeat(z, cbk, 1, bufs=bufs)      # (the re-check counter below is per LAUNCH: rounds 3-5 read it after the
    torch.cuda.synchronize()                          #  21-launch call and reported 21 launches' worth -- 12.5 % instead of 0.6 %)
    K, D = cbk.shape
    P = z.shape[0] * z.shape[2] * z.shape[3]
    flops = 2.0 * K * D * P                      # the filter product |e|^2 - 2 z.e as an f32 GEMM: the algorithmic count
    useful = flops / (ms * 1e-3) / 1e12
    # DM_VQ_AUTO runs the product on v_mfma_f32_16x16x32_bf16 with both operands split into a bf16 head and remainder:
    # 4 bf16 multiply-adds EXECUTED per algorithmic one, on the bf16 matrix pipe (2.5 PFLOP/s dense); DM_VQ_FILTER=f32 keeps
    # the f32-input instruction (157.3 TFLOP/s, 1 executed per algorithmic).  The roofline fraction is ALGORITHMIC work over
    # the peak of the pipe the kernel runs on; the executed rate is reported beside it, not as `achieved`.
    bf16 = D % 16 == 0 and os.environ.get("DM_VQ_FILTER", "")[:1] != "f"
    peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
    # which kernel dm_vq_forward takes (csrc/vq.hip, vq_forward_launch): 64 < K <= 4096 at embedding_dim 16 on grids that are a
    # multiple of 128 positions run vq_cells_kernel (csrc/vq_cells.h): v_mfma_f32_32x32x16_bf16, THREE products of the split
    cells = bf16 and D == 16 and 64 < K <= 4096 and (z.shape[2] * z.shape[3]) % 128 == 0 and os.environ.get("DM_VQ_CELLS", "")[:1] != "0"
    products = (4 if os.environ.get("DM_VQ_CELLS_PROD", "")[:1] == "4" else 3) if cells else (4 if bf16 else 1)
    executed = useful * products
    if cells:
        kname = (f"vq_cells_kernel<{products}> (K = {K} codes straight from L2 into a register ring, {products} products of the bf16 split on "
                 "v_mfma_f32_32x32x16_bf16, cell minima + exact evaluation of the best cell; distance + first-min argmin + gather + "
                 "straight-through value + squared error)")
        pipe = "bf16 matrix (v_mfma_f32_32x32x16_bf16)"
    else:
        kname = (f"vq_forward_mfma_kernel<{D}, false, ..., {'bf16-split' if bf16 else 'f32'} filter> (K = {K} codes through LDS pieces; "
                 "distance + first-min argmin + gather + straight-through value + squared error)")
        pipe = "bf16 matrix (v_mfma_f32_16x16x32_bf16)" if bf16 else "f32 matrix (v_mfma_f32_16x16x4_f32)"
    return {"kernel": kname, "bound": "mfma",
            "achieved": round(useful, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(useful / peak, 4),
            "pipe": pipe,
            "traffic": pmc_traffic("vq_cells_k4096" if cells else "vq_forward_mfma_k4096", x.shape[0]), "avg_launch_ms": round(ms, 5),
            "algorithmic_flops_per_launch": flops,
            "algorithmic_vs_f32_mfma_peak": round(useful / MFMA_F32_PEAK_TFLOPS, 4),
            "executed_tflops": round(executed, 2), "executed_frac_of_pipe_peak": round(executed / peak, 4),
            "executed_flops_per_launch": flops * products,
            "algorithmic_bytes_per_launch": P * (2 * D * 4 + 8),
            "rechecked_positions": int(bufs[4][:1].view(torch.int32).item()), "positions": P}


def roofline_dominant_kernel(model, x, workload):
    """Dominant kernel of the step by total time in profiles/ (c3: the fused backward decoder tail,
    dec_tail_backward_kernel; c2: the enc.0 o enc.1 composite 4x4/s2 convolution), launched alone on the bench
    shapes and timed with events on the launch stream.  achieved = algorithmic bytes / average duration."""
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    from dynamorph_amd.ops import Op, weight_view
    L = E.Layers(model)
    B, NIN, H, W = x.shape
    if workload == "c2":
        c1 = L.nh // 2
        weff, border = ops.e1_compose_border(L.enc0.weight.detach(), L.enc0.bias.detach(), L.enc1.weight.detach(),
                                             L.enc1.bias.detach())
        a1 = torch.empty(B, c1, H // 2, W // 2, device=x.device)

        def fn():
            ops.conv4x4s2(Op(x), weight_view(weff, (NIN + 1) * 16, 16, 4, 1), B, NIN, c1, H, W, out=a1,
                          want_stats=True, bias_border=border)
        key, name = "conv4x4s2_e1", "conv4x4s2_kernel<2,1,...,true> (enc.0 o enc.1 composite, border-bias table)"
        algo_bytes = B * (NIN * H * W + c1 * (H // 2) * (W // 2)) * 4          # read x once, write a1 once
    else:
        c2 = L.dec4.weight.shape[0]
        if not ops.dec_tail_supported(c2, NIN, H // 2, W // 2):
            return None
        # the kernel's real operand: d2 of this model on this batch (its sparsity sets the pace of the ReLU-masked
        # phases, so random data would not reproduce the in-step duration the profile shows)
        with torch.no_grad():
            z_b, _ = E.encoder_forward(L, x)
            z_q, _, _ = E.vq_forward(L.codebook.weight, z_b, float(model.commitment_cost))
            _, dcx = E.decoder_forward(L, z_q, x, None, defer_tail=True)
        d2 = dcx.d2
        w4, b4 = L.dec4.weight.detach(), L.dec4.bias.detach()
        w6, b6 = L.dec6.weight.detach(), L.dec6.bias.detach()
        var = L.channel_var.detach().to(x.device, torch.float32).reshape(-1).contiguous()
        gs = torch.ones(1, device=x.device)

        def fn():
            ops.dec_tail_train(d2, w4, b4, w6, b6, x, None, var, gs)
        key = "dec_tail_train"
        name = f"dec_tail_backward_kernel<{NIN}, true> (dm_dec_tail_train: dec.4/dec.5/dec.6 + loss, forward and backward fused)"
        # read d2 and x once, write g2 once
        algo_bytes = B * (2 * c2 * (H // 2) * (W // 2) + NIN * H * W) * 4
    # ten launches captured in one HIP graph and replayed, events on the launch stream around the replays: the average is
    # the kernel's (a fresh box's first eager launches can be host-bound: allocations, the ctypes call)
    def ten():
        for _ in range(10):
            fn()
    ms = event_time_ms(graphed(ten), iters=5, warmup=2) / 10.0
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    rec = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(key, B), "avg_launch_ms": round(ms, 4),
           "algorithmic_bytes_per_launch": algo_bytes}
    if workload != "c2":
        # the same launch against the OTHER roof: dec.4 (transposed 4x4/s2, c2 -> c2 channels) and dec.6 (1x1, c2 -> NIN), each
        # forward + data gradient + weight gradient, in exact fp32 (64 FLOP per cycle and SIMD on either pipe, DESIGN.md
        # section 3): 2 * 3 * H * W * (c2 * c2 * 4 + NIN * c2) FLOP per patch.  Its intensity lies above the machine balance
        # (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP per byte): by the roofline model this kernel is compute bound, the HBM fraction
        # above is kept as the headline because SURVEY 8(d) prices the thin layers against HBM.
        flops = 2.0 * 3 * B * H * W * (c2 * c2 * 4 + NIN * c2)
        tf = flops / (ms * 1e-3) / 1e12
        rec["compute"] = {"algorithmic_flops_per_launch": flops, "achieved_tflops": round(tf, 1), "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                          "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                          "flop_per_algorithmic_byte": round(flops / algo_bytes, 1),
                          "machine_balance_flop_per_byte": round(MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9), 1)}
    return rec


def north_star_targets(model, x):
    """The two numeric bars of BASELINE.json's north star, measured live on the bench shapes with events on the launch
    stream (the same kernels the timed step replays):
      vq        -- VectorQuantizer distance + argmin (+ gather, straight-through value, squared error, code counters):
                   34 816 algorithmic bytes per patch / average launch of the distance kernel alone
                   (dm_vq_forward_repeat: T(21 launches) - T(1 launch)); `call_ms` is the whole dm_vq_forward as the
                   training step and the inference path call it (hist = NULL: one launch -- the step's single scalar launch
                   reads the per-workgroup counters from the workspace), `call_with_hist_ms` the stand-alone form that also
                   reduces the counters into `hist` (two launches);
      enc_convs -- every forward convolution of the encoder: useful FLOPs (2 x MACs of SURVEY 2.2) / average launch
                   (inside a HIP graph of 10 launches, as the step replays them), against the f32 MFMA peak."""
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    from dynamorph_amd.ops import DM_LOAD_AFFINE_RELU, DM_LOAD_RELU, Op, weight_view
    L = E.Layers(model)
    B, NIN, H, W = x.shape
    w = lambda p: p.detach()
    with torch.no_grad():
        z, cx = E.encoder_forward(L, x)
    # ---- VQ
    cbk = L.codebook.weight.detach()
    bufs = ops.vq_forward_repeat(z, cbk, 1)
    t1 = event_time_ms(lambda: ops.vq_forward_repeat(z, cbk, 1, bufs=bufs), iters=30, warmup=5)
    t21 = event_time_ms(lambda: ops.vq_forward_repeat(z, cbk, 21, bufs=bufs), iters=10, warmup=2)
    # back-to-back calls from the host are launch bound: the whole call is timed as a graph replay of ten
    def call_time(want_hist):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.vq_forward_repeat(z, cbk, 1, bufs=bufs, want_hist=want_hist)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            for _ in range(10):
                ops.vq_forward_repeat(z, cbk, 1, bufs=bufs, want_hist=want_hist)
        return event_time_ms(g.replay, iters=10, warmup=2) / 10
    call_ms, call_hist_ms = call_time(False), call_time(True)
    k_ms = max((t21 - t1) / 20.0, 1e-6)
    D = z.shape[1]
    P = z.shape[0] * z.shape[2] * z.shape[3]
    vq_bytes = P * (2 * D * 4 + 8)
    # ONE dispatch at a time, as the step runs it: events on the launch stream around a single launch that follows its
    # producer (in the step the residual join writes z just before; here a device copy of z does).  ~150 us of unrelated
    # device work in front keeps the host ahead of the device, so the interval holds no wait for the host.
    z_src = z.clone()
    pad = torch.empty(64 << 20, device=z.device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24)]
    for e0, e1 in ev:
        pad.fill_(0.0)
        z.copy_(z_src)
        e0.record()
        ops.vq_forward_repeat(z, cbk, 1, bufs=bufs, want_hist=False)
        e1.record()
    torch.cuda.synchronize()
    d_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev[4:])
    d_ms = d_ms[len(d_ms) // 2]                                # median of 20
    prof_us, prof_file = profile_avg_us("vq_forward_mfma_kernel<16, true, 3, true, true, false>")
    f_ev = round(vq_bytes / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    f_prof = round(vq_bytes / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if prof_us else None
    vq = {"kernel": "vq_forward_mfma_kernel (MFMA filter + exact re-check)", "bytes": vq_bytes,
          # LIVE, per dispatch: events on the launch stream around single launches behind their producer (the interval also
          # holds the event records' own few microseconds); the committed profile's average is a record beside it
          "frac_hbm": f_ev,
          "frac_hbm_is": "live, per dispatch: events around single launches behind their producer (median of 20)",
          "dispatch_ms_events": round(d_ms, 5),
          "frac_hbm_profile": f_prof, "profile_avg_us": prof_us, "profile_file": prof_file,
          "back_to_back_launch_ms": round(k_ms, 5),
          "frac_hbm_back_to_back": round(vq_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "call_ms": round(call_ms, 5),
          "call_frac_hbm": round(vq_bytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
          "call_with_hist_ms": round(call_hist_ms, 5),
          "rechecked_positions": int(bufs[4][:1].view(torch.int32).item()) // 1}
    # the same kernel on four times the positions (the batch repeated): how much of the stand-alone launch's shortfall is the
    # launch's size -- 2.7 chunks per resident wave at B = 2048 -- and not the kernel's loop
    try:
        z4 = z.repeat(4, 1, 1, 1)
        bufs4 = ops.vq_forward_repeat(z4, cbk, 1)
        u1 = event_time_ms(lambda: ops.vq_forward_repeat(z4, cbk, 1, bufs=bufs4), iters=10, warmup=3)
        u11 = event_time_ms(lambda: ops.vq_forward_repeat(z4, cbk, 11, bufs=bufs4), iters=5, warmup=1)
        k4_ms = max((u11 - u1) / 10.0, 1e-6)
        vq["back_to_back_launch_ms_at_4x_batch"] = round(k4_ms, 5)
        vq["frac_hbm_at_4x_batch"] = round(4 * vq_bytes / (k4_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        del z4, bufs4
    except torch.cuda.OutOfMemoryError:
        pass
    # the quantiser as the STEP dispatches it since round 4: the encoder's last residual join in its load path
    # (dm_vq_forward_join reads rb and h_in, writes z, the quantised value and the codes: twice the bytes in one launch)
    sv = cx.res[-1] if cx.res else None
    if sv is not None and ops.vq_forward_join_supported(D, cbk.shape[0], z.shape[2], z.shape[3]):
        rb_src, h_src = sv.rb.clone(), sv.h_in.clone()
        rb_b, h_b = torch.empty_like(rb_src), torch.empty_like(h_src)
        for e0, e1 in ev:
            pad.fill_(0.0)
            rb_b.copy_(rb_src)
            h_b.copy_(h_src)
            e0.record()
            ops.vq_forward_join(rb_b, h_b, sv.coefb, cbk)
            e1.record()
        torch.cuda.synchronize()
        j_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev[4:])
        j_ms = j_ms[len(j_ms) // 2]
        j_bytes = P * (4 * D * 4 + 8)
        j_prof, j_file = profile_avg_us("vq_forward_mfma_kernel<16, true, 3, true, true, true>")
        jf_ev = round(j_bytes / (j_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        jf_prof = round(j_bytes / (j_prof * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if j_prof else None
        vq["in_step"] = {"kernel": "vq_forward_mfma_kernel<16, true, 3, true, true, JOIN> (dm_vq_forward_join: residual join + distance + "
                                   "argmin + gather + straight-through value + squared error + code counters)",
                         "bytes": j_bytes, "bytes_are": "rb + h_in read, z + quantised written (4 x 16 384 B per patch) + int64 codes",
                         "frac_hbm": jf_ev, "frac_hbm_is": "live, per dispatch (events, median of 20)",
                         "dispatch_ms_events": round(j_ms, 5),
                         "frac_hbm_profile": jf_prof, "profile_avg_us": j_prof, "profile_file": j_file,
                         "vq_only_bytes_frac_hbm": round(vq_bytes / (j_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del rb_src, h_src, rb_b, h_b
    del pad, z_src
    # ---- encoder forward convolutions, on the step's own activations
    nh, nrh, c1 = L.nh, L.nrh, L.nh // 2
    H1, W1, H2, W2, H3, W3 = cx.dims
    weff, border = ops.e1_compose_border(w(L.enc0.weight), w(L.enc0.bias), w(L.enc1.weight), w(L.enc1.bias))
    convs = []

    def graph_time_ms(fn, launches=10):
        """Average duration of one launch inside a HIP graph of `launches` back-to-back launches -- how the step runs
        them (host launch gaps of eager back-to-back calls are not part of the step)."""
        fn()
        gg = torch.cuda.CUDAGraph()
        sd = torch.cuda.Stream()
        sd.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sd):
            fn()
        torch.cuda.current_stream().wait_stream(sd)
        with torch.cuda.graph(gg):
            for _ in range(launches):
                fn()
        return event_time_ms(gg.replay, iters=10, warmup=2) / launches

    def add(name, macs_per_patch, fn, note=None, kernel=None):
        ms = graph_time_ms(fn)
        fl = 2.0 * macs_per_patch * B
        rec = {"layer": name, "flops": fl, "avg_launch_ms": round(ms, 5),
               "frac_mfma": round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
        if kernel:
            rec["kernel"] = kernel
            busy, bfile = profile_mfma_busy(kernel)
            if busy is not None:
                rec["mfma_busy_pct_profile"], rec["mfma_busy_profile_file"] = busy, bfile
        if note:
            rec["note"] = note
        convs.append(rec)
    a1 = torch.empty_like(cx.a1); a2 = torch.empty_like(cx.a2); a3 = torch.empty_like(cx.a3); a4 = torch.empty_like(cx.a4)
    # enc.1 o enc.0 is ONE 4x4/s2 conv over x with composite weights: it EXECUTES 16*NIN*8 MACs per output pixel
    # (1 048 576 per patch), a quarter of the reference's two layers (262 144 + 4 194 304); the executed count is priced
    add("enc.0+enc.1 (composite 4x4/s2, 2->8)", H1 * W1 * 16 * NIN * c1,
        lambda: ops.conv4x4s2(Op(x), weight_view(weff, (NIN + 1) * 16, 16, 4, 1), B, NIN, c1, H, W, out=a1, want_stats=True,
                              bias_border=border),
        note="HBM bound: reads x and writes a1, 262 144 B per patch; the reference's two layers would be 4 456 448 MACs per patch",
        kernel="conv4x4s2_pair_kernel<2, 8, 3>")
    add("enc.4 (4x4/s2, 8->16)", 2097152,
        lambda: ops.conv4x4s2(Op(cx.a1, DM_LOAD_AFFINE_RELU, cx.coef1), weight_view(w(L.enc4.weight), c1 * 16, 16, 4, 1), B, c1, nh,
                              H1, W1, out=a2, want_stats=True, bias=w(L.enc4.bias)),
        kernel="conv4x4s2_kernel<8, 1, 8, 32, 0, 2, false>")
    add("enc.7 (4x4/s2, 16->16)", 1048576,
        lambda: ops.conv4x4s2(Op(cx.a2, DM_LOAD_AFFINE_RELU, cx.coef2), weight_view(w(L.enc7.weight), nh * 16, 16, 4, 1), B, nh, nh,
                              H2, W2, out=a3, want_stats=True, bias=w(L.enc7.bias)),
        kernel="conv4x4s2_patch_forward_kernel")
    add("enc.10 (3x3, 16->16)", 589824,
        lambda: ops.conv3x3(Op(cx.a3, DM_LOAD_AFFINE_RELU, cx.coef3), weight_view(w(L.enc10.weight), nh * 9, 9, 3, 1), B, nh, nh,
                            H3, W3, taps=9, out=a4, want_stats=True, bias=w(L.enc10.bias)),
        kernel="conv3x3_kernel<16, 1, 1, 9, false, 16, 16, false, 0, 3>")
    sv = cx.res[0]
    ca, bna, cb2, bnb = L.res[0]
    ra = torch.empty_like(sv.ra); rb = torch.empty_like(sv.rb)
    add("enc.12 residual 3x3 (16->32), each of 2", 1179648,
        lambda: ops.conv3x3(Op(sv.h_in, DM_LOAD_RELU), weight_view(w(ca.weight), nh * 9, 9, 3, 1), B, nh, nrh, H3, W3, taps=9,
                            out=ra, want_stats=True, bias=w(ca.bias)),
        kernel="conv3x3_kernel<16, 2, 1, 9, false, 16, 16, false, 0, 2>")
    add("enc.12 residual 1x1 (32->16), each of 2", 131072,
        lambda: ops.conv3x3(Op(sv.ra, DM_LOAD_AFFINE_RELU, sv.coefa), weight_view(w(cb2.weight), nrh, 1, 0, 0), B, nrh, nh, H3, W3,
                            taps=1, out=rb, want_stats=True, bias=w(cb2.bias)),
        kernel="conv3x3_kernel<32, 1, 1, 1, false, 8, 16, false, 0, 3>", note="HBM bound: 49 152 B per patch")
    tot_fl = sum(c["flops"] * (2 if "each of 2" in c["layer"] else 1) for c in convs)
    tot_ms = sum(c["avg_launch_ms"] * (2 if "each of 2" in c["layer"] else 1) for c in convs)
    total = {"flops": tot_fl, "ms": round(tot_ms, 5),
             "frac_mfma": round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
    total["note"] = "frac_mfma = executed FLOPs / average launch (graph replay of 10 launches) / 157.3 TFLOP/s: clock- and prologue-inclusive"
    # the counter the north star names, from the committed profile: matrix-pipe busy cycles over the encoder's forward
    # convolutions, weighted by their (live) launch times
    wsum = sum(c["avg_launch_ms"] * (2 if "each of 2" in c["layer"] else 1) for c in convs if "mfma_busy_pct_profile" in c)
    if wsum > 0:
        total["mfma_busy_pct_profile_time_weighted"] = round(
            sum(c["mfma_busy_pct_profile"] * c["avg_launch_ms"] * (2 if "each of 2" in c["layer"] else 1)
                for c in convs if "mfma_busy_pct_profile" in c) / wsum, 1)
        total["mfma_busy_is"] = ("SQ_VALU_MFMA_BUSY_CYCLES / cycles a wave is resident, per kernel, from the committed rocprofv3 --pmc "
                                 "summary (a record of a profiled run, 1.7-1.9 GHz; the FLOP fraction above prices the same launches "
                                 "against the 2.4 GHz peak)")
    return {"vq": vq, "enc_convs": convs, "enc_convs_total": total}


def graphed(step_fn):
    """step_fn recorded once into a HIP graph (after a warm-up on a side stream) -> a function that replays it and
    returns the recorded call's output.  Same kernels in the same order; what goes away is the per-launch host work, which
    at 30 launches per 0.4 ms would otherwise make the inference step depend on how busy the host is."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step_fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step_fn()

    def replay():
        g.replay()
        return out
    return replay


def c2_record(model, steps=50, warmup=3, B=1024):
    """BASELINE.json configs[1] beside the headline line: inference latents of 1024 patches (enc + vq with per-sample
    BatchNorm statistics = process_VAE), inputs resident in HBM."""
    from dynamorph_amd import engine as E
    dev = next(model.parameters()).device
    x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(99)).to(dev)
    L = E.Layers(model)
    e1 = E.e1_operands(L)            # composite first-layer weights: once per model load (encode_patches: once per call)

    def step():
        with torch.no_grad():
            z_b, cx = E.encoder_forward(L, x, per_sample=True, e1=e1, join=False, latents_only=True)
            z_a = E.vq_forward(L.codebook.weight, z_b, float(model.commitment_cost), want_scalars=False)[0]
            cx.join()                # the running-statistics replay ran on a helper stream beside the quantiser
            return z_a
    step = graphed(step)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"workload": "C2: inference latents (enc + vq, per-sample BatchNorm statistics), batch 1024", "value": round(B * steps / el, 1),
            "unit": "patches/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps, "hip_graph": True}


def c5_record(dev, steps=30, warmup=5, B=1024):
    """BASELINE.json configs[4]'s single-GPU leg beside the headline line, budgeted to a few seconds: the large-codebook
    stress model (4-channel 256 x 256 patches, 4096 codes) training step at its bench batch, and its K = 4096 distance /
    argmin kernel timed per dispatch on the model's own latents (`python bench.py --workload c5` is the full record)."""
    import numpy as np
    from dynamorph_amd import VQ_VAE
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    from dynamorph_amd.train import FusedTrainer
    torch.manual_seed(0)
    model = VQ_VAE(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4)).to(dev)
    x = torch.randn(B, 4, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    def recheck_share():
        """Share of the batch's latent positions whose code the matrix filter could not settle (exact re-check), for the model
        as it is now: one launch, the counter of that launch alone."""
        with torch.no_grad():
            zz, _ = E.encoder_forward(E.Layers(model), x0)
        bb = ops.vq_forward_repeat(zz, model.vq.w.weight.detach(), 1)
        return int(bb[4][:1].view(torch.int32).item()) / (zz.shape[0] * zz.shape[2] * zz.shape[3])
    x0 = x
    share_init = recheck_share()                       # the reference's initialisation (nn.Embedding: N(0, 1) codes)
    tr = FusedTrainer(model, lr=1e-4)
    x = tr.prepare(x)
    for _ in range(warmup):
        out = tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.step(x)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L = E.Layers(model)
    with torch.no_grad():
        z, _ = E.encoder_forward(L, x)
    cbk = L.codebook.weight.detach()
    bufs = ops.vq_forward_repeat(z, cbk, 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for e0, e1 in ev:
        e0.record()
        ops.vq_forward_repeat(z, cbk, 1, bufs=bufs)          # preparation + kernel + counter reduction (3 launches)
        e1.record()
    torch.cuda.synchronize()
    call_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev[2:])[2]
    K, D = cbk.shape
    P = z.shape[0] * z.shape[2] * z.shape[3]
    flops = 2.0 * K * D * P
    useful = flops / (call_ms * 1e-3) / 1e12
    rec = {"workload": "C5: VQ_VAE(num_inputs=4, num_embeddings=4096) training step, 4x256x256 fp32, batch %d" % B,
           "value": round(B * steps / el, 1), "unit": "patches/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps,
           "vq_call_ms": round(call_ms, 4), "vq_call_is": "dm_vq_forward on the model's own latents: preparation + distance/argmin "
           "kernel + counter reduction, events around single calls (median of 5)",
           "vq_algorithmic_tflops": round(useful, 2), "vq_frac_of_bf16_matrix_peak": round(useful / MFMA_BF16_PEAK_TFLOPS, 4),
           "vq_frac_of_f32_matrix_peak": round(useful / MFMA_F32_PEAK_TFLOPS, 4),
           "rechecked_share": round(int(bufs[4][:1].view(torch.int32).item()) / P, 5), "positions": P,
           "rechecked_share_at_init": round(share_init, 5),
           "total_loss_after": round(float(out[2]), 6)}
    # ... and after 100 optimisation steps in all (lr 1e-4): what training does to the share of unsettled positions
    for _ in range(max(0, 100 - warmup - steps)):
        out = tr.step(x)
    x0 = x
    rec["rechecked_share_after_100_steps"] = round(recheck_share(), 5)
    rec["total_loss_after_100_steps"] = round(float(out[2]), 6)
    del tr, model, x, x0, z, bufs
    torch.cuda.empty_cache()
    return rec


def example_tm_matrix(B, dev):
    """The relation matrix of the z32ex workload: adjacent frames of one trajectory, 4 consecutive samples per group."""
    i = torch.arange(B)
    d = (i[:, None] - i[None, :]).abs()
    same = (i[:, None] // 4) == (i[None, :] // 4)
    m = torch.zeros(B, B)
    m[same & (d == 1)] = 2.0
    m[same & (d == 2)] = 1.0
    return m.to(dev)


def z32ex_record(dev, steps=20, warmup=4, B=768):
    """SURVEY section 8(f) row 2 beside the headline line, budgeted to a few seconds: VQ_VAE_z32 as the reference's example
    configuration trains it (config_example.yml:156-186: 64 / 64 / 512, batch 768, time-matching term on) through FusedTrainer
    (`python bench.py --workload z32ex` is the full record)."""
    from dynamorph_amd import VQ_VAE_z32
    from dynamorph_amd.train import FusedTrainer
    torch.manual_seed(0)
    model = VQ_VAE_z32(weight_matching=100., margin=1., w_a=1., w_t=0.5, w_n=-0.5, **EXAMPLE_CONFIG).to(dev)
    x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234)).to(dev)
    tm = example_tm_matrix(B, dev)
    tr = FusedTrainer(model, lr=1e-4)
    for _ in range(warmup):
        out = tr.step(x, None, tm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.step(x, None, tm)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rec = {"workload": "z32ex: VQ_VAE_z32(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512) training step with the "
                       "time-matching term, 2x128x128 fp32, batch %d" % B,
           "value": round(B * steps / el, 1), "unit": "patches/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps,
           "total_loss_after": round(float(out[2]), 6)}
    del tr, model, x, tm
    torch.cuda.empty_cache()
    return rec


def train_loop_record(dev, resident_ms_per_step, n=32768, B=2048, epochs=3, feed="auto", transform=True, masks=False,
                      relation=False, pinned=False):
    """The product's own training entry point, dynamorph_amd.train.train() (run_training.py:455-551), end to end on a
    synthetic HOST dataset: n patches, batch B, `epochs` timed epochs after one warm-up epoch (graph captures, allocator),
    augmentation on, validation block 1/8 of the data.  What `value` above cannot show: gathering a batch (on the device
    from the dataset in HBM, or over PCIe), the augmentation, the validation pass, the loss read-back.
    train_patches_per_s = training-phase samples / device time between the phase's first and last launch (every gap the
    host leaves inside the phase included); loop_patches_per_s = samples of both phases / wall clock of whole epochs
    (validation pass, loss read-back, early-stopping checkpoint included)."""
    import tempfile
    import numpy as np
    from dynamorph_amd import VQ_VAE
    from dynamorph_amd.train import train
    g = torch.Generator(device=dev).manual_seed(4321)
    host = torch.empty(n, 2, 128, 128, pin_memory=pinned)
    for lo in range(0, n, 4096):            # generated on the device (host randn of 2 GB takes seconds), handed over as a host tensor
        host[lo:lo + 4096].copy_(torch.randn(min(4096, n - lo), 2, 128, 128, device=dev, generator=g))
    data = torch.utils.data.TensorDataset(host)
    mask = rel = None
    if masks:
        mask = torch.utils.data.TensorDataset((torch.rand(n, 2, 128, 128) > 0.3).float() * 2 - 1)
    if relation:
        import scipy.sparse as sp
        i = np.arange(n - 1)
        same = (i // 8) == ((i + 1) // 8)           # trajectories of 8 consecutive frames
        rel = sp.coo_matrix((np.full(same.sum() * 2, 2.0), (np.r_[i[same], i[same] + 1], np.r_[i[same] + 1, i[same]])),
                            shape=(n, n)).tocsr()
    torch.manual_seed(0)
    np.random.seed(0)
    model = VQ_VAE().to(dev)
    st = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as out, open(os.devnull, "w") as null:
        import contextlib
        with contextlib.redirect_stdout(null):
            train(model, data, out, relation_mat=rel, mask=mask, n_epochs=epochs + 1, lr=1e-4, batch_size=B, device=dev,
                  shuffle_data=not relation, transform=transform, val_split_ratio=0.125, patience=epochs + 2, feed=feed, stats=st)
    total = time.perf_counter() - t0
    tr, va = st["phase_seconds"]["train"][1:], st["phase_seconds"]["val"][1:]
    ntr, nva = st["phase_samples"]["train"], st["phase_samples"]["val"]
    rate = ntr * len(tr) / sum(tr)
    rec = {"entry_point": "dynamorph_amd.train.train (run_training.py:455-551)", "feed": st["feed"], "dataset_patches": n,
           "batch": B, "timed_epochs": len(tr), "augmentation": bool(transform), "masks": masks, "relation_matrix": relation,
           "host_dataset": "pinned" if pinned else "pageable",
           "train_patches_per_s": round(rate, 1), "train_ms_per_step": round(1e3 * sum(tr) / len(tr) / (ntr / B), 4),
           "loop_patches_per_s": round((ntr + nva) * len(tr) / sum(st["epoch_seconds"][1:]), 1),
           "val_patches_per_s": round(nva * len(va) / sum(va), 1),
           "first_epoch_s": round(st["epoch_seconds"][0], 3), "setup_s": round(total - sum(st["epoch_seconds"]), 3)}
    if resident_ms_per_step:
        rec["vs_resident_bench"] = round(rate / (B / (resident_ms_per_step * 1e-3)), 4)
    return rec


def _host_cpu():
    """(model name, physical cores) of the host from lscpu; falls back to os.cpu_count() // 2."""
    import subprocess
    name, cores = "unknown", max((os.cpu_count() or 2) // 2, 1)
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        name = kv.get("Model name", name)
        cores = int(kv.get("Core(s) per socket", "0")) * int(kv.get("Socket(s)", "1")) or cores
    except Exception:
        pass
    return name, cores


def cpu_baseline(workload, budget_s=25.0):
    """The CPU oracle (kind "port": the restatement pinned to the reference by tests/golden) on this host's cores, on a
    bounded sample of the same workload, as BASELINE.md section 2 prescribes: thread count swept (a 128-thread run of a
    64-patch batch is oversubscribed, not the reference's speed), the best one kept, median of >= 10 repetitions after 2
    warm-ups; C1 (`forward` of 64 patches) and the process_VAE batch-of-one loop reported beside the headline workload."""
    from oracle import vqvae_oracle as O
    import statistics
    cpu_name, phys = _host_cpu()
    torch.manual_seed(0)
    nb = {"z32ex": 16, "c5": 4}.get(workload, 64)
    if workload == "c5":
        # the reference's distance tensor (vq_vae.py:65) is B*K*D*H*W floats = 268 MB per patch at K = 4096 on a 32 x 32
        # latent grid: the CPU path can only run this configuration in small chunks
        import numpy as np
        ref = O.OracleVQVAE(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4))
        x = torch.randn(nb, 4, 256, 256, generator=torch.Generator().manual_seed(1234))
    else:
        ref = O.OracleVQVAEz32(**EXAMPLE_CONFIG) if workload == "z32ex" else O.OracleVQVAE()
        x = torch.randn(nb, 2, 128, 128, generator=torch.Generator().manual_seed(1234))
    opt = O.make_adam(ref, 1e-4)

    def train_step():
        if workload == "z32ex":
            opt.zero_grad()
            ref(x)[1]["total_loss"].backward()
            opt.step()
        else:
            O.train_step(ref, opt, x)

    def latents():
        with torch.no_grad():
            O.encode_per_sample(ref, x)

    def forward():
        ref(x)

    headline = latents if workload == "c2" else train_step
    t_start = time.perf_counter()

    def timed(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)
    saved = torch.get_num_threads()
    tried = {}
    sweep = (16, 32) if workload == "c5" else (8, 16, 32, 64, phys)
    for n in sorted({t for t in sweep if 1 <= t <= (os.cpu_count() or 1)}):
        if time.perf_counter() - t_start > 0.5 * budget_s:
            break
        torch.set_num_threads(n)
        tried[n] = round(nb / timed(headline, 2 if workload == "c5" else 3, warm=1), 1)
    best = max(tried, key=tried.get) if tried else saved
    torch.set_num_threads(best)
    reps = 5 if workload == "c5" else 10
    med = timed(headline, reps, warm=1 if workload == "c5" else 2)
    rec = {"value": round(nb / med, 1), "unit": "patches/s", "cores": best, "kind": "port",
           "sample": f"median of {reps} x {'process_VAE batch-of-one loop (enc -> vq) over' if workload == 'c2' else 'training step (fwd+bwd+Adam) on a batch of'} "
                     f"{nb} patches, PyTorch CPU fp32, {best} threads (best of {sorted(tried)})",
           "cpu_model": cpu_name, "physical_cores": phys, "logical_cpus": os.cpu_count(), "threads_tried_patches_per_s": tried}
    if workload not in ("z32ex", "c5") and time.perf_counter() - t_start < budget_s:
        rec["c1_forward_b64_patches_per_s"] = round(nb / timed(forward, 10), 1)          # BASELINE.json configs[0]
        if workload != "c2" and time.perf_counter() - t_start < budget_s:
            rec["process_vae_loop_patches_per_s"] = round(nb / timed(latents, 5, warm=1), 1)
    torch.set_num_threads(saved)
    return rec


def main():
    args = parse()
    from dynamorph_amd import launch
    if args.gpus > 1 and not launch.launched():
        # started as a plain `python bench.py --gpus N`: this process stays off the GPU, starts N fresh ranks
        # (torch.distributed.run, one per GPU), forwards rank 0's JSON line and leaves with the job's exit code
        launch.check_devices(args.gpus)
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus))
    from dynamorph_amd import VQ_VAE
    from dynamorph_amd import dist as D
    from dynamorph_amd import engine as E
    from dynamorph_amd import ops
    from dynamorph_amd.train import FusedTrainer

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree")
    local = local % max(torch.cuda.device_count(), 1)      # (gloo rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.batch or {"c3": 2048, "c2": 1024, "c5": 1024, "z32ex": 768}[args.workload]

    torch.manual_seed(0)
    if args.workload == "c5":
        # BASELINE.json configs[4] / SURVEY 8(d): 4-channel 256 x 256 patches, 4096 codes
        import numpy as np
        x = torch.randn(B, 4, 256, 256, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
        model = VQ_VAE(num_inputs=4, num_embeddings=4096, channel_var=np.ones(4)).to(dev)
    else:
        x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    trainer, targs = None, ()
    if args.workload == "z32ex":
        # the variant, widths, batch and loss weights the reference's example configuration trains
        # (config_example.yml:156-186: VQ_VAE_z32 64 / 64 / 512, batch 768, weight_matching 100, margin 1, w_a 1, w_t 0.5,
        # w_n -0.5; the relation matrix marks adjacent frames of one trajectory: here 4 consecutive samples per group)
        if world != 1:
            raise SystemExit("--workload z32ex is a single-GPU measurement")
        from dynamorph_amd import VQ_VAE_z32
        model = VQ_VAE_z32(weight_matching=100., margin=1., w_a=1., w_t=0.5, w_n=-0.5, **EXAMPLE_CONFIG).to(dev)
        tm_mat = example_tm_matrix(B, dev)
        if args.no_fused:
            opt = torch.optim.Adam(model.parameters(), lr=1e-4)

            def step():
                _, ld = model(x, time_matching_mat=tm_mat)
                ld["total_loss"].backward()
                opt.step()
                model.zero_grad()
                return torch.stack([ld["recon_loss"].detach(), ld["commitment_loss"].detach(), ld["total_loss"].detach(),
                                    ld["perplexity"].detach()])
        else:
            trainer = FusedTrainer(model, lr=1e-4, use_graph=not args.no_graph)
            targs = (None, tm_mat)

            def step():
                return trainer.step(x, *targs)
    elif args.workload != "c5":
        model = VQ_VAE().to(dev)

    if args.workload == "z32ex":
        pass
    elif args.workload in ("c3", "c5"):
        trainer = FusedTrainer(model, lr=1e-4, use_graph=not args.no_graph)

        def step():
            return trainer.step(x)
    else:
        L = E.Layers(model)
        e1 = E.e1_operands(L)        # composite first-layer weights: once per model load (encode_patches: once per call)

        def step():
            with torch.no_grad():
                z_b, cx = E.encoder_forward(L, x, per_sample=True, e1=e1, join=False, latents_only=True)
                z_a, _, _ = E.vq_forward(L.codebook.weight, z_b, float(model.commitment_cost), want_scalars=False)
                cx.join()
            return z_a
        if not args.no_graph:
            step = graphed(step)

    if trainer is not None and not args.no_graph:
        # one-off set-up outside every counted step: capture the step's HIP graph (no optimisation step is taken) and move the
        # resident batch into the graph's input buffer, so that --warmup 0 still times K steps and nothing else
        buf = trainer.prepare(x, *targs)
        if buf is not None:
            x = buf
    for i in range(args.warmup):
        out = step()
        if i == 0 and trainer is not None and trainer.input_buffer() is not None:
            x = trainer.input_buffer()      # the batch now lives in the graph's input buffer: no per-step copy
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    # every rank's own wall time of the timed region (after the closing barrier they differ only by launch skew)
    rank_ms = D.gather_rank_values(1e3 * elapsed / args.steps, device=dev)
    elapsed = D.max_over_ranks(elapsed, device=dev)

    losses = out.tolist() if args.workload in ("c3", "c5", "z32ex") else None
    # evidence of the gradient exchange (after the timed region, every rank takes part): backend, world size and the
    # device / host time of the three parts of a step, from events on the launch stream around FusedTrainer._allreduce
    collective = None
    if trainer is not None:
        import torch.distributed as dist
        marks = []
        for _ in range(20 if world > 1 else 5):
            trainer.step(x, *targs, timers=marks)
        parts = FusedTrainer.timer_summary(marks)
        # backend, RCCL version, and that every rank holds bit-equal parameters after the steps above
        evidence = D.collective_evidence(trainer.flat)
        backend = evidence["backend"]
        collective = {"backend": backend, "world": world, "message_bytes": trainer.grad.numel() * 4,
                      "replicas_bit_equal": evidence["replicas_bit_equal"],
                      "allreduce_us": parts["allreduce_us"] if world > 1 else None,
                      "allreduce_host_us": parts["allreduce_host_us"] if world > 1 else None,
                      "fwd_bwd_graph_us": parts["fwd_bwd_us"], "adam_us": parts["adam_us"],
                      "fwd_bwd_host_us": parts["fwd_bwd_host_us"], "adam_host_us": parts["adam_host_us"],
                      "nccl_version": evidence["nccl_version"],
                      "rank_ms_per_step_min": round(min(rank_ms), 4), "rank_ms_per_step_max": round(max(rank_ms), 4),
                      "rank_ms_per_step": [round(v, 4) for v in rank_ms],
                      "op": ("all_reduce(SUM) of the flat fp32 gradient bucket; x 1/world inside the fused Adam's load"
                             if world > 1 else "none (one process)")}
        if world > 1:
            dist.barrier()
    if args.no_roofline or rank != 0:
        roof = None
    elif args.workload == "z32ex":
        roof = roofline_wide_conv(B)
    elif args.workload == "c5":
        roof = roofline_vq_large_codebook(model, x)
    else:
        roof = roofline_dominant_kernel(model, x, args.workload)
    targets = c2 = loop = c5 = z32 = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_targets:
        targets = north_star_targets(model, x)
        c2 = c2_record(model)
        loop = train_loop_record(dev, 1e3 * elapsed / args.steps, B=B)
        # the same entry point as run_training.py's main() drives it: cell masks and the relation matrix (time-matching term) on
        loop["with_masks_and_relation_matrix"] = {
            k: v for k, v in train_loop_record(dev, None, B=B, epochs=2, masks=True, relation=True).items()
            if k in ("train_patches_per_s", "train_ms_per_step", "loop_patches_per_s", "val_patches_per_s", "timed_epochs")}
        c5 = c5_record(dev)
        z32 = z32ex_record(dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload)

    if rank == 0:
        wl = {"c3": "C3: VQ_VAE training step (forward + backward + fused Adam), 2x128x128 fp32 synthetic patches",
              "c5": "C5: large-codebook stress, VQ_VAE(num_inputs=4, num_embeddings=4096) training step (forward + backward + fused "
                    "Adam), 4x256x256 fp32 synthetic patches",
              "c2": "C2: VQ_VAE inference latents (enc + vq, per-sample BatchNorm statistics = process_VAE), 2x128x128 fp32",
              "z32ex": "VQ_VAE_z32 as the reference's example configuration trains it (num_hiddens 64, num_residual_hiddens 64, 512 "
                       "codes, batch 768, time-matching term on with weight_matching 100): training step (forward + backward + "
                       "Adam), 2x128x128 fp32 synthetic patches"}[args.workload]
        line = {
            "metric": ("cell-patches/sec (256x256x4, 4096 codes) VQ-VAE fwd+bwd" if args.workload == "c5" else
                       "cell-patches/sec (128x128x2) VQ-VAE " + ("latent encoding" if args.workload == "c2" else "fwd+bwd")),
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "hip_graph": (not args.no_graph) if (args.workload in ("c3", "c2", "c5") or trainer is not None) else False,
                       "fused_trainer": trainer is not None,
                       "backward_products": ops.backward_precision()},
            "roofline": roof,
            "cpu_baseline": cpu,
            "collective": collective,
        }
        if targets is not None:
            line["targets"] = targets
            line["c2"] = c2
            line["train_loop"] = loop
            line["c5"] = c5
            line["z32ex"] = z32
        if losses is not None:
            line["final_losses"] = dict(zip(("recon", "commitment", "total", "perplexity"), [round(v, 6) for v in losses]))
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
