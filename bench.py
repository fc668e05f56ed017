#!/usr/bin/env python3
"""bench.py -- cell-patches/s of the VQ-VAE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic 2x128x128 fp32 patches that is already
resident in HBM.  Default workload = BASELINE.json configs[2]/[3]: one TRAINING step (forward + backward +
Adam, batch 2048 per GPU; weak scaling, one RCCL all-reduce of the flat 96 KB gradient bucket per step).
`--workload c2` times configs[1] instead (inference latents, batch 1024, per-sample BatchNorm statistics).

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c3", "c2", "z32ex"], default="c3")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 2048 for c3, 1024 for c2, 256 for z32ex)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


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
    profiles/r01_pmc_traffic.json (tools/pmc_traffic.py writes it; bench.py cannot run under the profiler
    itself).  None when no measurement for this kernel and batch has been committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel_key)
    except (OSError, ValueError):
        return None
    if not rec or rec.get("batch") != batch:
        return None
    return rec.get("hbm_bytes_per_launch")


MFMA_F32_PEAK_TFLOPS = 157.3     # 256 CUs x 256 FLOP/clk (v_mfma_f32_16x16x4_f32) x 2.4 GHz, MI355X_MICROARCH.md

EXAMPLE_CONFIG = dict(num_hiddens=64, num_residual_hiddens=64, num_embeddings=512)     # config_example.yml:157-163


def roofline_wide_conv(B):
    """z32ex: the 3x3 64 -> 64 convolution of the residual blocks on the 32 x 32 latent (conv_wide_kernel<1, 9, 4>, the
    largest share of that step), alone, timed with events.  MFMA bound: achieved = 2*P*64*64*9 FLOP / duration."""
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
    return {"kernel": "wide_pack_kernel + conv_wide_kernel<1, 9, 4> (residual 3x3, 64 -> 64 channels, 32 x 32)", "bound": "mfma",
            "achieved": round(ach, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": None, "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops}


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
    ms = event_time_ms(fn)
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(key, B), "avg_launch_ms": round(ms, 4),
            "algorithmic_bytes_per_launch": algo_bytes}


def cpu_baseline(workload, budget_s=15.0):
    """The CPU oracle on a bounded sample of the same workload (train step / latents on 64-patch batches)."""
    from oracle import vqvae_oracle as O
    torch.manual_seed(0)
    nb = 16 if workload == "z32ex" else 64
    ref = O.OracleVQVAEz32(**EXAMPLE_CONFIG) if workload == "z32ex" else O.OracleVQVAE()
    x = torch.randn(nb, 2, 128, 128, generator=torch.Generator().manual_seed(1234))
    if workload == "z32ex":
        opt = O.make_adam(ref, 1e-4)

        def one():
            opt.zero_grad()
            ref(x)[1]["total_loss"].backward()
            opt.step()
        sample = "VQ_VAE_z32 (64/64/512) training step on batches of 16 patches, PyTorch CPU fp32"
    elif workload == "c3":
        opt = O.make_adam(ref, 1e-4)

        def one():
            O.train_step(ref, opt, x)
        sample = "training step (fwd+bwd+Adam) on batches of 64 patches, PyTorch CPU fp32"
    else:
        def one():
            with torch.no_grad():
                O.encode_per_sample(ref, x)
        sample = "process_VAE loop (batch-of-one enc->vq) over 64 patches, PyTorch CPU fp32"
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40:
            break
    return {"value": round(nb * n / el, 1), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} x {sample}; {os.cpu_count()} logical CPUs on host"}


def main():
    args = parse()
    from dynamorph_amd import VQ_VAE
    from dynamorph_amd import dist as D
    from dynamorph_amd import engine as E
    from dynamorph_amd.train import FusedTrainer

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = local % max(torch.cuda.device_count(), 1)      # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.batch or {"c3": 2048, "c2": 1024, "z32ex": 256}[args.workload]

    torch.manual_seed(0)
    x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    if args.workload == "z32ex":
        # the variant and widths the reference's example configuration trains; the
        # run_training.py loop as it is (autograd + torch.optim.Adam), every conv on the implicit-GEMM kernels
        if world != 1:
            raise SystemExit("--workload z32ex is a single-GPU measurement")
        from dynamorph_amd import VQ_VAE_z32
        model = VQ_VAE_z32(**EXAMPLE_CONFIG).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)

        def step():
            _, ld = model(x)
            ld["total_loss"].backward()
            opt.step()
            model.zero_grad()
            return torch.stack([ld["recon_loss"].detach(), ld["commitment_loss"].detach(), ld["total_loss"].detach(),
                                ld["perplexity"].detach()])
    else:
        model = VQ_VAE().to(dev)

    if args.workload == "z32ex":
        pass
    elif args.workload == "c3":
        trainer = FusedTrainer(model, lr=1e-4, use_graph=not args.no_graph)

        def step():
            return trainer.step(x)
    else:
        L = E.Layers(model)

        def step():
            with torch.no_grad():
                z_b, _ = E.encoder_forward(L, x, per_sample=True)
                z_a, _, _ = E.vq_forward(L.codebook.weight, z_b, float(model.commitment_cost))
            return z_a

    for i in range(args.warmup):
        out = step()
        if i == 0 and args.workload == "c3" and trainer.input_buffer() is not None:
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
    elapsed = D.max_over_ranks(elapsed, device=dev)

    losses = out.tolist() if args.workload in ("c3", "z32ex") else None
    if args.no_roofline or rank != 0:
        roof = None
    elif args.workload == "z32ex":
        roof = roofline_wide_conv(B)
    else:
        roof = roofline_dominant_kernel(model, x, args.workload)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload)

    if rank == 0:
        wl = {"c3": "C3: VQ_VAE training step (forward + backward + fused Adam), 2x128x128 fp32 synthetic patches",
              "c2": "C2: VQ_VAE inference latents (enc + vq, per-sample BatchNorm statistics = process_VAE), 2x128x128 fp32",
              "z32ex": "VQ_VAE_z32 with the reference's example widths (num_hiddens 64, num_residual_hiddens 64, 512 codes): "
                       "training step (forward + backward + torch Adam), 2x128x128 fp32 synthetic patches"}[args.workload]
        line = {
            "metric": "cell-patches/sec (128x128x2) VQ-VAE " + ("latent encoding" if args.workload == "c2" else "fwd+bwd"),
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "hip_graph": (not args.no_graph) if args.workload == "c3" else False},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if losses is not None:
            line["final_losses"] = dict(zip(("recon", "commitment", "total", "perplexity"), [round(v, 6) for v in losses]))
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
