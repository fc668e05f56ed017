#!/usr/bin/env python3
"""train() end to end: 32 768 synthetic host patches, batch 2048, 3 timed epochs, augmentation on.
One JSON line per feed: the resident one (dataset in HBM, gather + augment kernel into the captured step's input buffer),
the streaming one (pinned staging + copy stream, PCIe-bound) and the reference's synchronous loop, each next to the rate of
the resident-batch bench step measured in the same process.

    python tools/trainbench.py [--n 32768] [--batch 2048] [--epochs 3] [--feeds resident,stream,sync] [--full]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def resident_step_ms(dev, B, full=False, steps=100):
    """The bench's step on one batch resident in HBM; full: with a mask and a relation block (the time-matching term on)."""
    from dynamorph_amd import VQ_VAE
    from dynamorph_amd.train import FusedTrainer
    torch.manual_seed(0)
    tr = FusedTrainer(VQ_VAE().to(dev), lr=1e-4)
    x = torch.randn(B, 2, 128, 128, device=dev)
    mask = tm = None
    if full:
        mask = (torch.rand(B, 1, 128, 128, device=dev) > 0.3).float()
        i = torch.arange(B - 1, device=dev)
        same = (i // 8) == ((i + 1) // 8)
        tm = torch.zeros(B, B, device=dev)
        tm[i[same], i[same] + 1] = 2.0
        tm[i[same] + 1, i[same]] = 2.0
    tr.prepare(x, mask, tm)
    x, mask, tm = tr.static_inputs(x.shape, None if mask is None else mask.shape, None if tm is None else tm.shape)
    for _ in range(10):
        tr.step(x, mask, tm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, mask, tm)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--feeds", default="resident,stream,sync")
    ap.add_argument("--full", action="store_true", help="also with masks and a relation matrix (the real run_training.py loop)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    base = {}
    for full in ((False, True) if a.full else (False,)):
        base[full] = resident_step_ms(dev, a.batch, full)
        print(json.dumps({"resident_bench": "with mask + time-matching term" if full else "plain",
                          "ms_per_step": round(base[full], 4), "patches_per_s": round(a.batch / base[full] * 1e3, 1)}), flush=True)
    for feed in a.feeds.split(","):
        for full in ((False, True) if a.full else (False,)):
            for pinned in ((False, True) if feed == "stream" else (False,)):
                rec = bench.train_loop_record(dev, base[full], n=a.n, B=a.batch, epochs=a.epochs, feed=feed, masks=full,
                                              relation=full, pinned=pinned)
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
