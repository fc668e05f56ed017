#!/usr/bin/env python3
"""dec_tail_backward_kernel<2, true, false> (dm_dec_tail_train) alone on the model's own d2 at B = 2048, as bench.py's roofline
leg times it (ten launches in one HIP graph, events on the launch stream).  Run under the measurement library with
DM_DEC_TAIL_DBG set to switch parts of the kernel off: tools/exp/dec_tail_parts.sh drives it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynamorph_amd import VQ_VAE  # noqa: E402

B = int(os.environ.get("KB_B", "2048"))
torch.manual_seed(0)
dev = torch.device("cuda", 0)
x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(1234)).to(dev)
model = VQ_VAE().to(dev)
r = bench.roofline_dominant_kernel(model, x, "c3")
r2 = bench.roofline_dominant_kernel(model, x, "c3")
print(f"dbg={os.environ.get('DM_DEC_TAIL_DBG', '0'):>3}  {1e3 * min(r['avg_launch_ms'], r2['avg_launch_ms']):8.1f} us per launch "
      f"({r['avg_launch_ms'] * 1e3:.1f} / {r2['avg_launch_ms'] * 1e3:.1f})", flush=True)
