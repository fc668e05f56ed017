#!/usr/bin/env python3
"""Which device operation does torch.zeros((512, 64)) issue on this build -- a fill kernel or a memset?  Run under
rocprofv3 --kernel-trace --memory-copy-trace and read the traces (the round-4 failure of the atomic dm_vq_backward form
touched 1008 floats = 4 KiB - 64 B: page-shaped, which is how a DMA / memset engine writes, not how a kernel does)."""
import torch
x = torch.randn(1024, device="cuda:0")
torch.cuda.synchronize()
for _ in range(3):
    a = torch.zeros(512, 64, device="cuda:0")
    b = torch.zeros(64, 16, device="cuda:0")
    c = torch.empty(512, 64, device="cuda:0").zero_()
torch.cuda.synchronize()
print("done", float(a.sum() + b.sum() + c.sum()))
