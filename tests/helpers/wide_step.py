"""One forward + backward of VQ_VAE_z32 at the reference's example widths (64 / 64 / 512) on fixed inputs; losses and every
parameter gradient are written to the file named on the command line.  tests/test_gpu_model.py runs it twice in child
processes -- once as shipped, once with DM_WIDE_STREAM=0 DM_WIDE_WGRAD1=0 (the tiled kernels of rounds 1-5; the switches are read
once per process) -- and holds the two against each other."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd  # noqa: E402

torch.manual_seed(77)
B = 6
m = dynamorph_amd.VQ_VAE_z32(weight_matching=1.0, num_hiddens=64, num_residual_hiddens=64, num_embeddings=512).to("cuda")
x = torch.randn(B, 2, 128, 128, generator=torch.Generator().manual_seed(78)).cuda()
mask = (torch.rand(B, 1, 128, 128, generator=torch.Generator().manual_seed(79)) > 0.4).float().cuda()
tm = torch.randint(0, 3, (B, B), generator=torch.Generator().manual_seed(80)).float().cuda()
_, ld = m(x, time_matching_mat=tm, batch_mask=mask)
ld["total_loss"].backward()
torch.cuda.synchronize()
torch.save({"losses": {k: float(v) for k, v in ld.items()},
            "grads": {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None},
            "buffers": {k: v.cpu() for k, v in m.state_dict().items() if "running" in k}}, sys.argv[1])
