"""Is the large-codebook codebook gradient bit-reproducible from launch to launch?  (csrc/vq.hip, vq_backward_kernel: LDS / global
float atomics.)  Prints the number of elements that differ between repeated launches on the same inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamorph_amd
from dynamorph_amd.vq_vae import VectorQuantizer
torch.manual_seed(0)
for K, D, B, H in ((512, 64, 3, 32), (512, 64, 64, 32), (4096, 16, 4, 64), (64, 16, 64, 32)):
    vq = VectorQuantizer(D, K, 0.25).cuda()
    z = torch.randn(B, D, H, H, device="cuda")
    grads = []
    for _ in range(4):
        vq.zero_grad()
        zz = z.clone().requires_grad_(True)
        out = vq(zz)
        (out[0].square().mean() + out[1]).backward()
        grads.append(vq.w.weight.grad.clone())
    nd = [int((grads[0] != g).sum()) for g in grads[1:]]
    md = [float((grads[0] - g).abs().max()) for g in grads[1:]]
    print(f"K {K} D {D} B {B}: elements differing from the first launch {nd}, largest difference {max(md):.2e} (gradient scale {float(grads[0].abs().max()):.2e})")
