"""Large-codebook VectorQuantizer backward at the bench shapes (z32ex: 512 x 64, 768 x 32 x 32 positions; C5: 4096 x 16, 1024 x 32 x 32)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops
for K, D, B, H in ((512, 64, 768, 32), (4096, 16, 1024, 32)):
    torch.manual_seed(0)
    z = torch.randn(B, D, H, H, device="cuda"); cb = torch.randn(K, D, device="cuda")
    idx = torch.cdist(z.permute(0, 2, 3, 1).reshape(-1, D)[:65536], cb).argmin(1)
    idx = idx.repeat((B * H * H + 65535) // 65536)[:B * H * H].reshape(B, H, H).contiguous()
    g = torch.randn_like(z); gl = torch.ones(1, device="cuda")
    fn = lambda: ops.vq_backward_slabs(z, cb, idx, g, gl, 0.25)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"K {K} D {D} positions {B*H*H}: {dt*1e6:.1f} us per call (3 tensors of {z.numel()*4/1e6:.0f} MB: {3*z.numel()*4/dt/1e12:.2f} TB/s)")
