import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamorph_amd import ops
from dynamorph_amd._lib import DM_VQ_EXACT, DM_VQ_MFMA
dev = "cuda:0"
for (B, D, K, H, W) in [(64, 16, 64, 16, 16), (2048, 16, 64, 16, 16), (2, 16, 4096, 32, 32), (2, 16, 576, 16, 16)]:
    z = torch.randn(B, D, H, W, device=dev, generator=torch.Generator(dev).manual_seed(1))
    cb = torch.randn(K, D, device=dev, generator=torch.Generator(dev).manual_seed(2))
    ie = ops.vq_forward(z, cb, variant=DM_VQ_EXACT)[0].reshape(-1).cpu().numpy()
    im, om, _, _, nre = ops.vq_forward(z, cb, variant=DM_VQ_MFMA, want_rechecked=True)
    im = im.reshape(-1).cpu().numpy()
    bad = np.nonzero(ie != im)[0]
    print((B, D, K, H, W), "mismatches", len(bad), "of", len(ie), "rechecked", int(nre.cpu()))
    for p in bad[:24]:
        within = p % 64
        print("  pos", p, "chunk", p // 64, "c", within // 4, "t", within % 4, "exact", ie[p], "mfma", im[p],
              "diff", im[p] - ie[p], "bits exact", format(ie[p], "07b"), "mfma", format(im[p], "07b"))
    # distances check for first bad position
    if len(bad):
        p = bad[0]
        zf = z.permute(0, 2, 3, 1).reshape(-1, D)[p].cpu().double()
        d = ((zf[None] - cb.cpu().double()) ** 2).sum(1)
        o = torch.argsort(d)[:4]
        print("  true order", o.tolist(), d[o].tolist())
