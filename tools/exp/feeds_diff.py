"""Experiment: how far apart are train(feed=sync) and train(feed=resident) with masks + relation matrix?"""
import copy, sys, os, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import dynamorph_amd
from dynamorph_amd.train import train
from test_gpu_feed import _dataset
DEV = "cuda:0"
data, mask, rel = _dataset(53, 17, True, True)
torch.manual_seed(2)
m0 = dynamorph_amd.VQ_VAE().to(DEV)
got = {}
for feed in ("sync", "resident", "sync", "resident"):
    m = copy.deepcopy(m0)
    np.random.seed(123)
    with tempfile.TemporaryDirectory() as out:
        train(m, data, out, relation_mat=rel, mask=mask, n_epochs=int(os.environ.get("EPOCHS", "1")), lr=1e-3, batch_size=16, device=DEV,
              shuffle_data=False, transform=True, val_split_ratio=0.3, patience=10, feed=feed)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    if feed in got:
        print(feed, "repeat equal:", all(torch.equal(v, got[feed][k]) for k, v in sd.items()))
    got[feed] = sd
worst = max(((got["sync"][k].float() - got["resident"][k].float()).abs().max().item(), k) for k in got["sync"])
print("sync vs resident worst abs diff", worst)
