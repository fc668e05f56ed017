"""Two caller-side extra losses for VQ_VAE_z32(extra_loss=...) (vae.py:463-469): plain torch on the flattened latents,
`loss_fn(labels, z_after_flat) -> (loss, frac_pos)` -- the contract of the reference's triplet miners
(HiddenStateExtractor/losses.py) without their data-dependent triplet counts, so that a fixture pins smooth numbers.
Used by tests/golden/make_golden_z32_extra.py (through the reference's model class) and by the parity tests."""
import torch


def class_spread(labels, z):
    """Mean squared distance of every latent to the mean latent of its class."""
    loss = z.new_zeros(())
    classes = torch.unique(labels)
    for k in classes:
        zk = z[labels == k]
        loss = loss + (zk - zk.mean(0, keepdim=True)).pow(2).mean()
    return loss / classes.numel(), (labels > 0).float().mean()


def weighted_norm(labels, z):
    """Label-weighted mean square of the latents."""
    w = labels.to(z.dtype) + 1.0
    return (z.pow(2).mean(1) * w).mean(), z.new_tensor(0.5)


EXTRA = {"class_spread": class_spread, "weighted_norm": weighted_norm}
