"""dynamorph_amd -- MI355X-native VQ-VAE latent-encoding path of mehta-lab/dynamorph.

Public surface = the reference's module surface for this path:
    VQ_VAE, VQ_VAE_z16, VQ_VAE_z32, VectorQuantizer, ResidualBlock      (dynamorph_amd.vq_vae)
computed by hand-written gfx950 HIP kernels behind the C ABI in include/dynamorph_hip.h.
"""
from .vq_vae import VQ_VAE, VQ_VAE_z16, VQ_VAE_z32, VectorQuantizer, ResidualBlock  # noqa: F401

__all__ = ["VQ_VAE", "VQ_VAE_z16", "VQ_VAE_z32", "VectorQuantizer", "ResidualBlock"]
