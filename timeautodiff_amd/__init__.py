"""timeautodiff_amd: MI355X-native drop-in for TimeAutoDiff's hot path.

    from timeautodiff_amd import DIFF, VAE
    ae, latent, output, history = VAE.train_autoencoder(...)
    model = DIFF.train_diffusion(latent, None, data_dict, 200, 2, 100, n_epochs)
    samples = DIFF.sample(t, B, T, K, 100, model, time_info, cond, mask)

The compute path is libtad_hip.so (hand-written gfx950 kernels, C ABI in
include/tad_hip.h); importing the package does not load it, using it does.
"""
from . import DIFF, VAE  # noqa: F401

__all__ = ["DIFF", "VAE"]
