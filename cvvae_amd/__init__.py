"""cvvae_amd: MI355X-native (gfx950) implementation of the CV-VAE latent video codec forward pass.

Host side = Python on PyTorch-ROCm (memory, streams, torch.distributed); compute = hand-written HIP kernels in
libcvvae_hip.so behind the C ABI of include/cvvae.h.  (`cv-vae_amd/` at the repo root is a symlink to this directory: the
hyphenated name is not a Python identifier.)"""
from . import _lib, ops  # noqa: F401
from .modeling import AutoencoderKLCVVAE, CVVAEModel, CVVAESD3Model  # noqa: F401

__all__ = ["_lib", "ops", "CVVAEModel", "CVVAESD3Model", "AutoencoderKLCVVAE"]
