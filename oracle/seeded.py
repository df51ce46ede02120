"""Deterministic, construction-order-independent synthetic weights.  TEST INFRASTRUCTURE.

Real CV-VAE checkpoints live on HuggingFace and cannot be fetched (no network), so every parity
statement in this repo is "random-weights parity".  To let golden vectors travel to the GPU box
without committing 180 M parameters, each tensor is generated from (key, shape, seed) alone:
the reference model (oracle/make_golden.py), the oracle and the HIP model all load the same dict.
Distributions mimic PyTorch default init (uniform +-1/sqrt(fan_in)) but give the affine norm
parameters non-trivial values so a swapped gamma/beta cannot pass.
"""
import zlib
import torch


def seeded_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) * 2654435761 + seed * 97 + 1) % (2**63 - 1))
    shape = tuple(shape)
    if key.endswith("weight") and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return (torch.rand(shape, generator=g) * 2 - 1) / fan_in ** 0.5
    if key.endswith("weight"):  # norm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)  # every bias / norm beta


def seeded_state_dict(shapes, seed: int = 0):
    """shapes: mapping key -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: seeded_tensor(k, tuple(s), seed) for k, s in sorted(shapes.items())}


def seeded_input(shape, seed: int = 0) -> torch.Tensor:
    """uniform [-1, 1) clip, as video/127.5-1 (cvvae_inference_video.py:34)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1234567 + seed)
    return torch.rand(tuple(shape), generator=g) * 2 - 1
