import torch


class RMSNorm(torch.nn.Module):  # never instantiated on the CV-VAE path (norm_type=None)
    def __init__(self, *a, **k):
        raise NotImplementedError("RMSNorm is unused by CV-VAE")
