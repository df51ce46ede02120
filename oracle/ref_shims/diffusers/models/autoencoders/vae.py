from dataclasses import dataclass
from typing import Optional
import torch
from ...utils import BaseOutput
from ...utils.torch_utils import randn_tensor


@dataclass
class DecoderOutput(BaseOutput):
    sample: torch.Tensor
    commit_loss: Optional[torch.Tensor] = None


class DiagonalGaussianDistribution:
    """diffusers posterior; identical maths in-tree at
    /root/reference/lvdm/modules/distributions/distributions.py:24-73."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator=None):
        noise = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device,
                             dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean
