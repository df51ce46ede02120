"""Drop-in for the reference's `models/modeling_vae.py`: `from models.modeling_vae import CVVAEModel, CVVAESD3Model`
(cvvae_inference_video.py:1, cvvae_sd3_inference_video.py:1) resolves to the MI355X implementation."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)

import cvvae_amd  # noqa: E402,F401
from cvvae_amd.modeling import (  # noqa: E402,F401
    AutoencoderKLCVVAE,
    AutoencoderKLOutput,
    CVVAEModel,
    CVVAESD3Model,
    DecoderOutput,
    DiagonalGaussianDistribution,
)
