"""`lvdm.modules.diffusionmodules.vae_models_sd3` as the reference's configs name it
(configs/cvvae_sd3_constraint_training.yaml:41: `target: lvdm.modules.diffusionmodules.vae_models_sd3.DecoderWith3DWrapper`):
the frozen 2-D constraint decoder on the MI355X kernels (cvvae_amd/constraint.py).  The 2-D `Encoder` of that file is not part
of the path (it is not instantiated by any shipped config)."""
from cvvae_amd.constraint import Decoder, DecoderWith3DWrapper  # noqa: F401
