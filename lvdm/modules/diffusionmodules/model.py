"""`lvdm.modules.diffusionmodules.model` as the reference's configs / engines name it (constraint_encoder_config /
constraint_decoder_config targets of lvdm/models/autoencoder.py:1271-1284, 1057-1069): the frozen 2-D halves of the SD2.1-compatible
image VAE on the MI355X kernels (cvvae_amd/constraint_ldm.py).  Only the classes on the codec's path exist here; the diffusion
`Model`, `LinAttnBlock`, cross-attention wrappers of that file are out of scope."""
from cvvae_amd.constraint_ldm import Decoder, DecoderWith3DWrapper, Encoder, EncoderWith3DWrapper  # noqa: F401
