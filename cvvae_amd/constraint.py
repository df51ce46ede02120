"""The frozen 2-D "constraint" decoder of CV-VAE's training path on the MI355X kernels.

Reference: `DecoderWith3DWrapper(Decoder)` in lvdm/modules/diffusionmodules/vae_models_sd3.py:196-398 (blocks:
lvdm/modules/diffusionmodules/vae_blocks_sd3.py) -- the SD3 image VAE decoder, frozen
(`self.constraint_decoder.requires_grad_(False)`, lvdm/models/autoencoder.py:1057-1058), applied frame by frame to the 3-D VAE's
latents to produce `xrec_2d` for the latent-compatibility loss (autoencoder.py:1069; configs/cvvae_sd3_constraint_training.yaml:40-51).
Same constructor keywords, same state-dict names and shapes.  The decoder is never trained, but the loss back-propagates THROUGH
it into the latents: when the input requires grad (and autograd is enabled) the call is recorded as one autograd node whose
backward is the frozen decoder's input gradient on the same kernels (cvvae_amd/grad.py).  Weight gradients are not built.
"""
from typing import Tuple

import torch
import torch.nn as nn

from . import engine
from .modeling import ConvP, NormP, _channel_constraints, _holder, _Net


def _resnet2d(cin: int, cout: int) -> nn.Module:
    m = _holder(norm1=NormP(cin), conv1=ConvP(cin, cout, (3, 3)), norm2=NormP(cout), conv2=ConvP(cout, cout, (3, 3)))
    if cin != cout:
        m.add_module("conv_shortcut", ConvP(cin, cout, (1, 1)))
    return m


class Decoder(_Net):
    """vae_models_sd3.py:196-362 (eval path).  `forward(sample)` takes [N, C, H, W]."""

    _program = staticmethod(engine.constraint_decoder2d)

    def __init__(self, in_channels: int = 3, out_channels: int = 3, up_block_types: Tuple[str, ...] = ("UpDecoderBlock2D",),
                 block_out_channels: Tuple[int, ...] = (64,), layers_per_block: int = 2, norm_num_groups: int = 32,
                 act_fn: str = "silu", norm_type: str = "group", mid_block_add_attention=True):
        super().__init__()
        boc = list(block_out_channels)
        if (any(t != "UpDecoderBlock2D" for t in up_block_types) or len(up_block_types) != len(boc) or norm_num_groups != 32
                or act_fn != "silu" or norm_type != "group" or any(c % 32 for c in boc)):
            raise NotImplementedError("the MI355X path covers the shipped constraint decoder: UpDecoderBlock2D blocks, "
                                      "GroupNorm(32) + SiLU, channel counts that are multiples of 32")
        bad = _channel_constraints(boc)  # buildable / loadable, but a forward pass raises this message (class default (64,))
        self._unsupported = bad[0] if bad else None
        self.layers_per_block = layers_per_block
        rev = list(reversed(boc))
        top = rev[0]
        self.conv_in = ConvP(in_channels, top, (3, 3))
        resnets = nn.ModuleList([_resnet2d(top, top), _resnet2d(top, top)])
        atts = nn.ModuleList()
        if mid_block_add_attention:
            a = _holder(group_norm=NormP(top), to_q=ConvP(top, top, ()), to_k=ConvP(top, top, ()), to_v=ConvP(top, top, ()))
            a.add_module("to_out", nn.ModuleList([ConvP(top, top, ()), nn.Identity()]))
            atts.append(a)
        self.mid_block = _holder(attentions=atts, resnets=resnets)
        self.up_blocks = nn.ModuleList()
        ch = top
        for i, co in enumerate(rev):
            blk = _holder(resnets=nn.ModuleList([_resnet2d(ch if j == 0 else co, co) for j in range(layers_per_block + 1)]))
            ch = co
            if i != len(rev) - 1:
                blk.add_module("upsamplers", nn.ModuleList([_holder(conv=ConvP(co, co, (3, 3)))]))
            self.up_blocks.append(blk)
        self.conv_norm_out = NormP(boc[0])
        self.conv_out = ConvP(boc[0], out_channels, (3, 3))
        self.gradient_checkpointing = True  # attribute of the reference class (vae_models_sd3.py:295); unused: forward only
        self._cfg = dict(block_out_channels=boc, layers_per_block=layers_per_block,
                         mid_block_add_attention=bool(mid_block_add_attention))

    def _run(self, z: torch.Tensor) -> torch.Tensor:
        """z [b,c,t,h,w] -> pixels; differentiable w.r.t. z when z requires grad (input gradient only: the module is frozen)"""
        if torch.is_grad_enabled() and z.requires_grad:
            self._check_input(z)
            if any(p.requires_grad for p in self.parameters()):
                raise NotImplementedError("only the frozen decoder's input gradient is built (the reference calls "
                                          "constraint_decoder.requires_grad_(False)); weight gradients are not")
            from .grad import ConstraintDecoderFn
            from .modeling import _autocast_dtype
            with self._cache().computing_in(_autocast_dtype(z) if self.conv_in.weight.dtype == torch.float32 else None):
                return ConstraintDecoderFn.apply(z, self)
        return _Net.forward(self, z)

    def forward(self, sample: torch.Tensor, latent_embeds=None) -> torch.Tensor:
        if latent_embeds is not None:
            raise NotImplementedError("norm_type='spatial' (latent_embeds) is not part of the shipped configuration")
        if sample.dim() != 4:
            raise ValueError(f"expected a [N,C,H,W] tensor, got shape {tuple(sample.shape)}")
        return self._run(sample.unsqueeze(2)).squeeze(2)


class DecoderWith3DWrapper(Decoder):
    """vae_models_sd3.py:365-398: 5-D latents [b,c,t,h,w] are decoded frame by frame, 4-D ones as images."""

    def forward(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        if z.dim() == 5:
            if kwargs.get("latent_embeds") is not None:
                raise NotImplementedError("norm_type='spatial' (latent_embeds) is not part of the shipped configuration")
            return self._run(z)
        return super().forward(z, **kwargs)
