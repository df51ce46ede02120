"""The frozen 2-D encoder / decoder of the SD2.1-compatible family on the MI355X kernels (second half).

Reference: `Encoder`, `Decoder`, `EncoderWith3DWrapper`, `DecoderWith3DWrapper` of lvdm/modules/diffusionmodules/model.py:491-887 --
the LDM / Stable-Diffusion-2.1 image VAE halves (ResnetBlock with GroupNorm(32, eps 1e-6) + swish, 1x1 nin_shortcut, single-head
AttnBlock in the middle, Downsample = zero pad right/bottom + 3x3 stride 2, Upsample = nearest x2 + 3x3), wrapped so that 5-D
clips are coded frame by frame ('b c t h w -> (b t) c h w'), with the `quant_conv` / `post_quant_conv` 1x1 layers of the legacy
checkpoints.  The training engines hold them FROZEN (`self.constraint_encoder.requires_grad_(False)` under `torch.no_grad()`,
lvdm/models/autoencoder.py:1271-1284: `z_d = self.constraint_encoder(x[:, :, ::time_n_compress])`).  Same constructor keywords,
same state-dict names and shapes; forward only (no gradients: the reference calls the encoder under no_grad; the input gradient
of a constraint DECODER exists for the SD3 one only, cvvae_amd/grad.py).
"""
from typing import Sequence

import torch
import torch.nn as nn

from . import engine
from .modeling import ConvP, NormP, _channel_constraints, _holder, _Net


def _resblock(cin: int, cout: int) -> nn.Module:
    m = _holder(norm1=NormP(cin), conv1=ConvP(cin, cout, (3, 3)), norm2=NormP(cout), conv2=ConvP(cout, cout, (3, 3)))
    if cin != cout:
        m.add_module("nin_shortcut", ConvP(cin, cout, (1, 1)))
    return m


def _attn(c: int) -> nn.Module:
    return _holder(norm=NormP(c), q=ConvP(c, c, (1, 1)), k=ConvP(c, c, (1, 1)), v=ConvP(c, c, (1, 1)), proj_out=ConvP(c, c, (1, 1)))


def _check(ch, ch_mult, attn_resolutions, resamp_with_conv, attn_type, use_linear_attn, dropout, extra: str = ""):
    if attn_resolutions or not resamp_with_conv or use_linear_attn or attn_type not in ("vanilla", "vanilla-xformers") or dropout:
        raise NotImplementedError("the MI355X path covers the shipped SD2.1-family configuration: no attention inside the levels "
                                  "(attn_resolutions=[]), resamp_with_conv=True, vanilla single-head attention in the middle, "
                                  "dropout 0" + extra)
    return _channel_constraints([ch * m for m in ch_mult])


class Encoder(_Net):
    """model.py:491-614.  `forward(x)` takes images [N,C,H,W] (or, through the wrapper, clips)."""

    _program = staticmethod(engine.ldm2d_encoder)

    def __init__(self, *, ch, out_ch=3, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", gradient_checkpointing=True, **ignore_kwargs):
        super().__init__()
        bad = _check(ch, ch_mult, attn_resolutions, resamp_with_conv, attn_type, use_linear_attn, dropout)
        self._unsupported = bad[0] if bad else None
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.gradient_checkpointing, self.temb_ch = resolution, in_channels, gradient_checkpointing, 0
        self.conv_in = ConvP(in_channels, ch, (3, 3))
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        bi = ch
        for lvl in range(self.num_resolutions):
            bi, bo = ch * in_ch_mult[lvl], ch * ch_mult[lvl]
            blocks = nn.ModuleList()
            for _ in range(num_res_blocks):
                blocks.append(_resblock(bi, bo))
                bi = bo
            lv = _holder(block=blocks, attn=nn.ModuleList())
            if lvl != self.num_resolutions - 1:
                lv.add_module("downsample", _holder(conv=ConvP(bi, bi, (3, 3))))
            self.down.append(lv)
        self.mid = _holder(block_1=_resblock(bi, bi), attn_1=_attn(bi), block_2=_resblock(bi, bi))
        self.norm_out = NormP(bi)
        self.conv_out = ConvP(bi, 2 * z_channels if double_z else z_channels, (3, 3))
        self._cfg = dict(ch_mult=list(ch_mult), num_res_blocks=num_res_blocks)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError(f"expected a [N,C,H,W] tensor, got shape {tuple(x.shape)}")
        return _Net.forward(self, x.unsqueeze(2)).squeeze(2)


class Decoder(_Net):
    """model.py:617-772."""

    _program = staticmethod(engine.ldm2d_decoder)

    def __init__(self, *, ch, out_ch, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", gradient_checkpointing=True, **ignorekwargs):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not part of the shipped configuration")
        bad = _check(ch, ch_mult, attn_resolutions, resamp_with_conv, attn_type, use_linear_attn, dropout)
        self._unsupported = bad[0] if bad else None
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.give_pre_end, self.tanh_out = resolution, in_channels, give_pre_end, tanh_out
        self.gradient_checkpointing, self.temb_ch = gradient_checkpointing, 0
        bi = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = ConvP(z_channels, bi, (3, 3))
        self.mid = _holder(block_1=_resblock(bi, bi), attn_1=_attn(bi), block_2=_resblock(bi, bi))
        ups = [None] * self.num_resolutions
        for lvl in reversed(range(self.num_resolutions)):
            bo = ch * ch_mult[lvl]
            blocks = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                blocks.append(_resblock(bi, bo))
                bi = bo
            lv = _holder(block=blocks, attn=nn.ModuleList())
            if lvl != 0:
                lv.add_module("upsample", _holder(conv=ConvP(bi, bi, (3, 3))))
            ups[lvl] = lv
        self.up = nn.ModuleList(ups)
        self.norm_out = NormP(bi)
        self.conv_out = ConvP(bi, out_ch, (3, 3))
        self.last_z_shape = None
        self._cfg = dict(ch_mult=list(ch_mult), num_res_blocks=num_res_blocks)

    def forward(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        if z.dim() != 4:
            raise ValueError(f"expected a [N,C,H,W] tensor, got shape {tuple(z.shape)}")
        self.last_z_shape = z.shape
        return _Net.forward(self, z.unsqueeze(2)).squeeze(2)


class EncoderWith3DWrapper(Encoder):
    """model.py:832-887: clips [b,c,t,h,w] are encoded frame by frame; legacy=True adds `quant_conv` = Conv2d(2z, 2z, 1)."""

    def __init__(self, *, legacy=True, z_channels, **kw):
        super().__init__(z_channels=z_channels, **kw)
        self.legacy = legacy
        self.quant_conv = ConvP(2 * z_channels, 2 * z_channels, (1, 1)) if legacy else nn.Identity()

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.dim() == 5:
            return _Net.forward(self, x)
        return super().forward(x)


class DecoderWith3DWrapper(Decoder):
    """model.py:775-830: latents [b,c,t,h,w] are decoded frame by frame; legacy=True adds `post_quant_conv` = Conv2d(z, z, 1)."""

    def __init__(self, *, legacy=True, z_channels, **kw):
        super().__init__(z_channels=z_channels, **kw)
        self.legacy = legacy
        self.post_quant_conv = ConvP(z_channels, z_channels, (1, 1)) if legacy else nn.Identity()

    def forward(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        if z.dim() == 5:
            self.last_z_shape = z.shape
            return _Net.forward(self, z)
        return super().forward(z)
