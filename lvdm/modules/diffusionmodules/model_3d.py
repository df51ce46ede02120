"""`lvdm.modules.diffusionmodules.model_3d` -- the training-side import path of the SD2.1-compatible family's 3-D networks
(/root/reference/lvdm/modules/diffusionmodules/model_3d.py:562-886 = models/vae_models.py:679-1002 plus dead comments and the
`Decoder2` ablation): `Encoder` / `Decoder` with the reference's constructor keys and state-dict names, trainable
on the MI355X kernels (cvvae_amd/modeling.py, cvvae_amd/grad3d.py).  Config keys the kernels cannot honour raise at construction."""
from cvvae_amd import modeling as _m


def _check(ch, ch_mult, attn_resolutions, dropout, use_linear_attn, attn_type, ok_attn, use_3d_conv, half_3d, **more):
    bad = []
    if list(attn_resolutions or []):
        bad.append("attn_resolutions must be []")
    if dropout != 0.0:
        bad.append("dropout must be 0")
    if use_linear_attn or attn_type not in ok_attn:
        bad.append(f"attn_type {attn_type!r} (supported: {sorted(ok_attn)})")
    if not use_3d_conv or not half_3d:
        bad.append("use_3d_conv / half_3d must be True")
    for k, v in more.items():
        if v:
            bad.append(f"{k}={v!r}")
    bad += _m._channel_constraints([ch * m for m in ch_mult])
    if bad:
        raise NotImplementedError("model_3d on MI355X supports the shipped CV-VAE configuration only: " + "; ".join(bad))


class Encoder(_m.Encoder):
    """model_3d.py:562-706 (models/vae_models.py:679-823)"""

    def __init__(self, *, ch, out_ch=3, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution=None, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", use_3d_conv=True, half_3d=True, causal=True, half_t_mult=True, gradient_checkpointing=True,
                 **ignore_kwargs):
        _check(ch, ch_mult, attn_resolutions, dropout, use_linear_attn, attn_type, ("vanilla", "vanilla-xformers"), use_3d_conv,
               half_3d, **{"resamp_with_conv=False": not resamp_with_conv})
        super().__init__(ch=ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks, in_channels=in_channels, z_channels=z_channels,
                         double_z=double_z, causal=causal)


class Decoder(_m.Decoder):
    """model_3d.py:709-886 (models/vae_models.py:826-1002); the reference's `CVVAEModel` builds it with
    attn_type='spatial-temporal-xformer' (models/modeling_vae.py:68-82), the only decoder attention the kernels run"""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=None, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", use_3d_conv=True, half_3d=True, causal=True, half_t_mult=True,
                 gradient_checkpointing=True, **ignorekwargs):
        _check(ch, ch_mult, attn_resolutions, dropout, use_linear_attn, attn_type, ("spatial-temporal-xformer",), use_3d_conv,
               half_3d, **{"resamp_with_conv=False": not resamp_with_conv, "give_pre_end": give_pre_end, "tanh_out": tanh_out})
        super().__init__(ch=ch, out_ch=out_ch, ch_mult=ch_mult, num_res_blocks=num_res_blocks, z_channels=z_channels, causal=causal)
