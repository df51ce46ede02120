"""`lvdm.modules.diffusionmodules.vae_models3d_sd3` as the reference's training config names it
(configs/cvvae_sd3_constraint_training.yaml:10-37: `encoder_config.target: lvdm.modules.diffusionmodules.vae_models3d_sd3.Encoder3D`,
`decoder_config.target: ...Decoder3D`; the file is a byte-identical copy of models/vae_models3d_sd3.py): the
TRAINABLE 3-D networks of the vae3d_sd3 family on the MI355X kernels -- `instantiate_from_config` of the YAML's two entries yields
modules with the reference's state-dict keys whose forward AND backward (input + every parameter gradient) run on the HIP extension
(cvvae_amd/modeling.py, cvvae_amd/grad3d.py).  Config keys the kernels cannot honour raise at construction, naming the key."""
from cvvae_amd import modeling as _m


def _check(kind, block_types, expected, norm_num_groups, act_fn, half_3d, block_out_channels):
    bad = []
    if block_types is not None and any(t != expected for t in block_types):
        bad.append(f"{kind}_block_types other than {expected}")
    if norm_num_groups != 32:
        bad.append("norm_num_groups != 32")
    if act_fn not in ("silu", "swish"):
        bad.append(f"act_fn {act_fn!r}")
    if not half_3d:
        bad.append("half_3d=False")
    bad += _m._channel_constraints(block_out_channels)
    if bad:
        raise NotImplementedError("vae_models3d_sd3 on MI355X supports the shipped CV-VAE configuration only: " + "; ".join(bad))


class Encoder3D(_m.Encoder3D):
    """models/vae_models3d_sd3.py:55-208 (constructor keys :78-92)"""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock3D",), block_out_channels=(64,),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True, mid_block_add_attention=True, causal=True,
                 half_3d=True, **ignored):
        _check("down", down_block_types, "DownEncoderBlock3D", norm_num_groups, act_fn, half_3d, block_out_channels)
        if down_block_types is not None and len(down_block_types) != len(block_out_channels):
            raise ValueError("down_block_types and block_out_channels differ in length")
        super().__init__(in_channels=in_channels, out_channels=out_channels, block_out_channels=block_out_channels,
                         layers_per_block=layers_per_block, double_z=double_z, mid_block_add_attention=mid_block_add_attention,
                         causal=causal)


class Decoder3D(_m.Decoder3D):
    """models/vae_models3d_sd3.py:211-391 (constructor keys :232-246)"""

    def __init__(self, in_channels=3, out_channels=3, up_block_types=("UpDecoderBlock3D",), block_out_channels=(64,),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", norm_type="group", mid_block_add_attention=True,
                 causal=False, half_3d=True, **ignored):
        _check("up", up_block_types, "UpDecoderBlock3D", norm_num_groups, act_fn, half_3d, block_out_channels)
        if norm_type != "group":
            raise NotImplementedError("vae_models3d_sd3.Decoder3D on MI355X: norm_type must be 'group'")
        if up_block_types is not None and len(up_block_types) != len(block_out_channels):
            raise ValueError("up_block_types and block_out_channels differ in length")
        super().__init__(in_channels=in_channels, out_channels=out_channels, block_out_channels=block_out_channels,
                         layers_per_block=layers_per_block, mid_block_add_attention=mid_block_add_attention, causal=causal)
