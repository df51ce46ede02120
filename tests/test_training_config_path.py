"""The reference's training YAML reaches the trainable 3-D networks unchanged.
configs/cvvae_sd3_constraint_training.yaml:10-37 instantiates `lvdm.modules.diffusionmodules.vae_models3d_sd3.Encoder3D` /
`Decoder3D` through `instantiate_from_config` (lvdm/util.py:168-185: importlib + `target(**params)`); the SD2.1-compatible family's
training path is `lvdm.modules.diffusionmodules.model_3d.Encoder` / `Decoder`.  Here: the YAML's two entries (params copied below,
block widths cut to three levels so that the emulated step stays in CPU-test time) go through the same call, the modules load a
seeded state dict STRICTLY under the reference's key names, and one training step -- forward, loss, backward, SGD update, second
forward -- runs with every kernel emulated (tests/emu_ops.py); gradients are checked against autograd over the oracle."""
import importlib

import pytest
import torch

from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from tests import emu_ops

# configs/cvvae_sd3_constraint_training.yaml:10-37, verbatim keys
YAML_ENCODER = dict(target="lvdm.modules.diffusionmodules.vae_models3d_sd3.Encoder3D", params=dict(
    in_channels=3, out_channels=16, down_block_types=["DownEncoderBlock3D"] * 4, block_out_channels=[128, 256, 512, 512],
    layers_per_block=2, norm_num_groups=32, act_fn="silu", double_z=True, mid_block_add_attention=True, causal=True, half_3d=True))
YAML_DECODER = dict(target="lvdm.modules.diffusionmodules.vae_models3d_sd3.Decoder3D", params=dict(
    in_channels=16, out_channels=3, up_block_types=["UpDecoderBlock3D"] * 4, block_out_channels=[128, 256, 512, 512],
    layers_per_block=2, norm_num_groups=32, act_fn="silu", mid_block_add_attention=True, causal=False, half_3d=True))


def instantiate_from_config(config):
    """lvdm/util.py:168-185"""
    module, cls = config["target"].rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)(**config.get("params", dict()))


def _rel(a, b, floor=0.0):
    return float((a - b).norm() / b.norm().clamp_min(floor if floor else 1e-30))


def test_yaml_entries_build_the_full_networks_with_the_reference_keys():
    """the full-size YAML entries: same parameter names and shapes as the wrapper's networks (= the reference's, pinned by
    tests/test_state_dict_contract.py), strict load of a seeded state dict"""
    import cvvae_amd
    enc, dec = instantiate_from_config(YAML_ENCODER), instantiate_from_config(YAML_DECODER)
    whole = cvvae_amd.CVVAESD3Model()
    for net, ref in ((enc, whole.encoder), (dec, whole.decoder)):
        a = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert a == b
        net.load_state_dict(seeded_state_dict(b, 3), strict=True)
        assert net._trainable and isinstance(net, type(ref))
    assert enc.get_last_layer() is enc.conv_out.weight and dec.get_last_layer() is dec.conv_out.weight


@pytest.mark.parametrize("key,val", [("norm_num_groups", 16), ("act_fn", "relu"), ("half_3d", False),
                                     ("block_out_channels", [96, 192, 384, 384])])
def test_unsupported_yaml_values_fail_at_construction(key, val):
    with pytest.raises(NotImplementedError):
        instantiate_from_config(dict(YAML_ENCODER, params=dict(YAML_ENCODER["params"], **{key: val})))


def test_one_emulated_training_step_through_the_yaml_modules():
    small = dict(block_out_channels=[128, 256, 256], layers_per_block=1)
    cfg_e = dict(YAML_ENCODER, params=dict(YAML_ENCODER["params"], down_block_types=["DownEncoderBlock3D"] * 3, **small))
    cfg_d = dict(YAML_DECODER, params=dict(YAML_DECODER["params"], up_block_types=["UpDecoderBlock3D"] * 3, **small))
    enc, dec = instantiate_from_config(cfg_e), instantiate_from_config(cfg_d)
    sd_e = seeded_state_dict({"encoder." + k: v.shape for k, v in enc.state_dict().items()}, 7)
    sd_d = seeded_state_dict({"decoder." + k: v.shape for k, v in dec.state_dict().items()}, 7)
    enc.load_state_dict({k[8:]: v for k, v in sd_e.items()}, strict=True)
    dec.load_state_dict({k[8:]: v for k, v in sd_d.items()}, strict=True)
    ref = {k: v.clone().requires_grad_(True) for k, v in {**sd_e, **sd_d}.items()}
    x = seeded_input((1, 3, 5, 16, 16), 21)
    # the step of lvdm/models/autoencoder.py:1057-1090 without the regulariser's noise: z = mode(encoder(x)), xrec = decoder(z)
    zr = O.sd3_encoder(x, ref, dict(small))[:, :16]
    yr = O.sd3_decoder(zr, ref, dict(small))
    loss_r = (yr - x).abs().mean() + 1e-3 * zr.pow(2).mean()
    loss_r.backward()
    with emu_ops.patched(whole_model=True):
        enc.train()
        dec.train()
        opt = torch.optim.SGD(list(enc.parameters()) + list(dec.parameters()), lr=1e-3)
        z = enc(x)[:, :16]
        y = dec(z)
        loss = (y - x).abs().mean() + 1e-3 * z.pow(2).mean()
        assert abs(loss.item() - loss_r.item()) <= 1e-5 * abs(loss_r.item())
        loss.backward()
        for pre, net in (("encoder.", enc), ("decoder.", dec)):
            names = [n for n, _ in net.named_parameters()]
            scale = max(float(ref[pre + n].grad.norm()) for n in names)
            for n, p in net.named_parameters():
                assert p.grad is not None, n
                assert _rel(p.grad, ref[pre + n].grad, 1e-3 * scale) < 5e-4, (pre + n, _rel(p.grad, ref[pre + n].grad))
        opt.step()
        # the update is seen by the next pass (the weight cache follows the optimiser's in-place writes)
        with torch.no_grad():
            ref2 = {k: (v - 1e-3 * v.grad).detach() for k, v in ref.items()}
            z2r = O.sd3_encoder(x, ref2, dict(small))[:, :16]
            z2 = enc(x)[:, :16]
        assert torch.allclose(z2, z2r, rtol=1e-4, atol=1e-5), float((z2 - z2r).abs().max())
        assert not torch.allclose(z2, z.detach(), rtol=1e-4, atol=1e-6)


def test_model_3d_path_of_the_sd21_family():
    """lvdm/modules/diffusionmodules/model_3d.py: Encoder / Decoder with the reference's constructor keys (the kwargs CVVAEModel
    passes, models/modeling_vae.py:53-82) and key names"""
    import cvvae_amd
    mod = importlib.import_module("lvdm.modules.diffusionmodules.model_3d")
    common = dict(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
                  z_channels=4, double_z=True, use_3d_conv=True, half_3d=True, resolution=256)
    enc = mod.Encoder(**common, attn_type="vanilla-xformers", causal=True)
    dec = mod.Decoder(**common, attn_type="spatial-temporal-xformer", causal=False)
    whole = cvvae_amd.CVVAEModel()
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == {k: tuple(v.shape) for k, v in whole.encoder.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == {k: tuple(v.shape) for k, v in whole.decoder.state_dict().items()}
    with pytest.raises(NotImplementedError):
        mod.Decoder(**common, attn_type="vanilla", causal=False)
    with pytest.raises(NotImplementedError):
        mod.Encoder(**dict(common, attn_resolutions=[32]), attn_type="vanilla", causal=True)
