"""The on-disk contract: the product's state_dict keys/shapes must equal the reference's.
oracle/shapes.py is the travelling copy of that table; when /root/reference is present it is re-derived live."""
import pytest
import torch

from oracle.ref_loader import load_reference, reference_available
from oracle.shapes import state_dict_shapes


def _product(family, **cfg):
    import cvvae_amd
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    return cls(**cfg)


@pytest.mark.parametrize("family", ["sd3", "vae3d"])
def test_product_keys_match_table(family):
    m = _product(family)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == state_dict_shapes(family)
    assert len(got) == (244 if family == "sd3" else 254)  # SURVEY.md 8b [probe]


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted (GPU box)")
@pytest.mark.parametrize("family", ["sd3", "vae3d"])
def test_table_matches_reference(family):
    ref = load_reference()
    cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
    with torch.device("meta"):
        m = cls()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == state_dict_shapes(family)
    # config keys and defaults (modeling_vae.py:26-50 / 350-380)
    mine = _product(family).config.to_dict()
    theirs = dict(m.config)
    assert set(mine) == set(theirs)
    for k in theirs:
        assert mine[k] == theirs[k], k


def test_from_pretrained_roundtrip(tmp_path):
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(tile_spatial_size=144)
    d = tmp_path / "ckpt" / "vae3d_sd3"
    m.save_pretrained(str(d))
    m2 = cvvae_amd.CVVAESD3Model.from_pretrained(str(tmp_path / "ckpt"), subfolder="vae3d_sd3", torch_dtype=torch.float16)
    assert m2.dtype == torch.float16 and m2.config.tile_spatial_size == 144 and not m2.training
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.half(), v2)
    assert m2.pixel_tile_size == 144 and m2.latent_tile_size == 18
    assert m2.encode_n_frames_a_time == 16 and m2.decode_n_frames_a_time == 4


def test_no_cpu_fallback():
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encoder(torch.zeros(1, 3, 1, 32, 32))


def test_constraint_decoder_keys_match_table_and_reference():
    """the frozen 2-D constraint decoder, instantiated through the reference's module path with the yaml's
    params (configs/cvvae_sd3_constraint_training.yaml:40-51): same state-dict keys and shapes as the table, and -- when the
    reference is mounted -- as the reference's own DecoderWith3DWrapper; unsupported configurations fail at construction."""
    from lvdm.modules.diffusionmodules.vae_models_sd3 import DecoderWith3DWrapper
    from oracle.golden_cases import CONSTRAINT_CFG
    from oracle.ref_loader import load_reference_constraint
    from oracle.shapes import constraint2d_shapes

    m = DecoderWith3DWrapper(**CONSTRAINT_CFG)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == constraint2d_shapes(CONSTRAINT_CFG) and len(got) == 138
    if reference_available():
        with torch.device("meta"):
            r = load_reference_constraint().DecoderWith3DWrapper(**CONSTRAINT_CFG)
        assert {k: tuple(v.shape) for k, v in r.state_dict().items()} == got
    with pytest.raises(NotImplementedError):
        DecoderWith3DWrapper(**dict(CONSTRAINT_CFG, norm_type="spatial"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 16, 2, 8, 8))


def test_from_pretrained_accepts_deprecated_attention_keys(tmp_path):
    """A checkpoint whose mid-block attention parameters carry diffusers' pre-0.18 names (query / key / value / proj_attn) loads
    like through diffusers' ModelMixin.from_pretrained (which renames them before its strict load)."""
    import json

    from safetensors.torch import save_file

    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model()
    d = tmp_path / "ckpt" / "vae3d_sd3"
    d.mkdir(parents=True)
    ren = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
    sd, n = {}, 0
    for k, v in m.state_dict().items():
        for new, old in ren.items():
            if ".attentions." in k and new in k:
                k = k.replace(new, old)
                n += 1
                break
        sd[k] = v.detach().clone()
    assert n == 16  # 2 mid blocks x 4 projections x (weight, bias)
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    (d / "config.json").write_text(json.dumps(dict(m.config.to_dict(), _class_name="CVVAESD3Model")))
    m2 = cvvae_amd.CVVAESD3Model.from_pretrained(str(tmp_path / "ckpt"), subfolder="vae3d_sd3")
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
