"""Checkpoint tooling: a Lightning-style training checkpoint (codec tensors next to loss / EMA /
constraint-decoder tensors) -> the diffusers-style directory from_pretrained reads.  CPU only."""
import os

import pytest
import torch

import cvvae_amd
from cvvae_amd import checkpoint
from oracle.seeded import seeded_state_dict


SMALL = {"sd3": dict(block_out_channels=[32, 32, 64, 64]), "vae3d": dict(ch=32, ch_mult=[1, 1, 2, 2])}


@pytest.mark.parametrize("family", ["sd3", "vae3d"])
@pytest.mark.parametrize("use_ema", [False, True])
def test_convert_training_checkpoint(tmp_path, family, use_ema):
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    cfg = SMALL[family]  # a narrow network of the same structure keeps the file I/O of this test small
    shapes = {k: v.shape for k, v in cls(**cfg).state_dict().items()}
    live = seeded_state_dict(shapes, 3)
    ema = seeded_state_dict(shapes, 4)
    sd = {"first_stage_model." + k: v for k, v in live.items()}
    sd.update({"model_ema." + ("first_stage_model." + k).replace(".", ""): v for k, v in ema.items()})
    sd["loss.discriminator.main.0.weight"] = torch.zeros(4, 3, 3, 3, 3)          # training-only tensors are ignored
    sd["constraint_decoder.decoder.conv_in.weight"] = torch.zeros(8, 16, 3, 3)
    src = os.path.join(tmp_path, "last.ckpt")
    torch.save({"state_dict": sd, "global_step": 7}, src)
    dst = os.path.join(tmp_path, "out", "vae")
    checkpoint.convert_training_checkpoint(src, dst, family, use_ema=use_ema, prefix="first_stage_model.", **cfg)
    m = cls.from_pretrained(os.path.join(tmp_path, "out"), subfolder="vae", torch_dtype=torch.float16)
    want = ema if use_ema else live
    got = m.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k].to(torch.float16)), k


def test_missing_or_mismatched_tensors_raise(tmp_path):
    shapes = {k: v.shape for k, v in cvvae_amd.CVVAESD3Model(**SMALL["sd3"]).state_dict().items()}
    sd = seeded_state_dict(shapes, 0)
    sd.pop("decoder.conv_out.bias")
    src = os.path.join(tmp_path, "bad.ckpt")
    torch.save({"state_dict": sd}, src)
    with pytest.raises(KeyError, match="1 missing"):
        checkpoint.convert_training_checkpoint(src, os.path.join(tmp_path, "o"), "sd3", **SMALL["sd3"])


class Opaque:  # a non-tensor object as Lightning checkpoints carry them: not allow-listed for weights_only unpickling
    pass


def test_checkpoint_with_pickled_objects_gets_a_clear_error(tmp_path):
    """Lightning .ckpt files can carry arbitrary pickled objects; weights_only loading refuses them -- the tool must say what to do."""
    import pytest
    import torch
    from cvvae_amd.checkpoint import load_any

    p = tmp_path / "last.ckpt"
    torch.save({"state_dict": {"w": torch.zeros(1)}, "hyper_parameters": Opaque()}, p)
    with pytest.raises(RuntimeError, match="add_safe_globals"):
        load_any(str(p))
