"""The CPU oracle (oracle/cvvae_oracle.py) must reproduce the golden vectors that were produced by the
reference's own modules (oracle/make_golden.py).  Runs everywhere (no GPU, no /root/reference needed)."""
import os

import numpy as np
import pytest
import torch

from oracle import cvvae_oracle as O
from oracle.golden_cases import CASES
from oracle.seeded import seeded_input, seeded_state_dict
from oracle.shapes import state_dict_shapes

# fp32 CPU vs fp32 CPU, different op order (explicit softmax vs SDPA, folded frames): rounding-level only
TOL_MOMENTS = 2e-5
TOL_RECON = 1e-4


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_golden(name, golden_dir):
    family, over, shape, wseed, xseed = CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = dict(over)
    sd = seeded_state_dict(state_dict_shapes(family, cfg), wseed)
    assert len(sd) == int(gold["n_tensors"])
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(gold["weight_abs_sum"])) < 1e-6 * wsum, "oracle/seeded.py drifted from the fixtures"
    x = seeded_input(shape, xseed)
    with torch.no_grad():
        mom = O.encode_moments(x, sd, cfg, family)
        rec = O.decode_sample(O.posterior_mode(mom), sd, cfg, family)
    assert tuple(mom.shape) == gold["moments"].shape and tuple(rec.shape) == gold["recon"].shape
    assert np.abs(mom.numpy() - gold["moments"]).max() <= TOL_MOMENTS
    assert np.abs(rec.numpy() - gold["recon"]).max() <= TOL_RECON


from oracle.golden_cases import CONSTRAINT_CASES  # noqa: E402
from oracle.shapes import constraint2d_shapes  # noqa: E402


@pytest.mark.parametrize("name", sorted(CONSTRAINT_CASES))
def test_oracle_constraint_decoder_matches_reference_golden(name, golden_dir):
    """the frozen 2-D constraint decoder (DecoderWith3DWrapper) -- the oracle restatement against fixtures
    produced by the reference's own module (oracle/make_golden.py constraint)."""
    cfg, zshape, wseed, zseed = CONSTRAINT_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = seeded_state_dict(constraint2d_shapes(cfg), wseed)
    assert len(sd) == int(gold["n_tensors"])
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(gold["weight_abs_sum"])) < 1e-6 * wsum
    with torch.no_grad():
        rec = O.constraint_decoder(seeded_input(zshape, zseed), sd, cfg)
    assert tuple(rec.shape) == gold["recon"].shape
    assert np.abs(rec.numpy() - gold["recon"]).max() <= TOL_RECON


from oracle.golden_cases import BIG_CASES, recon_subsample  # noqa: E402


@pytest.mark.parametrize("name", sorted(BIG_CASES))
def test_baseline_size_fixtures(name, golden_dir):
    """BASELINE-size fixtures (oracle/make_golden.py big): integrity of every fixture against oracle/seeded.py and the shape
    laws; the oracle restatement itself is re-run against the cfg 1 fixture (seconds of CPU; the larger ones cost minutes and
    are the GPU tests' yardstick)."""
    path = os.path.join(golden_dir, name + ".npz")
    if not os.path.isfile(path):
        pytest.skip(f"{name}.npz not generated")
    family, over, shape, wseed, xseed, s = BIG_CASES[name]
    gold = np.load(path)
    sd = seeded_state_dict(state_dict_shapes(family, dict(over)), wseed)
    assert len(sd) == int(gold["n_tensors"])
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(gold["weight_abs_sum"])) < 1e-6 * wsum
    B, _, T, H, W = shape
    zc = 32 if family == "sd3" else 8
    mshape = gold["moments"].shape if "moments" in gold else tuple(int(v) for v in gold["moments_shape"])
    assert mshape == (B, zc, 1 + (T - 1) // 4, H // 8, W // 8)
    assert tuple(gold["recon_shape"]) == shape and int(gold["recon_stride"]) == s
    assert gold["recon_sub"].shape == (B, 3, T, H // s, W // s)
    assert np.isfinite(gold["moments"] if "moments" in gold else gold["moments_mean"]).all() and np.isfinite(gold["recon_sub"]).all()
    if name.startswith("cfg1"):
        x = seeded_input(shape, xseed)
        with torch.no_grad():
            mom = O.encode_moments(x, sd, dict(over), family)
            rec = O.decode_sample(O.posterior_mode(mom), sd, dict(over), family)
        assert np.abs(mom.numpy() - gold["moments"]).max() <= TOL_MOMENTS
        assert np.abs(recon_subsample(rec, s).numpy() - gold["recon_sub"]).max() <= TOL_RECON
        assert abs(float(rec.double().mean()) - float(gold["recon_mean"])) <= 1e-5
