"""Parity at the BASELINE.json configurations' FULL sizes (cfg 1 vae3d 1x256^2 image, cfg 2 vae3d 17x256^2, cfg 3 sd3 17x512^2 --
the configuration the metric is quoted on -- and one 17-frame window of cfg 4 at 720x1280 with its 2x3 blended spatial tiles):
the HIP path against fixtures produced by the reference's OWN modules on the CPU in fp32 (oracle/make_golden.py big).  At
these sizes the library selects the benchmarked kernel instances (two-frame tiles, odd-frame split, short tiles last,
128-pixel tiles), so this is the parity check of exactly what bench.py times.

Every measurement is also appended to gpurun_out/parity_gpu.txt (copied to profiles/parity_r2.txt when kept)."""
import os

import pytest
import torch

from oracle import parity as P
from oracle.golden_cases import BIG_CASES

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# max / mean |delta| of the LATENT (posterior mean) and recon PSNR against the reference's fp32 outputs.  The bands are the
# reference's own low-precision noise with the same head-room as tests/test_gpu_model.py; the fp32
# model (3xfp16 split MFMA) must meet north_star's |delta| <= 1e-3 as a MAX.
# (the maximum over N latent values of a noise of fixed sigma grows like sqrt(2 ln N): the fp16 band of the small fixtures, 4e-3 at
# N ~ 1e4-1e5, becomes 5e-3 at the 1.15e6 latent values of the 720x1280 window; the mean band does not move)
TOL = {
    torch.float16: dict(latent_max=5.0e-3, latent_mean=8.0e-4, psnr=62.0),
    torch.bfloat16: dict(latent_max=3.5e-2, latent_mean=6.0e-3, psnr=45.0),
    torch.float32: dict(latent_max=1.0e-3, latent_mean=1.0e-4, psnr=80.0),
}
TAG = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}


def _log(line: str):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_gpu.txt"), "a") as f:
            f.write(line + "\n")


def _model(family, over, dtype, wseed):
    import cvvae_amd
    cls = cvvae_amd.CVVAESD3Model if family == "sd3" else cvvae_amd.CVVAEModel
    m = cls(**over)
    P.load_seeded(m, wseed)
    return m.to(dtype).cuda().eval()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("name", sorted(BIG_CASES))
def test_baseline_shape_golden(name, dtype, golden_dir):
    if not os.path.isfile(os.path.join(golden_dir, name + ".npz")):
        pytest.skip(f"fixture {name}.npz not generated")
    family, over, shape, wseed, xseed, s = BIG_CASES[name]
    from cvvae_amd import ops
    if dtype == torch.float32 and not getattr(ops, "SUPPORTS_FP32", False):
        pytest.skip("fp32 models (3xfp16 split MFMA) not built in this revision")
    m = _model(family, over, dtype, wseed)
    r = P.measure(m, name, golden_dir)
    line = P.fmt(TAG[dtype], r)
    print("\n" + line)
    _log(line)
    t = TOL[dtype]
    assert r["latent_max_abs"] <= t["latent_max"], line
    assert r["latent_mean_abs"] <= t["latent_mean"], line
    assert r["recon_psnr_db"] >= t["psnr"], line
    if dtype == torch.float16:
        assert r["latent_mean_abs"] <= 1.0e-3  # north_star's bound, met in the mean by the fp16 path
