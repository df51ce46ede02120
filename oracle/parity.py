"""Parity measurement of a device model against the reference-generated fixtures (tests/golden).  TEST INFRASTRUCTURE:
imported by tests/, __graft_entry__.smoke() and bench.py's parity leg only -- as the checker, never on the product path.

Metrics follow SURVEY 8(d): latent max / mean |delta| on `moments`, recon PSNR = 10 log10(4 / MSE) on [-1, 1] data.
"""
import os

import numpy as np
import torch

from .golden_cases import BIG_CASES, CASES, DEC_CASES, ENC_CASES, recon_subsample
from .seeded import seeded_input, seeded_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# the reference's OWN low-precision noise (its fp16 / bf16 CPU run vs its fp32 run; T=9, 96x96)
REFERENCE_SELF_NOISE = {
    "f16": {"latent_max": 3.2e-3, "latent_mean": 6.5e-4, "recon_psnr_db": 65.8},
    "bf16": {"latent_max": 2.8e-2, "latent_mean": 5.3e-3, "recon_psnr_db": 47.5},
}


def reference_self_noise(case: str, tag: str, golden_dir: str = GOLDEN_DIR):
    """how far the reference's OWN fp16 / bf16 CPU run is from its fp32 run on fixture `case` (tests/golden/ref_self_noise.json,
    written by oracle/make_noise.py with the metric of measure() below); the T=9 96x96 probe above when that shape was not run"""
    import json
    path = os.path.join(golden_dir, "ref_self_noise.json")
    if os.path.isfile(path):
        with open(path) as f:
            e = json.load(f).get(case, {}).get(tag)
        if e:
            return {"latent_max": e["latent_max_abs"], "latent_mean": e["latent_mean_abs"], "recon_psnr_db": e["recon_psnr_db"],
                    "recon_max": e["recon_max_abs"], "shape": e["shape"], "source": "tests/golden/ref_self_noise.json"}
    e = REFERENCE_SELF_NOISE.get(tag)
    return dict(e, source="BASELINE.md section 2 (T=9, 96x96 probe)") if e else None


def load_seeded(model, wseed: int):
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
    model.load_state_dict(sd, strict=True)
    return sd


def case_of(name: str):
    """-> (family, overrides, shape, wseed, xseed, recon stride or 0 for a fully stored recon)"""
    if name in BIG_CASES:
        return BIG_CASES[name]
    return CASES[name] + (0,)


@torch.no_grad()
def measure(model, name: str, golden_dir: str = GOLDEN_DIR) -> dict:
    """`model`: a device model that already carries seeded_state_dict(shapes, wseed of the case), in its run dtype.
    encode(x) is compared on `moments`; decode() is run on the REFERENCE's latent so the decoder is judged on identical
    input.  Returns plain floats."""
    family, over, shape, wseed, xseed, s = case_of(name)
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    dtype, dev = model.dtype, model.device
    x = seeded_input(shape, xseed).to(dtype).to(dev)
    mom = model.encode(x).latent_dist.parameters.float().cpu().numpy()
    zc = mom.shape[1] // 2
    if "moments" in gold:
        gm = gold["moments"]
        assert mom.shape == gm.shape, (mom.shape, gm.shape)
        gmean = gm[:, :zc]
        d = np.abs(mom - gm)
    else:  # large fixtures keep the posterior mean in full and the log-variance at stride 2 over H and W
        assert tuple(mom.shape) == tuple(int(v) for v in gold["moments_shape"])
        gmean = gold["moments_mean"]
        d = np.concatenate([np.abs(mom[:, :zc] - gmean).ravel(),
                            np.abs(mom[:, zc:, :, ::2, ::2] - gold["moments_logvar_sub"]).ravel()])
    z = torch.from_numpy(gmean).to(dtype).to(dev)
    rec = model.decode(z).sample.float().cpu()
    if s:
        assert tuple(rec.shape) == tuple(int(v) for v in gold["recon_shape"])
        r, g = recon_subsample(rec, s).numpy(), gold["recon_sub"]
        extra = {"recon_mean_delta": float(abs(rec.double().mean().item() - float(gold["recon_mean"]))),
                 "recon_sampled_fraction": round(1.0 / (s * s), 4)}
    else:
        r, g = rec.numpy(), gold["recon"]
        extra = {}
    assert r.shape == g.shape, (r.shape, g.shape)
    dm = np.abs(mom[:, :zc] - gmean)  # the latent proper (posterior mean); `moments` also holds logvar
    mse = float(((r - g).astype(np.float64) ** 2).mean())
    out = {
        "case": name, "shape": list(shape),
        "latent_max_abs": float(dm.max()), "latent_mean_abs": float(dm.mean()),
        "moments_max_abs": float(d.max()), "moments_mean_abs": float(d.mean()),
        "recon_max_abs": float(np.abs(r - g).max()), "recon_psnr_db": float(10 * np.log10(4.0 / max(mse, 1e-30))),
    }
    out.update(extra)
    return out


@torch.no_grad()
def measure_encode(model, name: str, golden_dir: str = GOLDEN_DIR, latents=None) -> dict:
    """encode-only fixtures (golden_cases.ENC_CASES: BASELINE cfg 5's batch slice) against the reference's `moments`:
    `model.encode(x).latent_dist.parameters` (mean and log-variance), or -- `latents` given: the latent pre-compute entry point
    bench.py times, x -> posterior MODE -- the posterior mean alone."""
    family, over, shape, wseed, xseed = ENC_CASES[name][:5]
    ms = ENC_CASES[name][5] if len(ENC_CASES[name]) > 5 else 1  # stride of the stored mean (per-frame phase: recon_subsample)
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    x = seeded_input(shape, xseed).to(model.dtype).to(model.device)
    mshape = tuple(int(v) for v in gold["moments_shape"])
    zc = mshape[1] // 2
    if latents is not None:
        mean = latents(x).float().cpu().numpy()
        assert tuple(mean.shape) == (mshape[0], zc) + mshape[2:], (mean.shape, mshape)
        dl = None
    else:
        mom = model.encode(x).latent_dist.parameters.float().cpu().numpy()
        assert tuple(mom.shape) == mshape, (mom.shape, mshape)
        mean = mom[:, :zc]
        lv = mom[:, zc:, :, ::2, ::2] if ms == 1 else recon_subsample(mom[:, zc:], 2 * ms)
        dl = np.abs(lv - gold["moments_logvar_sub"])
    dm = np.abs((mean if ms == 1 else recon_subsample(mean, ms)) - gold["moments_mean"])
    out = {"case": name, "shape": list(shape), "latent_max_abs": float(dm.max()), "latent_mean_abs": float(dm.mean()),
           "latent_max_abs_per_batch_item": [float(dm[b].max()) for b in range(dm.shape[0])]}
    if dl is not None:
        out["moments_max_abs"] = float(max(dm.max(), dl.max()))
    return out


@torch.no_grad()
def measure_decode(model, name: str, golden_dir: str = GOLDEN_DIR) -> dict:
    """decode-only fixtures (golden_cases.DEC_CASES): `model.decode(z).sample` of the seeded latent against the reference's
    reconstruction, sampled at the fixture's stride (per-frame phase), plus the fp64 mean of the whole reconstruction"""
    family, over, zshape, wseed, zseed, s = DEC_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    z = seeded_input(zshape, zseed).to(model.dtype).to(model.device)
    rec = model.decode(z).sample.float().cpu()
    assert tuple(rec.shape) == tuple(int(v) for v in gold["recon_shape"]), (tuple(rec.shape), gold["recon_shape"])
    r, g = recon_subsample(rec, s).numpy(), gold["recon_sub"]
    mse = float(((r - g).astype(np.float64) ** 2).mean())
    return {"case": name, "shape": list(zshape), "recon_max_abs": float(np.abs(r - g).max()),
            "recon_psnr_db": float(10 * np.log10(4.0 / max(mse, 1e-30))),
            "recon_mean_delta": float(abs(rec.double().mean().item() - float(gold["recon_mean"]))),
            "recon_sampled_fraction": round(1.0 / (s * s), 4)}


def fmt(tag: str, m: dict) -> str:
    return (f"{m['case']:28s} {tag:5s} latent max|d| {m['latent_max_abs']:.3e} mean|d| {m['latent_mean_abs']:.3e}  "
            f"moments max|d| {m['moments_max_abs']:.3e}  recon max|d| {m['recon_max_abs']:.3e} PSNR {m['recon_psnr_db']:.2f} dB")
