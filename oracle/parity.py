"""Parity measurement of a device model against the reference-generated fixtures (tests/golden).  TEST INFRASTRUCTURE:
imported by tests/, __graft_entry__.smoke() and bench.py's parity leg only -- as the checker, never on the product path.

Metrics follow SURVEY 8(d): latent max / mean |delta| on `moments`, recon PSNR = 10 log10(4 / MSE) on [-1, 1] data.
"""
import os

import numpy as np
import torch

from .golden_cases import BIG_CASES, CASES, DEC_CASES, ENC_CASES, grad_sample_index, recon_subsample
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


def _rel_l2(a, b, floor=0.0):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(floor if floor else 1e-30))


def compare_grads_with_fixture(gold, part, net, y, gin):
    """(forward rel, dL/d input rel, [(error, name)] worst first) of one network against the `part` ('enc' / 'dec') half of a
    grad fixture: per parameter the worse of |norm - norm_ref| / norm_ref and the relative L2 error of the stored sample; tensors
    whose reference gradient is (numerically) zero -- the attention's key bias -- are measured against 1e-3 of the largest norm"""
    so, si = int(gold[part + "_out_stride"]), int(gold[part + "_gin_stride"])
    yf, gf = y.detach().float().cpu(), gin.detach().float().cpu()
    e_y = _rel_l2(recon_subsample(yf, so) if so > 1 else yf, gold[part + "_out_sub"])
    e_x = _rel_l2(recon_subsample(gf, si) if si > 1 else gf, gold[part + "_gin_sub"])
    names = [str(n) for n in gold[part + "_param_names"]]
    norms, lens = gold[part + "_param_grad_norm"], gold[part + "_param_sample_len"]
    samples = np.split(gold[part + "_param_grad_sample"], np.cumsum(lens)[:-1])
    pars = dict(net.named_parameters())
    assert sorted(pars) == names, sorted(set(pars) ^ set(names))
    scale = float(norms.max())
    errs = []
    for n, nr, sr in zip(names, norms, samples):
        g = pars[n].grad
        assert g is not None and g.shape == pars[n].shape, n
        g = g.detach().float().flatten().cpu()
        idx = grad_sample_index(n, g.numel())
        frac = (len(sr) / g.numel()) ** 0.5
        e_n = abs(float(g.double().norm()) - float(nr)) / max(float(nr), 1e-3 * scale)
        e_s = _rel_l2(g[idx], sr, 1e-3 * scale * frac)
        errs.append((max(e_n, e_s), n))
    return e_y, e_x, sorted(errs, reverse=True)


def measure_backward(model, name: str = "grad_sd3_t17_256", golden_dir: str = GOLDEN_DIR) -> dict:
    """`model`: a device model in train() mode carrying the case's seeded weights: forward + backward of both networks at the
    fixture's size against the gradients of the reference's own modules (tests/golden/<name>.npz, oracle/make_golden.py grad)"""
    from .golden_cases import GRAD_CASES
    family, over, shape, wseed, xseed, cseeds, zseed, strides = GRAD_CASES[name]
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    dtype, dev = model.dtype, model.device
    zshape = (shape[0], int(gold["enc_out_shape"][1]) // 2) + tuple(int(v) for v in gold["enc_out_shape"][2:])
    out = {}
    for part, net, inp, cseed in (("enc", model.encoder, seeded_input(shape, xseed), cseeds[0]),
                                  ("dec", model.decoder, seeded_input(zshape, zseed), cseeds[1])):
        net.zero_grad(set_to_none=True)
        xin = inp.to(dtype).to(dev).requires_grad_(True)
        y = net(xin)
        cot = seeded_input(tuple(y.shape), cseed).to(dtype).to(dev)
        (y.float() * cot.float()).sum().backward()
        e_y, e_x, errs = compare_grads_with_fixture(gold, part, net, y, xin.grad)
        out[part] = {"forward_rel": float(f"{e_y:.3e}"), "input_grad_rel": float(f"{e_x:.3e}"), "param_worst_rel": float(f"{errs[0][0]:.3e}"),
                     "param_worst": errs[0][1], "param_median_rel": float(f"{errs[len(errs) // 2][0]:.3e}"), "n_param_tensors": len(errs)}
        net.zero_grad(set_to_none=True)
        del y, cot, xin
    return out
