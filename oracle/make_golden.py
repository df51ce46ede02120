"""Generate tests/golden/*.npz by running the reference's OWN modules (imported unmodified from
/root/reference through oracle/ref_loader.py) on seeded weights and inputs.  TEST INFRASTRUCTURE.

Run in the build container only:   python -m oracle.make_golden
Each fixture holds: moments (encode().latent_dist.parameters), recon (decode(mode()).sample), both fp32,
plus a weight checksum so a drift of oracle/seeded.py is detected.  Inputs/weights are NOT stored: they are
re-derived from (key, shape, seed) by oracle/seeded.py wherever the fixture is consumed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.golden_cases import CASES  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402
from oracle.seeded import seeded_input, seeded_state_dict  # noqa: E402


def main():
    ref = load_reference()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, (family, over, shape, wseed, xseed) in CASES.items():
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        model = cls(**over).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        x = seeded_input(shape, xseed)
        post = model.encode(x).latent_dist
        moments = post.parameters
        recon = model.decode(post.mode()).sample
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            moments=moments.numpy().astype(np.float32),
            recon=recon.numpy().astype(np.float32),
            weight_abs_sum=np.float64(wsum),
            n_tensors=np.int64(len(sd)),
        )
        print(f"{name}: moments {tuple(moments.shape)} recon {tuple(recon.shape)} wsum {wsum:.6f}")


def main_constraint():
    """fixtures of the frozen 2-D constraint decoder, from the reference's own DecoderWith3DWrapper"""
    from oracle.golden_cases import CONSTRAINT_CASES
    from oracle.ref_loader import load_reference_constraint

    ref = load_reference_constraint()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (cfg, zshape, wseed, zseed) in CONSTRAINT_CASES.items():
        model = ref.DecoderWith3DWrapper(**cfg).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        z = seeded_input(zshape, zseed)
        recon = model(z)
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), recon=recon.numpy().astype(np.float32),
                            weight_abs_sum=np.float64(wsum), n_tensors=np.int64(len(sd)))
        print(f"{name}: latents {tuple(zshape)} recon {tuple(recon.shape)} wsum {wsum:.6f}")


def main_ldm():
    """fixtures of the frozen SD2.1-family 2-D encoder / decoder wrappers, from the reference's own classes"""
    from oracle.golden_cases import LDM_CASES
    from oracle.ref_loader import load_reference_ldm

    ref = load_reference_ldm()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (cls, cfg, shape, wseed, xseed) in LDM_CASES.items():
        model = getattr(ref, cls)(**cfg).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        out = model(seeded_input(shape, xseed))
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), out=out.numpy().astype(np.float32),
                            weight_abs_sum=np.float64(wsum), n_tensors=np.int64(len(sd)))
        print(f"{name}: in {tuple(shape)} out {tuple(out.shape)} wsum {wsum:.6f}")


def main_big(only=None):
    """BASELINE-size fixtures (cfg 1 / 2 / 3 and one window of cfg 4) from the reference's own modules; minutes of CPU each.
    The reference's fp16 / bf16 CPU runs of the same case are recorded too (its own low-precision noise at THIS shape: the
    yardstick the GPU tolerances are read against)."""
    import time

    from oracle.golden_cases import BIG_CASES, recon_subsample

    ref = load_reference()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (family, over, shape, wseed, xseed, s) in BIG_CASES.items():
        if only and name not in only:
            continue
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        model = cls(**over).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        x = seeded_input(shape, xseed)
        t0 = time.time()
        post = model.encode(x).latent_dist
        t1 = time.time()
        moments = post.parameters
        recon = model.decode(post.mode()).sample
        t2 = time.time()
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        r64 = recon.double()
        mnp = moments.numpy().astype(np.float32)
        zc = mnp.shape[1] // 2
        # fixtures stay small: above 4 MB the posterior mean is kept in full and the log-variance at stride 2 over H and W
        mom_kw = dict(moments=mnp) if mnp.nbytes <= (4 << 20) else dict(
            moments_mean=(mnp[:, :zc].copy() if ms == 1 else recon_subsample(mnp[:, :zc], ms).copy()), mean_stride=np.int64(ms),
            moments_logvar_sub=(mnp[:, zc:, :, ::2, ::2].copy() if ms == 1 else recon_subsample(mnp[:, zc:], 2 * ms).copy()),
            moments_shape=np.asarray(mnp.shape, dtype=np.int64))
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            recon_sub=recon_subsample(recon, s).numpy().astype(np.float32),
            **mom_kw,
            recon_shape=np.asarray(recon.shape, dtype=np.int64),
            recon_stride=np.int64(s),
            recon_mean=np.float64(r64.mean()), recon_sqmean=np.float64((r64 * r64).mean()),
            recon_frame_mean=r64.mean(dim=(0, 1, 3, 4)).numpy(),
            weight_abs_sum=np.float64(wsum),
            n_tensors=np.int64(len(sd)),
            ref_cpu_seconds=np.asarray([t1 - t0, t2 - t1]), ref_cpu_threads=np.int64(torch.get_num_threads()),
        )
        print(f"{name}: moments {tuple(moments.shape)} recon {tuple(recon.shape)} wsum {wsum:.6f} "
              f"reference CPU fp32 encode {t1 - t0:.1f}s decode {t2 - t1:.1f}s ({torch.get_num_threads()} threads)", flush=True)


def main_enc(only=None):
    """encode-only fixtures at full size (BASELINE cfg 5's batch slice) from the reference's own modules"""
    import time

    from oracle.golden_cases import ENC_CASES

    ref = load_reference()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    from oracle.golden_cases import recon_subsample
    for name, case in ENC_CASES.items():
        family, over, shape, wseed, xseed = case[:5]
        ms = case[5] if len(case) > 5 else 1
        if only and name not in only:
            continue
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        model = cls(**over).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        x = seeded_input(shape, xseed)
        t0 = time.time()
        mnp = model.encode(x).latent_dist.parameters.numpy().astype(np.float32)
        t1 = time.time()
        np.save(os.path.join(os.path.dirname(HERE), "gpurun_out", name + "_raw_moments.npy"), mnp)  # (hours of CPU: keep the raw result)
        zc = mnp.shape[1] // 2
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            moments_mean=(mnp[:, :zc].copy() if ms == 1 else recon_subsample(mnp[:, :zc], ms).copy()), mean_stride=np.int64(ms),
            moments_logvar_sub=(mnp[:, zc:, :, ::2, ::2].copy() if ms == 1 else recon_subsample(mnp[:, zc:], 2 * ms).copy()),
            moments_shape=np.asarray(mnp.shape, dtype=np.int64),
            moments_mean_f64=np.float64(mnp.astype(np.float64).mean()),
            weight_abs_sum=np.float64(wsum), n_tensors=np.int64(len(sd)),
            ref_cpu_seconds=np.asarray([t1 - t0]), ref_cpu_threads=np.int64(torch.get_num_threads()),
        )
        print(f"{name}: moments {tuple(mnp.shape)} wsum {wsum:.6f} reference CPU fp32 encode {t1 - t0:.1f}s "
              f"({torch.get_num_threads()} threads)", flush=True)


def main_dec(only=None):
    """decode-only fixtures at full size (BASELINE cfg 4's decode side) from the reference's own modules"""
    import time

    from oracle.golden_cases import DEC_CASES, recon_subsample

    ref = load_reference()
    torch.set_grad_enabled(False)
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (family, over, zshape, wseed, zseed, s) in DEC_CASES.items():
        if only and name not in only:
            continue
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        model = cls(**over).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        z = seeded_input(zshape, zseed)
        t0 = time.time()
        recon = model.decode(z).sample
        t1 = time.time()
        r64 = recon.double()
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            recon_sub=recon_subsample(recon, s).numpy().astype(np.float32), recon_shape=np.asarray(recon.shape, dtype=np.int64),
            recon_stride=np.int64(s), recon_mean=np.float64(r64.mean()), recon_sqmean=np.float64((r64 * r64).mean()),
            recon_frame_mean=r64.mean(dim=(0, 1, 3, 4)).numpy(),
            weight_abs_sum=np.float64(wsum), n_tensors=np.int64(len(sd)),
            ref_cpu_seconds=np.asarray([t1 - t0]), ref_cpu_threads=np.int64(torch.get_num_threads()),
        )
        print(f"{name}: latent {tuple(zshape)} recon {tuple(recon.shape)} wsum {wsum:.6f} reference CPU fp32 decode {t1 - t0:.1f}s "
              f"({torch.get_num_threads()} threads)", flush=True)


def main_grad(only=None):
    """BACKWARD fixtures from the reference's own Encoder3D / Decoder3D under torch.autograd (golden_cases.GRAD_CASES): the
    training step of /root/reference/lvdm/models/autoencoder.py:1057-1090 differentiates exactly these modules
    (models/vae_models3d_sd3.py:162-208, 323-388).  eval() mode: the modules' torch.utils.checkpoint wrapping (train() mode)
    recomputes the same ops and gives the same gradients; dropout is 0."""
    import time

    from oracle.golden_cases import GRAD_CASES, grad_sample_index, recon_subsample

    ref = load_reference()
    out_dir = os.path.join(os.path.dirname(HERE), "tests", "golden")
    for name, (family, over, shape, wseed, xseed, cseeds, zseed, strides) in GRAD_CASES.items():
        if only and name not in only:
            continue
        cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
        model = cls(**over).eval()
        sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, wseed)
        model.load_state_dict(sd, strict=True)
        wsum = float(sum(v.double().abs().sum() for v in sd.values()))
        out = dict(weight_abs_sum=np.float64(wsum), n_tensors=np.int64(len(sd)), ref_cpu_threads=np.int64(torch.get_num_threads()))
        secs = []
        for part, net, inp, cseed, s in (
                ("enc", model.encoder, seeded_input(shape, xseed), cseeds[0], strides[0]),
                ("dec", model.decoder, None, cseeds[1], strides[1])):
            if inp is None:  # the decoder is differentiated at a SEEDED latent of the encoder's output geometry
                zc = out["enc_out_shape"][1] // 2 if family == "sd3" else out["enc_out_shape"][1] // 2
                inp = seeded_input((shape[0], int(zc)) + tuple(int(v) for v in out["enc_out_shape"][2:]), zseed)
            for p in net.parameters():
                p.grad = None
            x = inp.clone().requires_grad_(True)
            t0 = time.time()
            y = net(x)
            cot = seeded_input(tuple(y.shape), cseed)
            (y * cot).sum().backward()
            secs.append(time.time() - t0)
            out[part + "_out_shape"] = np.asarray(y.shape, dtype=np.int64)
            yd = y.detach()
            out[part + "_out_sub"] = (recon_subsample(yd, 4) if part == "dec" else yd).numpy().astype(np.float32)
            out[part + "_out_stride"] = np.int64(4 if part == "dec" else 1)
            gx = x.grad.detach()
            out[part + "_gin_sub"] = (recon_subsample(gx, s) if s > 1 else gx).numpy().astype(np.float32)
            out[part + "_gin_stride"] = np.int64(s)
            out[part + "_gin_norm"] = np.float64(gx.double().norm())
            names = sorted(n for n, _ in net.named_parameters())
            pars = dict(net.named_parameters())
            norms, samples = [], []
            for n in names:
                g = pars[n].grad.detach().flatten()
                norms.append(float(g.double().norm()))
                samples.append(g[grad_sample_index(n, g.numel())].numpy().astype(np.float32))
            out[part + "_param_names"] = np.asarray(names)
            out[part + "_param_grad_norm"] = np.asarray(norms, dtype=np.float64)
            out[part + "_param_grad_sample"] = np.concatenate(samples)
            out[part + "_param_sample_len"] = np.asarray([len(v) for v in samples], dtype=np.int64)
            print(f"{name} {part}: in {tuple(inp.shape)} out {tuple(y.shape)} |dL/din| {float(gx.norm()):.4e} {len(names)} parameter "
                  f"tensors, reference CPU fp32 forward+backward {secs[-1]:.1f}s ({torch.get_num_threads()} threads)", flush=True)
            del y, x, cot
        out["ref_cpu_seconds"] = np.asarray(secs)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grad":
        main_grad(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "dec":
        main_dec(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "enc":
        main_enc(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "constraint":
        main_constraint()
    elif len(sys.argv) > 1 and sys.argv[1] == "ldm":
        main_ldm()
    elif len(sys.argv) > 1 and sys.argv[1] == "big":
        main_big(sys.argv[2:])
    else:
        main()
        main_constraint()
