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


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "constraint":
        main_constraint()
    else:
        main()
        main_constraint()
