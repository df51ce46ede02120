"""Live pin of the CPU oracle: whenever /root/reference is mounted (the build container), the reference's OWN unmodified
modules are imported (oracle/ref_loader.py) and run next to the oracle's restatement on the same seeded weights and inputs.
Skipped on the GPU box (no /root/reference there): the committed fixtures of tests/golden carry the pin."""
import pytest
import torch

from oracle import cvvae_oracle as O
from oracle.ref_loader import load_reference, reference_available
from oracle.seeded import seeded_input, seeded_state_dict

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")


@pytest.mark.parametrize("family,shape", [("sd3", (1, 3, 5, 32, 32)), ("vae3d", (1, 3, 5, 32, 32)), ("sd3", (2, 3, 1, 32, 48)),
                                          ("vae3d", (1, 3, 21, 32, 32))])
def test_oracle_equals_reference_modules(family, shape):
    ref = load_reference()
    cls = ref.CVVAESD3Model if family == "sd3" else ref.CVVAEModel
    model = cls().eval()
    sd = seeded_state_dict({k: v.shape for k, v in model.state_dict().items()}, 5)
    model.load_state_dict(sd, strict=True)
    x = seeded_input(shape, 9)
    with torch.no_grad():
        post = model.encode(x).latent_dist
        rec = model.decode(post.mode()).sample
        mom = O.encode_moments(x, sd, {}, family)
        rec_o = O.decode_sample(O.posterior_mode(mom), sd, {}, family)
    assert (mom - post.parameters).abs().max() <= 2e-5
    assert (rec_o - rec).abs().max() <= 1e-4
