"""CPU check of the HOST logic of the frozen constraint decoder's input-gradient pass (cvvae_amd/grad.py): with every kernel
replaced by a plain-PyTorch emulation of its documented arithmetic (tests/emu_ops.py), the taped forward must reproduce the
oracle's reconstruction and the backward pass autograd's gradient -- i.e. the right tensors are taped, the right weights are
transposed, and every launch is fed the right operand.  (The kernels themselves are compared with autograd on the GPU:
tests/test_gpu_grad.py.)"""
import pytest
import torch

from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from tests import emu_ops

SMALL = dict(in_channels=16, out_channels=3, up_block_types=["UpDecoderBlock2D"] * 3, block_out_channels=[128, 256, 256],
             layers_per_block=1, norm_num_groups=32, act_fn="silu", mid_block_add_attention=True)


@pytest.mark.parametrize("zshape", [(1, 16, 2, 4, 6), (2, 16, 1, 5, 4)])
def test_constraint_decoder_backward_wiring(zshape):
    from cvvae_amd import engine, grad
    from cvvae_amd.constraint import DecoderWith3DWrapper
    m = DecoderWith3DWrapper(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 5)
    m.load_state_dict(sd, strict=True)
    m = m.eval().requires_grad_(False)
    z = seeded_input(zshape, 9)
    zr = z.clone().requires_grad_(True)
    yr = O.constraint_decoder(zr, {k: v.float() for k, v in sd.items()}, SMALL)
    cot = seeded_input(tuple(yr.shape), 3)
    (yr * cot).sum().backward()
    with emu_ops.patched(), torch.no_grad():
        wc = engine.WeightCache(m)
        tape = []
        y = engine.constraint_decoder2d(wc, z, m._cfg, tape)
        assert torch.allclose(y, yr, rtol=1e-4, atol=1e-5), float((y - yr).abs().max())
        assert torch.equal(y, engine.constraint_decoder2d(wc, z, m._cfg))
        gz = grad.constraint_decoder2d_backward(wc, tape, cot)
    assert gz.shape == z.shape
    err = float((gz - zr.grad).norm() / zr.grad.norm())
    assert err < 1e-4, err
