"""CPU check of the HOST logic of the trainable 3-D encoder's backward pass (cvvae_amd/grad3d.py): with every kernel replaced by a
plain-PyTorch emulation of its documented arithmetic (tests/emu_ops.py), the taped forward must reproduce the oracle's moments and
the backward pass torch.autograd's gradients of the oracle's ops -- for the INPUT and for EVERY PARAMETER of the network.  That
pins what is taped, which operand / padding / stride each weight-gradient launch gets, the full-correlation + pad-fold form of the
replicate-padded input gradients, the zero-stuffed strided ones, and the autograd.Function wiring (parameter order, dtypes).
(The kernels themselves are compared with autograd on the GPU: tests/test_gpu_grad3d.py.)"""
import pytest
import torch

from oracle import cvvae_oracle as O
from oracle.seeded import seeded_input, seeded_state_dict
from tests import emu_ops

SMALL = dict(block_out_channels=[128, 256, 256], layers_per_block=1)


def _rel(a, b, floor=0.0):
    """relative L2 error; `floor`: an absolute scale below which the reference counts as zero (the key bias of an attention block
    has an exactly zero gradient -- softmax rows are invariant to a constant added to every score -- and autograd returns noise)"""
    return float((a - b).norm() / b.norm().clamp_min(floor if floor else 1e-30))


@pytest.mark.parametrize("shape", [(1, 3, 5, 16, 24), (2, 3, 1, 8, 8)])
def test_sd3_encoder_backward_wiring(shape):
    import cvvae_amd
    from cvvae_amd import engine, grad3d
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 7)
    m.load_state_dict(sd, strict=True)
    enc = m.encoder
    # reference: autograd over the oracle's ops on the same weights
    ref_sd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
    x = seeded_input(shape, 11)
    xr = x.clone().requires_grad_(True)
    cfg = dict(block_out_channels=SMALL["block_out_channels"], layers_per_block=1)
    yr = O.sd3_encoder(xr, ref_sd, cfg)
    cot = seeded_input(tuple(yr.shape), 4)
    (yr * cot).sum().backward()
    with emu_ops.patched(whole_model=True):
        with torch.no_grad():
            wc = engine.WeightCache(enc)
            tape = []
            y = engine.sd3_encoder(wc, x, dict(enc._cfg), tape)
            assert torch.allclose(y, yr.detach(), rtol=1e-4, atol=1e-5), float((y - yr).abs().max())
            gx, grads = grad3d.sd3_encoder_backward(wc, tape, cot, need_input_grad=True)
        assert _rel(gx, xr.grad) < 1e-4, _rel(gx, xr.grad)
        names = [n for n, _ in enc.named_parameters()]
        assert sorted(grads) == sorted(names), sorted(set(names) ^ set(grads))
        scale = max(float(ref_sd["encoder." + n].grad.norm()) for n in names)
        worst = max((_rel(grads[n].reshape(ref_sd["encoder." + n].shape), ref_sd["encoder." + n].grad, 1e-4 * scale), n) for n in names)
        assert worst[0] < 2e-4, worst
        # the autograd.Function wiring: module in train() mode, grad mode on
        enc.train()
        xa = x.clone().requires_grad_(True)
        ya = enc(xa)
        assert ya.requires_grad and torch.equal(ya.detach(), y)
        (ya * cot).sum().backward()
        assert _rel(xa.grad, xr.grad) < 1e-4
        for n, p in enc.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape, n
            assert _rel(p.grad, ref_sd["encoder." + n].grad, 1e-4 * scale) < 2e-4, n
        # eval() mode stays the inference pass (no graph)
        enc.eval()
        assert not enc(x).requires_grad


@pytest.mark.parametrize("zshape", [(1, 16, 3, 4, 6), (2, 16, 1, 4, 4)])
def test_sd3_decoder_backward_wiring(zshape):
    """Decoder3D: conv_in over the latent, mid block, up blocks with BOTH Upsample3D kinds (time shuffle + frame drop on even
    blocks, spatial only on odd ones), norm_out + conv_out (the taps-in-N forward form): dL/dz and every parameter gradient"""
    import cvvae_amd
    from cvvae_amd import engine, grad3d
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 8)
    m.load_state_dict(sd, strict=True)
    dec = m.decoder
    ref_sd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input(zshape, 13)
    zr = z.clone().requires_grad_(True)
    cfg = dict(block_out_channels=SMALL["block_out_channels"], layers_per_block=1)
    yr = O.sd3_decoder(zr, ref_sd, cfg)
    cot = seeded_input(tuple(yr.shape), 5)
    (yr * cot).sum().backward()
    with emu_ops.patched(whole_model=True):
        dec.train()
        za = z.clone().requires_grad_(True)
        ya = dec(za)
        assert ya.requires_grad and torch.allclose(ya.detach(), yr.detach(), rtol=1e-4, atol=1e-5), float((ya - yr).abs().max())
        (ya * cot).sum().backward()
        assert _rel(za.grad, zr.grad) < 1e-4, _rel(za.grad, zr.grad)
        names = [n for n, _ in dec.named_parameters()]
        scale = max(float(ref_sd["decoder." + n].grad.norm()) for n in names)
        for n, p in dec.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape, n
            assert _rel(p.grad, ref_sd["decoder." + n].grad, 1e-4 * scale) < 2e-4, (n, _rel(p.grad, ref_sd["decoder." + n].grad))
        dec.eval()
        assert not dec(z).requires_grad


@pytest.mark.parametrize("shape,causal", [((1, 3, 5, 16, 24), True), ((1, 3, 5, 16, 16), False)])
def test_vae3d_encoder_backward_wiring(shape, causal):
    """the SD2.1-compatible family's Encoder (vae_models.py:790-823) through the same tape walker: zero H / W padding with the
    causal replicate time pad (or zero padding everywhere), Downsample3D's asymmetric (0, 1) pads, GroupNorm eps 1e-5,
    nin_shortcut, the mid block's spatial attention"""
    import cvvae_amd
    over = dict(ch=128, ch_mult=(1, 2, 2), num_res_blocks=1, causal_encoder=causal) if causal else dict(ch=128, ch_mult=(1, 2, 2), num_res_blocks=1, causal_encoder=False)
    m = cvvae_amd.CVVAEModel(**over)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 9)
    m.load_state_dict(sd, strict=True)
    enc = m.encoder
    ref_sd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
    x = seeded_input(shape, 14)
    xr = x.clone().requires_grad_(True)
    yr = O.v3_encoder(xr, ref_sd, dict(over))
    cot = seeded_input(tuple(yr.shape), 6)
    (yr * cot).sum().backward()
    with emu_ops.patched(whole_model=True):
        enc.train()
        xa = x.clone().requires_grad_(True)
        ya = enc(xa)
        assert ya.requires_grad and torch.allclose(ya.detach(), yr.detach(), rtol=1e-4, atol=1e-5), float((ya - yr).abs().max())
        (ya * cot).sum().backward()
        assert _rel(xa.grad, xr.grad) < 1e-4, _rel(xa.grad, xr.grad)
        names = [n for n, _ in enc.named_parameters()]
        scale = max(float(ref_sd["encoder." + n].grad.norm()) for n in names)
        for n, p in enc.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape, n
            assert _rel(p.grad, ref_sd["encoder." + n].grad, 1e-4 * scale) < 2e-4, (n, _rel(p.grad, ref_sd["encoder." + n].grad))


@pytest.mark.parametrize("zshape", [(1, 4, 3, 4, 6), (2, 4, 1, 4, 4)])
def test_vae3d_decoder_backward_wiring(zshape):
    """the SD2.1-compatible family's Decoder (vae_models.py:960-1002): the spatial-temporal attention block of its mid block (no
    inner residual; LayerNorm + per-pixel attention over time), Upsample3D with zero H / W padding in both kinds, norm_out"""
    import cvvae_amd
    over = dict(ch=128, ch_mult=(1, 2, 2), num_res_blocks=1)
    m = cvvae_amd.CVVAEModel(**over)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 10)
    m.load_state_dict(sd, strict=True)
    dec = m.decoder
    ref_sd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input(zshape, 15)
    zr = z.clone().requires_grad_(True)
    yr = O.v3_decoder(zr, ref_sd, dict(over))
    cot = seeded_input(tuple(yr.shape), 7)
    (yr * cot).sum().backward()
    with emu_ops.patched(whole_model=True):
        dec.train()
        za = z.clone().requires_grad_(True)
        ya = dec(za)
        assert ya.requires_grad and torch.allclose(ya.detach(), yr.detach(), rtol=1e-4, atol=1e-5), float((ya - yr).abs().max())
        (ya * cot).sum().backward()
        assert _rel(za.grad, zr.grad) < 1e-4, _rel(za.grad, zr.grad)
        names = [n for n, _ in dec.named_parameters()]
        scale = max(float(ref_sd["decoder." + n].grad.norm()) for n in names)
        for n, p in dec.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape, n
            assert _rel(p.grad, ref_sd["decoder." + n].grad, 1e-4 * scale) < 2e-4, (n, _rel(p.grad, ref_sd["decoder." + n].grad))


def test_frozen_network_gives_the_input_gradient_only():
    """a network whose parameters are frozen but whose input needs a gradient (the decoder under a latent-space loss): the same
    input gradient, no parameter gradients, and none of the weight-gradient / bias / affine launches"""
    import cvvae_amd
    from cvvae_amd import ops
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 8)
    m.load_state_dict(sd, strict=True)
    dec = m.decoder.train().requires_grad_(False)
    ref_sd = {k: v.float() for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input((1, 16, 3, 4, 6), 13)
    zr = z.clone().requires_grad_(True)
    yr = O.sd3_decoder(zr, ref_sd, dict(block_out_channels=SMALL["block_out_channels"], layers_per_block=1))
    cot = seeded_input(tuple(yr.shape), 5)
    (yr * cot).sum().backward()
    calls = []
    with emu_ops.patched(whole_model=True):
        for name in ("conv_wgrad", "bias_grad", "gn_bwd_input_params", "gn_bwd_params"):
            setattr(ops, name, (lambda n: (lambda *a, **k: calls.append(n)))(name))
        za = z.clone().requires_grad_(True)
        (dec(za) * cot).sum().backward()
    assert not calls, calls
    assert _rel(za.grad, zr.grad) < 1e-4
    assert all(p.grad is None for p in dec.parameters())


def test_last_layer_gradient_runs_only_the_tail():
    """the reference's adaptive adversarial weight asks  torch.autograd.grad(loss, decoder.get_last_layer(), retain_graph=True)  twice
    per step before the real backward (lvdm/modules/autoencoding/losses/discriminator_loss.py:211-220): the tail (GroupNorm + SiLU +
    conv_out) is its own autograd node, so those calls launch ONE weight gradient -- and the full backward afterwards still gives
    every gradient (retain_graph: the tape is not consumed)"""
    import cvvae_amd
    from cvvae_amd import ops
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 8)
    m.load_state_dict(sd, strict=True)
    dec = m.decoder.train()
    ref_sd = {k: v.float().clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    z = seeded_input((1, 16, 3, 4, 6), 13)
    zr = z.clone().requires_grad_(True)
    yr = O.sd3_decoder(zr, ref_sd, dict(block_out_channels=SMALL["block_out_channels"], layers_per_block=1))
    cot = seeded_input(tuple(yr.shape), 5)
    cot2 = seeded_input(tuple(yr.shape), 6)
    ref_last = torch.autograd.grad((yr * cot2).sum(), ref_sd["decoder.conv_out.weight"], retain_graph=True)[0]
    (yr * cot).sum().backward()
    with emu_ops.patched(whole_model=True):
        calls = []
        real_wgrad = ops.conv_wgrad
        ops.conv_wgrad = lambda *a, **k: (calls.append("conv_wgrad"), real_wgrad(*a, **k))[1]
        za = z.clone().requires_grad_(True)
        ya = dec(za)
        last = dec.get_last_layer()
        assert last is dec.conv_out.weight
        g1 = torch.autograd.grad((ya * cot2).sum(), last, retain_graph=True)[0]
        assert calls == ["conv_wgrad"], calls                     # the tail's weight gradient and nothing of the body
        g2 = torch.autograd.grad((ya * cot2).sum(), last, retain_graph=True)[0]
        assert torch.equal(g1, g2) and len(calls) == 2
        assert _rel(g1, ref_last) < 2e-4
        (ya * cot).sum().backward()                                # the step's real backward, after the two probes
    assert _rel(za.grad, zr.grad) < 1e-4
    scale = max(float(v.grad.norm()) for v in ref_sd.values())
    for n, p in dec.named_parameters():
        assert p.grad is not None, n
        assert _rel(p.grad, ref_sd["decoder." + n].grad, 1e-4 * scale) < 2e-4, n


def test_autocast_runs_an_fp32_model_on_16_bit_weight_copies():
    """torch.autocast over an fp32 model (the reference's trainer at precision 16 / bf16, main.py:905-912: fp32 master weights, 16-bit
    compute): the pass runs on 16-bit COPIES of the weight tensors (biases and norm affines stay fp32), returns the autocast dtype,
    the gradients come back in fp32 for the masters, the backward -- which autograd runs OUTSIDE the autocast context -- uses the same
    copies, and an optimizer step on the masters refreshes the copies.  Checked against a model that HOLDS those 16-bit weights."""
    import copy

    import cvvae_amd
    from cvvae_amd import modeling
    m = cvvae_amd.CVVAESD3Model(**SMALL)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 7)
    m.load_state_dict(sd, strict=True)
    m16 = copy.deepcopy(m)
    for p in m16.parameters():
        if p.dim() >= 2:
            p.data = p.data.to(torch.bfloat16)
    x = seeded_input((1, 3, 5, 16, 24), 11)
    saved = modeling._autocast_dtype
    with emu_ops.patched(whole_model=True):
        enc, enc16 = m.encoder.train(), m16.encoder.train()
        xa = x.to(torch.bfloat16).requires_grad_(True)
        y16 = enc16(xa)
        cot = seeded_input(tuple(y16.shape), 4)
        (y16.float() * cot).sum().backward()
        try:
            modeling._autocast_dtype = lambda x: torch.bfloat16      # "inside torch.autocast(dtype=bfloat16)"
            xb = x.clone().requires_grad_(True)
            y = enc(xb)
            assert y.dtype == torch.bfloat16 and torch.equal(y, y16)
            assert enc.conv_in.weight.dtype == torch.float32           # the masters are untouched
            modeling._autocast_dtype = lambda x: None                 # autograd's backward thread: no autocast state
            (y.float() * cot).sum().backward()
            assert xb.grad.dtype == torch.float32 and torch.equal(xb.grad.to(torch.bfloat16), xa.grad)
            for (n, p), (_, q) in zip(enc.named_parameters(), enc16.named_parameters()):
                assert p.grad is not None and p.grad.dtype == torch.float32, n
                assert torch.equal(p.grad.to(q.grad.dtype), q.grad), n   # (16-bit holder: its gradient is this one, rounded)
            with torch.no_grad():                                      # an optimizer step on the fp32 masters
                for p, q in zip(enc.parameters(), enc16.parameters()):
                    p.add_(1e-3 * p.grad / (p.grad.abs().max() + 1e-12))
                    q.copy_(p)
            modeling._autocast_dtype = lambda x: torch.bfloat16
            y2 = enc(x)
            assert torch.equal(y2, enc16(x.to(torch.bfloat16))) and not torch.equal(y2, y)
            modeling._autocast_dtype = lambda x: None                 # outside autocast: the fp32 model again
            assert enc.eval()(x).dtype == torch.float32
        finally:
            modeling._autocast_dtype = saved


TILED = dict(block_out_channels=[128, 256, 256], layers_per_block=1, spatial_n_compress=4, time_n_compress=2,
             en_de_n_frames_a_time=4, tile_spatial_size=36)


@pytest.mark.parametrize("recompute", [False, True])
def test_training_through_the_windowed_and_tiled_wrapper(recompute):
    """the reference's Autoencoding3DEngine chunks time and tiles space UNDER AUTOGRAD (lvdm/models/autoencoder.py:809-974); here
    `tiled_encode` / `tiled_decode` (2 windows x 2x2 blended tiles) in train() mode under grad mode: every (window, tile) call is
    its own pair of autograd nodes (or, `recompute`, one node that keeps only its input), the blends are linear nodes, crops and
    concatenations are torch's.  Gradients of the input and of EVERY parameter against autograd over the oracle's wrapper."""
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(**TILED)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 12)
    m.load_state_dict(sd, strict=True)
    ref = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    x = seeded_input((1, 3, 9, 40, 48), 31)
    xr = x.clone().requires_grad_(True)
    mr = O.encode_moments(xr, ref, dict(TILED), "sd3")
    zr = mr[:, :16]
    yr = O.decode_sample(zr, ref, dict(TILED), "sd3")
    cm, cy = seeded_input(tuple(mr.shape), 5), seeded_input(tuple(yr.shape), 6)
    ((mr * cm).sum() + (yr * cy).sum()).backward()
    with emu_ops.patched(whole_model=True):
        m.train()
        m.encoder.recompute = m.decoder.recompute = recompute
        assert len(m._windows(9, m.encode_n_frames_a_time)) == 2 and len(m._tile_grid(40, 48, 36, 28)) == 2
        xa = x.clone().requires_grad_(True)
        mo = m.tiled_encode(xa)
        assert mo.requires_grad and torch.allclose(mo.detach(), mr.detach(), rtol=1e-4, atol=1e-5), float((mo - mr).abs().max())
        ya = m.tiled_decode(mo[:, :16])
        assert torch.allclose(ya.detach(), yr.detach(), rtol=1e-4, atol=1e-4), float((ya - yr).abs().max())  # (z itself differs by ~5e-6)
        ((mo * cm).sum() + (ya * cy).sum()).backward()
        assert _rel(xa.grad, xr.grad) < 2e-4, _rel(xa.grad, xr.grad)
        scale = max(float(v.grad.norm()) for v in ref.values())
        for n, p in m.named_parameters():
            assert p.grad is not None and p.grad.shape == p.shape, n
            assert _rel(p.grad, ref[n].grad, 1e-4 * scale) < 3e-4, (n, _rel(p.grad, ref[n].grad))
        # the public encode() / decode() stay inference entry points (no graph), whatever the module mode
        assert not m.encode(x).latent_dist.parameters.requires_grad


def test_refresh_weights_after_a_write_through_data():
    """`p.data.copy_()` (the reference's EMA swap, lvdm/modules/ema.py:61-86) leaves `p._version` alone, so the packed forms keyed on
    it go stale: refresh_weights() / the training path's checksum guard / weight_guard pick the new weights up"""
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(block_out_channels=[128, 256, 256], layers_per_block=1)
    sd = seeded_state_dict({k: v.shape for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = seeded_input((1, 3, 5, 16, 16), 2)
    with emu_ops.patched(whole_model=True):
        y0 = m.encoder(x)
        w = m.encoder.conv_in.weight
        v0 = w._version
        w.data.mul_(1.5)
        assert w._version == v0                       # the write is invisible to the cache's keys (on the GPU the stale packed
        assert m.refresh_weights()                    # weights would still answer: tests/test_gpu_round5.py; the emulated packers alias)
        y1 = m.encoder(x)
        assert not torch.allclose(y1, y0)
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2["encoder.conv_in.weight"] = sd["encoder.conv_in.weight"] * 1.5
        ref = O.sd3_encoder(x, {k: v for k, v in sd2.items()}, dict(block_out_channels=[128, 256, 256], layers_per_block=1))
        assert torch.allclose(y1, ref, rtol=1e-4, atol=1e-5)
        # the opt-in guard of the inference path: a checksum per pass
        m.encoder.weight_guard = True
        m.encoder(x)                                  # (takes the first checksum)
        w.data.mul_(2.0)
        y2 = m.encoder(x)
        assert not torch.allclose(y2, y1)
        m.encoder.weight_guard = False
        # a train() / eval() transition arms one check
        m.train()
        xa = x.clone().requires_grad_(True)
        ya = m.encoder(xa)
        m.eval()
        w.data.mul_(0.5)
        yb = m.encoder(x)
        assert torch.allclose(yb, y1, rtol=1e-5, atol=1e-6) and not torch.allclose(ya.detach(), yb)


def test_backward_after_a_parameter_update_raises():
    """the backward reads the weights live: a parameter modified between forward and backward must raise (as PyTorch's own conv
    backward does), not return gradients of the new weights over the old activations"""
    import cvvae_amd
    m = cvvae_amd.CVVAESD3Model(block_out_channels=[128, 256, 256], layers_per_block=1)
    m.train()
    x = seeded_input((1, 3, 5, 16, 16), 2)
    with emu_ops.patched(whole_model=True):
        y = m.encoder(x)
        with torch.no_grad():
            m.encoder.conv_in.weight.mul_(1.01)       # an optimizer.step() between forward and backward
        with pytest.raises(RuntimeError, match="modified"):
            y.sum().backward()
